// extend_kernels.h -- launch interface of the device half of the extension stage behind the planner (round 6). Everything between
// the planner's band list and the match records happens in HBM, ranking chunk by ranking chunk (/root/reference/src/align/extend.cpp:
// 289-336: a query's targets are taken ranking_chunk_size at a time in the order of their seed-hit scores until a chunk brings no new
// hit and the tail rule says stop); most queries have one chunk:
//   DpTargets of round 1 from the bands, their launch order (band class ascending, longest first), trace offsets and item pairs
//                         (what api.hip's dmnd_swipe_keep prepares on the host: DP::BandedSwipe::bin, /root/reference/src/dp/swipe/swipe_wrapper.cpp:75-102)
//   best HSP per target, report cutoff          (/root/reference/src/align/gapped_score.cpp:182-268, target.h:97-113)
//   culling: sort by (e-value, score, target), first -k targets            (/root/reference/src/align/culling.cpp:97-113, 189-203)
//   round 2 = a walk of the kept traces of the survivors                     (/root/reference/src/align/gapped_final.cpp:66-160)
//   match records in (query, e-value, score, target) order                   (/root/reference/src/align/extend.h:51-56, extend.cpp:341)
// The e-value is double arithmetic with exp / erfc (evalue.h); the device's versions of those differ from the host library's in the
// last bits, so the device value only DECIDES (cutoff, order, first k) and a decision that two values closer than 1e-9 relative
// could flip marks the query `ambiguous`: the host redoes that query. The records leave with the device value; the host overwrites
// it with its own (and the bit score) and checks the order. Queries with a group the planner left to the host, with an item the
// traceback path cannot take or with more than EXT_MAX_GROUPS groups stay on the host path (extend_host.hip extend_range).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/diamond_hip.h"
#include "plan_kernels.h"
#include "swipe_kernels.h"

namespace dmnd {

enum { EXT_CLASSES = 16, EXT_MAX_CHUNK = 1024, EXT_MAX_GROUPS = 1 << 16, EXT_MAX_ITERATIONS = 64 };

struct ExtEvalue {             // Evaluer (evalue.h) as plain data + the report cutoff
	double lambda, K, ln_k, db_letters, a, b, alpha, beta, sigma, tau, v_thr, c_thr, max_evalue;
};

struct ExtCounters {
	uint32_t n_items;                        // of the current iteration
	uint32_t n_active;                       // queries that go on to another ranking chunk
	uint32_t n_saturated, n_kept, n_ambiguous, n_resweep;      // n_resweep: survivors whose round-1 sweep kept no trace
	int32_t tb_status;                       // traceback_kernel's status word (0 = every walk ended at a cell with score 0)
	uint32_t pad;                            // (the arrays and the 64-bit counters below lie back to back: reset_iteration clears them in one go)
	uint32_t class_count[EXT_CLASSES];       // items of the current iteration per launch class c (P = class_of_index(c), swipe_core.h)
	uint32_t class_max_steps[EXT_CLASSES];
	unsigned long long total_rows;           // trace bytes of the current iteration's items
	unsigned long long cells1, cells2;       // DP cells of all round-1 items / of the items walked in round 2 (the reference's round-2 targets)
	unsigned long long cells_again;          // ... of those of them that round 2 swept again (their round-1 sweep kept no trace rows)
	unsigned long long window_targets, window_bound;      // of the current iteration: targets in the active queries' windows, and sum over the queries of min(-k, targets): what can survive the culling
	unsigned long long diag_steps, lane_steps;   // over the round-1 items: band diagonals x anti-diagonal steps, and the 128 P diagonals the item's wavefront holds x steps (lane use of the sweeps)
};

struct ExtArgs {
	// planner output (HBM) and the blocks' limits
	const PlanGroup* groups; const PlanQuery* queries; const PlanBand* bands;
	uint32_t n_groups, n_queries, n_bands;
	const dmnd_seed_hit* hits;
	const int64_t* qlimits; const int64_t* tlimits;
	int use_cbs;
	uint32_t row_min_items;        // items of an iteration from which on the row classes of the packed 16-bit sweeps are used (sweep_rows_min_items)
	uint32_t chunk_size;           // ranking_chunk_size
	int k;                         // max_target_seqs
	int64_t max_swipe_dp;
	ExtEvalue ev;
	// per query
	uint8_t* qstate;               // EXT_Q_*
	uint8_t* q_active;             // still ranking: its window [q_i0, q_i1) of the order below is the next chunk
	uint32_t* q_i0; uint32_t* q_i1;
	int32_t* q_tail; int32_t* q_prev;        // tail_score / previous_tail_score of the ranking loop
	uint32_t* q_swept;             // targets of its current window that are swept (0: none, or not active)
	// per group
	uint64_t* okeys; uint64_t* okeys_sorted; uint32_t* oidx;      // ranking order: sort keys (query, 0xffff - score), group numbers
	uint32_t* gorder;              // groups of a query in ranking order (TargetScore::operator<: score descending, then load order)
	uint8_t* aligned;              // the target is in the query's aligned_targets
	uint32_t* g_first; uint32_t* g_cnt;      // its round-1 items (once its chunk has been swept)
	uint32_t* cnt; uint32_t* item_off;       // (+ 1) items of the current iteration per group, exclusive scan
	uint32_t* kept; uint32_t* kept_pos;      // (+ 1) survives the final culling / its record slot
	uint32_t* cand_item;           // the item of its best HSP
	double* cand_ev;
	// per item: all iterations' items one after the other (a group is swept once, so n_bands bounds them)
	uint32_t item_base;            // first item of the current iteration
	uint32_t item_cap;             // room in the per-item arrays: n_bands + one copy of every survivor (round 2 sweeps those again whose traces were not kept)
	dmnd_dp_target* items;
	int64_t* off_item;             // trace offset: inside the iteration's arena for its sweeps, from the first arena on afterwards
	int32_t* p_of_item;
	SwipeEnd* ends;
	dmnd_hsp* hsps;
	// per item of the current iteration (indices relative to item_base)
	uint32_t* keys; uint32_t* keys_sorted; uint32_t* idx; uint32_t* order;      // launch order: slot -> item
	int64_t* rows; int64_t* rows_slot; int64_t* off_slot;
	int32_t* pairs;
	// round 2 and output
	int32_t* r2_order; int32_t* r2_p; int64_t* r2_off; int64_t* r2_tr;       // slot -> item, band class, trace offset, (zero) transcript offsets
	uint32_t* r2_group;            // slot -> group
	dmnd_match* records;
	ExtCounters* ctr;
	void** scan_tmp; size_t* scan_tmp_bytes;
};

enum { EXT_Q_HOST = 0, EXT_Q_DEVICE = 1, EXT_Q_AMBIGUOUS = 2 };

// once per call: which queries run here, their ranking order, the state of their ranking loops
hipError_t launch_ext_begin(const ExtArgs& a, hipStream_t st);
// one iteration, before its sweeps: the DpTargets of every active query's current chunk, launch order, trace offsets, pairs
// (a.item_base = items of the earlier iterations); the counts stay in ctr
hipError_t launch_ext_prepare(const ExtArgs& a, hipStream_t st);
// ... behind its sweeps: best HSP per target, append_hits, the next chunk or the end of the query's ranking. kept: the sweeps ran in
// traceback mode and their trace rows stay (rel = the iteration's arena relative to the first one); else round 2 sweeps the
// survivors of this iteration again. Then -- speculatively: it only counts if ctr->n_active comes back 0 -- the final culling and
// the round-2 list
hipError_t launch_ext_append(const ExtArgs& a, uint32_t n_items, bool kept, int64_t rel, hipStream_t st);
// round 2, first half, for the survivors without kept trace rows: copies of their items as one more iteration (launch order, trace
// offsets, pairs); then, behind its traceback-mode sweeps, launch_ext_rewalk points the round-2 list at the copies
hipError_t launch_ext_resweep(const ExtArgs& a, uint32_t n_kept, hipStream_t st);
hipError_t launch_ext_rewalk(const ExtArgs& a, uint32_t n_items, uint32_t n_kept, int64_t rel, hipStream_t st);
// after the walk: the records
hipError_t launch_ext_records(const ExtArgs& a, uint32_t n_kept, hipStream_t st);

// the host's (e-value, bit score) pairs into the records where they lie in HBM; records of a context gathered for a join with
// their block-local target ids turned into database-wide ordinals
hipError_t launch_ext_patch(dmnd_match* records, const double* ev_bits, uint32_t n, hipStream_t st);
hipError_t launch_ext_gather(dmnd_match* dst, const dmnd_match* src, uint32_t n, uint32_t target_offset, hipStream_t st);

}  // namespace dmnd
