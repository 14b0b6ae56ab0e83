// extend_kernels.h -- launch interface of the device half of the extension stage behind the planner (round 6). For the queries
// whose targets fit ONE ranking chunk (the common case: at most ranking_chunk_size (query, target) groups) everything between the
// planner's band list and the match records happens in HBM:
//   DpTargets of round 1 from the bands, their launch order (band class ascending, longest first), trace offsets and item pairs
//                         (what api.hip's dmnd_swipe_keep prepares on the host: DP::BandedSwipe::bin, /root/reference/src/dp/swipe/swipe_wrapper.cpp:75-102)
//   best HSP per target, report cutoff          (/root/reference/src/align/gapped_score.cpp:182-268, target.h:97-113)
//   culling: sort by (e-value, score, target), first -k targets            (/root/reference/src/align/culling.cpp:97-113, 189-203)
//   round 2 = a walk of the kept traces of the survivors                     (/root/reference/src/align/gapped_final.cpp:66-160)
//   match records in (query, e-value, score, target) order                   (/root/reference/src/align/extend.h:51-56, extend.cpp:341)
// The e-value is double arithmetic with exp / erfc (evalue.h); the device's versions of those differ from the host library's in the
// last bits, so the device value only DECIDES (cutoff, order, first k) and a decision that two values closer than 1e-9 relative
// could flip marks the query `ambiguous`: the host redoes that query. The records leave with the device value; the host overwrites
// it with its own (and the bit score) and checks the order. Queries with more groups than a chunk, with a group the planner left
// to the host, or with an item the traceback path cannot take stay on the host path (extend_host.hip extend_range).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/diamond_hip.h"
#include "plan_kernels.h"
#include "swipe_kernels.h"

namespace dmnd {

enum { EXT_CLASSES = 16, EXT_MAX_CHUNK = 1024 };

struct ExtEvalue {             // Evaluer (evalue.h) as plain data + the report cutoff
	double lambda, K, db_letters, a, b, alpha, beta, sigma, tau, v_thr, c_thr, max_evalue;
};

struct ExtCounters {
	uint32_t n_items, n_eligible, n_saturated, n_kept, n_ambiguous;
	int32_t tb_status;                       // traceback_kernel's status word (0 = every walk ended at a cell with score 0)
	uint32_t class_count[EXT_CLASSES];       // items per band class P = 1 << c
	uint32_t class_max_steps[EXT_CLASSES];
	unsigned long long total_rows;           // trace bytes of all items
	unsigned long long cells1, cells2;       // DP cells of the round-1 items / of the items walked in round 2
};

struct ExtArgs {
	// planner output (HBM) and the blocks' limits
	const PlanGroup* groups; const PlanQuery* queries; const PlanBand* bands;
	uint32_t n_groups, n_queries, n_bands;
	const dmnd_seed_hit* hits;
	const int64_t* qlimits; const int64_t* tlimits;
	int use_cbs;
	uint32_t chunk_size;           // ranking_chunk_size: a query with more groups is ranked in chunks -- on the host
	int k;                         // max_target_seqs
	int64_t max_swipe_dp;
	ExtEvalue ev;
	// work arrays
	uint32_t* gq;                  // group -> index of its query in `queries`, 0xffffffff: the query stays on the host
	uint8_t* qstate;               // per query: EXT_Q_*
	uint32_t* cnt; uint32_t* item_off;       // per group (+ 1): its round-1 items
	dmnd_dp_target* items;
	uint32_t* item_group;
	uint32_t* keys; uint32_t* keys_sorted; uint32_t* idx; uint32_t* order;      // launch order: slot -> item
	int64_t* rows; int64_t* rows_slot; int64_t* off_slot; int64_t* off_item;
	int32_t* p_of_item;
	int32_t* pairs;
	SwipeEnd* ends;                // filled by the sweeps
	// selection
	uint32_t* kept; uint32_t* kept_pos;      // per group (+ 1): survives the culling / its record slot
	uint32_t* cand_item;           // per group: the item of its best HSP
	double* cand_ev;
	int32_t* r2_order; int32_t* r2_p; int64_t* r2_off; int64_t* r2_tr;       // round-2 walk: slot -> item, band class, trace offset, (zero) transcript offsets
	dmnd_hsp* hsps;                // indexed by item
	dmnd_match* records;
	ExtCounters* ctr;
	void** scan_tmp; size_t* scan_tmp_bytes;
};

enum { EXT_Q_HOST = 0, EXT_Q_DEVICE = 1, EXT_Q_AMBIGUOUS = 2 };

// items, launch order, trace offsets, pairs (everything the sweeps read); sizes by the upper bound n_bands, the counts stay in ctr
hipError_t launch_ext_prepare(const ExtArgs& a, hipStream_t st);
// after the sweeps: best HSP per target, culling, the round-2 list
hipError_t launch_ext_select(const ExtArgs& a, hipStream_t st);
// after the walk: the records
hipError_t launch_ext_records(const ExtArgs& a, uint32_t n_kept, hipStream_t st);

}  // namespace dmnd
