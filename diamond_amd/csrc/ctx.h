// ctx.h -- shared internals of libdiamond_hip.so: error plumbing, device buffers, the context object.
#pragma once
#include "tuning.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/diamond_hip.h"
#include "evalue.h"

namespace dmnd {

int fail(int code, const std::string& msg);      // sets dmnd_last_error(), returns code

#define HIP_TRY(expr)                                                                                     \
	do {                                                                                                  \
		hipError_t e_ = (expr);                                                                           \
		if (e_ != hipSuccess)                                                                             \
			return ::dmnd::fail(DMND_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));        \
	} while (0)

// Blocking copy on the context's own stream. A plain hipMemcpy runs on the null stream, which waits for (and stalls) every
// blocking stream of the process: the concurrent sub-batches of dmnd_extend and independent contexts would serialise on it.
// Host wait for a stream. hipStreamSynchronize spins on the completion signal: a runner of dmnd_extend or the seed stage's
// thread then burns a whole core for as long as the GPU works, which is what a CPU-quota'd container (cgroup cpu.max; the
// MI355X boxes of this project give 16 CPUs to 256 hardware threads) can least afford -- the quota runs out and every thread
// of the process is frozen for the rest of the 100 ms period. So by default the wait is an interrupt-driven one on a
// hipEventBlockingSync event on a device that dmnd_init put into hipDeviceScheduleBlockingSync mode (round 6: the event flag
// alone does not stop the runtime from spinning); DMND_SPIN_SYNC=1 restores the spinning wait (lowest latency on an idle host).
inline bool spin_sync()
{
	return dmnd::tuning().spin_sync;
}

// One interrupt-driven event per STREAM (created on first use, released by forget_stream when the stream's owner goes away).
// The event used to be thread-local: the extension stage's runner threads are short-lived, so every dmnd_extend call leaked a
// handful of events and their interrupt signals, and after a few thousand calls the driver's finite pool of them ran out --
// waits then returned early ("device not ready" from hipEventElapsedTime, stale results behind copy_now).
hipError_t sync_stream(hipStream_t s);
hipError_t wait_event(hipEvent_t ev);      // the wait of sync_stream on an event of the caller's: polls DMND_SYNC_SPIN_US microseconds, then sleeps
void forget_stream(hipStream_t s);

inline hipError_t copy_now(hipStream_t s, void* dst, const void* src, size_t bytes, hipMemcpyKind kind)
{
	const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, s);
	return e != hipSuccess ? e : sync_stream(s);
}

struct DevBuf {
	void* p = nullptr;
	size_t cap = 0;
	int ensure(size_t bytes);
	bool own = true;                       // false: alias of another context's buffer (auxiliary contexts)
	void release() { if (p && own) (void)hipFree(p); p = nullptr; cap = 0; own = true; }
	template<typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// page-locked host staging (grows, never shrinks): asynchronous copies to and from it are real DMA transfers, and one buffer
// holding several arrays goes to the device as ONE copy
struct PinBuf {
	void* p = nullptr;
	size_t cap = 0;
	int ensure(size_t bytes);
	void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
	template<typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace dmnd

// DMND_TRACE=1: wall-clock laps of a call on stderr (what the host waits for inside an entry point)
struct TraceLaps {
	const char* call;
	bool on;
	std::chrono::steady_clock::time_point t0, last;
	explicit TraceLaps(const char* name) : call(name) { static const bool env = std::getenv("DMND_TRACE") != nullptr; on = env; if (on) t0 = last = std::chrono::steady_clock::now(); }
	void lap(const char* what)
	{
		if (!on) return;
		const auto now = std::chrono::steady_clock::now();
		std::fprintf(stderr, "%s %8.2f ms (+%.2f)  %s\n", call, std::chrono::duration<double, std::milli>(now - t0).count(), std::chrono::duration<double, std::milli>(now - last).count(), what);
		last = now;
	}
};

struct dmnd_ctx;
namespace dmnd { int download_bytes(dmnd_ctx* c, void* dst, const void* src, size_t bytes); }      // api.hip: HBM -> pageable host memory through the context's page-locked chunks

struct KeptTrace;

struct dmnd_ctx {
	int device = 0;
	int pool_id = -1;                           // host worker pool of this context's dmnd_extend calls (host_pool.h)
	hipStream_t stream = nullptr;
	hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
	dmnd_params params;
	dmnd::Evaluer evaluer;
	dmnd::DevBuf block[2], cbs, matrix, bias_ids;
	dmnd::DevBuf adj_matrices;                 // composition-adjusted scoring matrices of the items this context sweeps (dmnd_upload_matrices;
	int64_t n_adj_matrices = 0;                // dmnd_extend: those of the targets planned so far), 32 x 32 int8 each
	dmnd::DevBuf xd_hits, xd_out;             // device x-drop stage of dmnd_extend: the call's seed hits, one XdropSeg per hit
	dmnd::PinBuf xd_host;
	dmnd::DevBuf plan_dev;                    // device planner of dmnd_extend (plan_kernels.hip): its work arrays ...
	dmnd::PinBuf plan_host;                   // ... and the group / query / band lists it hands to the host
	void* plan_tmp = nullptr; size_t plan_tmp_bytes = 0;      // rocPRIM scan scratch of the planner
	dmnd::DevBuf ext_dev, ext_trace;          // device half of dmnd_extend behind the planner (extend_kernels.hip): work arrays, kept traces
	std::vector<dmnd::DevBuf> ext_trace_more; // ... the kept traces of the ranking chunks behind the first
	dmnd::DevBuf ext_ev;                      // the host's (e-value, bit score) pairs on their way into the device copy of the records
	const dmnd_match* ext_records_dev = nullptr; int64_t ext_records_n = -1;      // the records of the last dmnd_extend where they lie in HBM (complete: host e-values in); n = -1: part of them only exists on the host
	dmnd::PinBuf ext_host;                    // ... its counters, records and query states on the host
	double ext_dev_stats[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };      // of the last dmnd_extend: queries extended on the device, of them redone by the host (ambiguous e-value order / 16-bit saturation), items, records, band diagonals x steps, wavefront diagonals x steps
	std::vector<int32_t> h_bias_ids;           // block sequence ids of the queries with seed hits (Hauser bias of one dmnd_extend call)
	int64_t block_len[2] = { 0, 0 }, cbs_len = 0;
	std::vector<int64_t> limits[2];
	// coarse[which][p >> COARSE_SHIFT] = index of the last sequence starting at or before block offset (p >> COARSE_SHIFT) << COARSE_SHIFT:
	// the host's "which sequence holds this block offset" lookups (one per seed hit in load_hits) become one read of this
	// L2-sized table plus a search over the few sequences of a 4 KiB stretch, instead of a binary search over all limits
	std::vector<uint32_t> coarse[2];
	int64_t max_query_len = 0;                 // longest sequence of the query block (+ delimiter), for the width of the hit sort's key
	uint64_t max_query_len_generation = ~(uint64_t)0;
	enum { COARSE_SHIFT = 12 };
	dmnd::DevBuf d_limits[2];
	// banded-swipe work buffers
	dmnd::DevBuf items, order, p_of_slot, trace_off, transcript_off, ends, hsps, trace, transcript, status;
	dmnd::DevBuf pairs, trace_off_item;        // packed-int16 sweep: item pairs per wavefront, trace offset by item index
	std::vector<int32_t> h_pairs;              // their host staging (outlive the asynchronous copies of a call)
	std::vector<int64_t> h_trace_off_item;
	dmnd::PinBuf up_stage[2];                  // dmnd_upload_block: page-locked double buffer of a pageable source
	hipEvent_t up_ev[2] = { nullptr, nullptr };
	bool up_busy[2] = { false, false };
	// a second transfer lane for the reference block alone (own stream, own staging): a driver may upload the reference block on a
	// helper thread while the query block of the same context is uploaded and masked on its main stream
	hipStream_t t_stream = nullptr;
	dmnd::PinBuf t_stage[2];
	hipEvent_t t_ev[2] = { nullptr, nullptr };
	bool t_busy[2] = { false, false };
	dmnd::PinBuf stage_h, ends_h;              // dmnd_swipe_keep: all launch arrays of a sweep in one upload; its results
	dmnd::DevBuf stage_d;
	std::vector<KeptTrace>* kts = nullptr;     // kept traces of the extension stage's ranking iterations (reused from call to call)
	dmnd::DevBuf host_q, host_t, host_cbs;      // staging for dmnd_banded_swipe_host
	double swipe_ms = 0.0, traceback_ms = 0.0;
	size_t trace_arena_max = (size_t)8 << 30;
	// seed-stage buffers (seed_api.hip)
	dmnd::DevBuf qid_of, mask_time, seed_keys, seed_next, seed_qlist, seed_qkeys, seed_slot2, seed_loc2, seed_survivors, seed_scored, seed_need, seed_qfold, seed_tfold, seed_tcodes, seed_tflags, seed_tplanes, seed_tclass, matched_slot, matched_loc, counters, seed_hits, seed_bitmap, seed_deferred, seed_eslot, seed_eloc, seed_hits_sorted, sort_keys[2], sort_idx[2];
	void* sort_tmp = nullptr; size_t sort_tmp_bytes = 0;      // rocPRIM radix sort scratch
	dmnd::DevBuf join_keep, join_pos, join_in, join_out, join_recv;      // dmnd_join_blocks_device: survivor flags and their numbers; staging of the host form
	int64_t n_seed_hits = 0;
	// gapped filter (gapped_api.hip)
	dmnd::DevBuf gf_tables, gf_hits, gf_flags, gf_scores, gf_units;
	double gapped_filter_evalue = 0.0, gf_ms = 0.0;
	// tantan masking (mask_api.hip)
	dmnd::DevBuf mask_lr, mask_pb, mask_scale, mask_pos, mask_ids, mask_soff, mask_long_ids, mask_long_soff, mask_long_pb, mask_long_scale;
	// motif soft masking (dmnd_soft_mask_block): a copy of each block with the motif stretches masked, read by the seed
	// stage for seed generation only; soft_valid is dropped whenever the block changes
	dmnd::DevBuf soft[2], motif_hit, motif_table;
	bool soft_valid[2] = { false, false };
	std::vector<uint64_t> motifs;              // dmnd_set_context_motif_table: this context's motif table instead of the process-wide one
	bool own_motifs = false;
	double mask_ms = 0.0;
	double seed_ms[5] = { 0, 0, 0, 0, 0 };
	// extension-stage statistics of the last dmnd_extend (extend_host.hip)
	std::vector<dmnd::DevBuf> keep_trace;      // trace arenas of dmnd_swipe_keep (one per ranking-chunk iteration of a pass)
	std::vector<dmnd_ctx*> aux;                // auxiliary contexts (own stream + work buffers) for concurrent sub-batches of dmnd_extend
	int8_t* pinned_cbs = nullptr; size_t pinned_cbs_cap = 0;      // Hauser bias of the query block, pinned host copy (parallel to the block letters)
	double ext_stats[12] = { 0 };
	double ext_plan_stats[3] = { 0, 0, 0 };    // device planner of the last dmnd_extend: groups, groups left to the host, bands (0: the host planned)
	double host_ms[3] = { 0, 0, 0 };           // host wall time inside dmnd_banded_swipe: prepare, launch+wait, unpack (DMND_TRACE)
	int comp_based_stats = 1;                  // config.comp_based_stats: 1 = Hauser bias (default), 0 = off
	int query_contexts = 1;                    // align_mode.query_contexts: 6 for blastx (basic/basic.cpp:40-60)
	int max_target_seqs = 25;                  // config.max_target_seqs (-k), basic/config.h:55
	int frame_shift = 0;                       // config.frame_shift (-F): > 0 = frameshift alignment through the legacy pipeline (frameshift_host.hip)
	bool range_culling = false;                // config.query_range_culling
	double range_cover = 50.0;                 // config.query_range_cover
	int fs_channels = 16;                      // int16 channels of the reference's three-frame score-only vectors (AVX2)
	int max_hsps = 1;                          // config.max_hsps (--max-hsps): HSPs reported per target, 0 = all (dmnd_set_max_hsps)
	int global_ranking = 0;                    // config.global_ranking_targets (--global-ranking): dmnd_set_global_ranking
	dmnd::DevBuf alt_targets;                  // masked target copies of the alternative-HSP rounds (extend_host.hip)
	double top_percent = -1.0;                 // config.toppercent (--top); < 0 = off
	double min_id = 0, query_cover = 0, subject_cover = 0, min_bit_score = 0;      // --id, --query-cover, --subject-cover, --min-score
	dmnd_same_title_fn same_title = nullptr;   // --no-self-hits: title comparison of the caller (dmnd_set_no_self_hits)
	void* same_title_user = nullptr;
	bool reuse_query_index = false;            // dmnd_set_query_index_reuse
	uint64_t query_generation = 0;             // bumped whenever the query block or its masks change
	uint64_t cbs_generation = ~(uint64_t)0;    // query_generation the Hauser bias in `cbs` was computed for by dmnd_extend (all sequences); ~0: none / someone else's
	std::string qindex_signature;              // what the resident query seed index was built for (empty: nothing resident)
	std::vector<int32_t> source_lens;          // translated queries: DNA read lengths of the query block (query cover)
	int ext_mode = -1;                         // --ext: DMND_EXT_DEFAULT = what the sensitivity selects
	int band_mode_fast = 1;                    // Extension::Mode::BANDED_FAST up to --sensitive, BANDED_SLOW from --more-sensitive up (align/extend.cpp:62-75)
	std::vector<unsigned long long> seed_trace;   // DMND_TRACE: per shape Hamming survivors and deferred pairs of the last seed search
	double ranking_block_letters = 2e9;        // default_letters of ranking_chunk_size (align/extend.cpp:87): 8e8 from --very-sensitive up
};

// internal (not part of the C ABI)
// banded swipe on the work buffers / stream of `work` over the blocks, bias and scoring matrix resident in `blocks`
// Round-1 sweep that keeps its trace (api.hip): the extension stage's second round re-runs the same DpTargets with traceback
// (gapped_final.cpp), so the first sweep already writes the trace rows and the end cells of every target whose matrix fits the
// traceback path, and round 2 only walks the kept traces of the targets that survive culling (dmnd_traceback_kept).
// A KeptTrace describes one such sweep: per item the offset of its trace in arena `arena` of the work context, its band class
// and its end cell. kept = false: the arena would exceed the context's trace budget, the call ran score-only instead.
struct KeptTrace {
	int arena = -1;
	bool kept = false;
	std::vector<int64_t> trace_off;
	std::vector<int32_t> P;
	std::vector<int32_t> score, end_i, end_j;
};
int dmnd_swipe_keep(dmnd_ctx* work, const dmnd_ctx* blocks, const dmnd_dp_target* items, int64_t n, int arena, dmnd_hsp* out, KeptTrace& kt);
// items[k] with its KeptTrace entry src[k] (index into kt's vectors): statistics and coordinates from the kept trace, no transcripts
int dmnd_traceback_kept(dmnd_ctx* work, const dmnd_ctx* blocks, const dmnd_dp_target* items, const KeptTrace& kt, const int64_t* src, int64_t n, dmnd_hsp* out);
namespace dmnd { struct SwipeEnd; }
namespace dmnd { int64_t sweep_rows_min_items(); }      // items of a call / ranking iteration from which on the row classes of the packed 16-bit sweeps are used (api.hip)
int dmnd_sweep_classes(dmnd_ctx* work, const dmnd_ctx* blocks, const dmnd_dp_target* d_items, const uint32_t* class_count, const uint32_t* class_max_steps, int n_classes,
	const int32_t* order_dev, const int64_t* off_slot_dev, const int32_t* pairs_dev, const int64_t* off_item_dev, uint8_t* trace_dev, dmnd::SwipeEnd* ends_dev);
int dmnd_swipe_shared(dmnd_ctx* work, const dmnd_ctx* blocks, const dmnd_dp_target* items, int64_t n, int mode, uint32_t hsp_values,
	dmnd_hsp* out, uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used);
int dmnd_swipe_targets(dmnd_ctx* work, const dmnd_ctx* blocks, const int8_t* t, int64_t t_len, const dmnd_dp_target* items, int64_t n, int mode, uint32_t hsp_values,
	dmnd_hsp* out, uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used);
// gapped_api.hip: dmnd_gapped_filter on hits that may be in HBM already (hits_dev) and whose flags may stay there (flags == NULL)
int dmnd_gapped_filter_on(dmnd_ctx* c, const dmnd_seed_hit* hits, const dmnd_seed_hit* hits_dev, int64_t n_hits, int use_cbs_flag, uint8_t* flags, int32_t* scores);
// frameshift_host.hip: the extension stage of blastx -F for the seed hits of a block pair (sorted by query)
int dmnd_extend_frameshift(dmnd_ctx* c, const int8_t* qdata, const int8_t* tdata, const dmnd_seed_hit* hits, int64_t n_hits, int threads,
	std::vector<dmnd_match>& out, uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used);
// the first n_keep adjusted matrices of c stay, n more are appended (dmnd_upload_matrices = the same with n_keep = 0)
int dmnd_append_matrices(dmnd_ctx* c, int64_t n_keep, const int8_t* matrices, int64_t n);
// k-th of `split` auxiliary contexts of c (created on first use; owned and destroyed by c)
dmnd_ctx* aux_context(dmnd_ctx* c, int k, int split);
