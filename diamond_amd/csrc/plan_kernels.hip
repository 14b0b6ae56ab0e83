// plan_kernels.hip -- device planner of the extension stage (round 6; SURVEY.md 8 rows a10, a12-a14 moved off the host).
//
// Input: the seed hits of a block pair in HBM, sorted by (query, subject location, seed offset) as dmnd_seed_search leaves them,
// the x-drop extension of every hit (xdrop_seg_kernel) and, with the gapped filter on, its flag per hit. Output: per (query, target)
// group the bands [d_begin, d_end) of its round-1 DpTargets. Reference behaviour replaced, per group:
//   load_hits            target of a hit = the sequence that holds its location; groups in ascending target order, their best
//                        stage-1 score (/root/reference/src/align/load_hits.h:44-127)
//   ungapped_stage       hits sorted by (diagonal, j); a hit inside the previous segment of its diagonal is skipped; segments
//                        with a positive x-drop score are kept (align/ungapped.cpp:62-126)
//   Chaining::run        one segment = one chain; more: the segment graph of chain_graph.h (chaining/greedy_align.cpp:362-497),
//                        instantiated here over fixed arrays in the lane's private memory
//   add_dp_targets       band = [d_min - b, d_max + 1 + b) clipped to the matrix, overlapping bands merged (align/gapped_score.cpp:107-180)
// Kernels (all one thread per hit or per group; the work per group is a few hundred scalar operations, the launch is latency-bound):
//   plan_mark_kernel     target of every hit (binary search in the block's limits), group / query head flags, order check
//   rocPRIM inclusive scan of the packed head flags  -> group and query numbers
//   plan_fill_kernel     group and query records
//   plan_segments_kernel per group: score, filter flag, sorted segments; single-segment groups are finished here
//   plan_chain_kernel    per multi-segment group: chaining + band merge
//   rocPRIM exclusive scan of the band counts, plan_gather_kernel: dense band list
// A group with more than PLAN_MAX_HITS hits or PLAN_MAX_SEGS segments, or whose chaining outgrows the fixed arrays, is marked
// PLAN_ON_HOST and planned by the host as before (extend_host.hip plan_one_group): same result either way.
// Compiled with -ffp-contract=off: the chaining truncates double expressions to int (joined_score, faded) exactly as the host does.
#include <hip/hip_runtime.h>
#include <climits>
#include <rocprim/device/device_scan.hpp>
#include "plan_kernels.h"
#include "chain_graph.h"

namespace dmnd {

namespace {

__device__ inline int band_for_dev(int len, bool fast)           // Extension::band, gapped_score.cpp:41-73 (extend_host.hip band_for)
{
	if (fast) return len < 50 ? 12 : len < 100 ? 16 : len < 250 ? 30 : len < 350 ? 40 : 64;
	return len < 50 ? 15 : len < 100 ? 20 : len < 150 ? 30 : len < 200 ? 50 : len < 250 ? 60 : len < 350 ? 100 : len < 500 ? 120 : 150;
}

__global__ __launch_bounds__(256) void plan_mark_kernel(PlanArgs a)
{
	const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= a.n_hits) return;
	const dmnd_seed_hit h = a.hits[k];
	int64_t lo = 0, hi = a.n_targets;                           // last sequence whose first letter is at or before the location
	while (lo + 1 < hi) { const int64_t mid = (lo + hi) >> 1; if (a.tlimits[mid] <= h.subject) lo = mid; else hi = mid; }
	a.tgt[k] = (uint32_t)lo;
	uint64_t gh = 1, qh = 1;
	if (k > 0) {
		const dmnd_seed_hit p = a.hits[k - 1];
		qh = p.query != h.query;
		gh = qh || p.subject < a.tlimits[lo];
		if (p.query > h.query || (!qh && (p.subject > h.subject || (p.subject == h.subject && p.seed_offset > h.seed_offset)))) a.counters->unsorted = 1;
	}
	a.heads[k] = gh | (qh << 32);
}

__global__ __launch_bounds__(256) void plan_fill_kernel(PlanArgs a)
{
	const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= a.n_hits) return;
	const uint64_t s = a.head_scan[k], f = a.heads[k];
	const uint32_t g = (uint32_t)s - 1, q = (uint32_t)(s >> 32) - 1;
	if (f & 1u) { a.groups[g].hit_begin = (uint32_t)k; a.groups[g].target = a.tgt[k]; }
	if (f >> 32) a.queries[q] = PlanQuery{ a.hits[k].query, g, (uint32_t)k };
	if (k == a.n_hits - 1) {
		a.counters->n_groups = g + 1; a.counters->n_queries = q + 1;
		a.groups[g + 1].hit_begin = (uint32_t)a.n_hits;
		a.queries[q + 1] = PlanQuery{ 0xffffffffu, g + 1, (uint32_t)a.n_hits };
	}
}

struct BandOut {
	PlanBand* slots; int cap; int n; bool overflow;
	__device__ void emit(int d0, int d1) { if (n < cap) slots[n] = PlanBand{ d0, d1 }; else overflow = true; ++n; }
};

// add_dp_targets (gapped_score.cpp:107-180) over chains sorted by d_min: overlapping bands are merged
template<typename CV>
__device__ inline void merge_bands(const CV& chains, int n_chains, int base_band, int qlen, int tlen, BandOut& out)
{
	int d0 = INT_MAX, d1 = INT_MIN;
	for (int x = 0; x < n_chains; ++x) {
		const Chain& c = chains[x];
		const int b0 = chain_max(c.d_min - base_band, -(tlen - 1)), b1 = chain_min(c.d_max + 1 + base_band, qlen);
		const int lo = chain_max(d0, b0), hi = chain_min(d1, b1);
		const double overlap = hi > lo ? hi - lo : 0;
		// (d1 - d0) wraps for the initial (INT_MAX, INT_MIN) pair exactly as in the reference: the first chain never merges
		const double wd = (double)(int)((unsigned)d1 - (unsigned)d0);
		if (overlap / wd > 0.0 || overlap / (b1 - b0) > 0.0) { d0 = chain_min(d0, b0); d1 = chain_max(d1, b1); }
		else {
			if (d0 != INT_MAX) out.emit(d0, d1);
			d0 = b0; d1 = b1;
		}
	}
	if (d0 != INT_MAX) out.emit(d0, d1);
}

__global__ __launch_bounds__(64) void plan_segments_kernel(PlanArgs a)
{
	const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= a.counters->n_groups) return;
	PlanGroup grp = a.groups[g];
	const uint32_t b = grp.hit_begin, e = a.groups[g + 1].hit_begin, n = e - b;
	int score = 0;
	bool pass = a.gf_flags == nullptr;
	for (uint32_t k = b; k < e; ++k) {
		score = chain_max(score, (int)(uint16_t)a.hits[k].score);
		if (a.gf_flags) pass |= a.gf_flags[k] != 0;
	}
	grp.n_hits = n; grp.score = (uint16_t)score; grp.pass = pass ? 1 : 0; grp.band_begin = 0; grp.n_bands = 0;
	if (!pass) { a.groups[g] = grp; return; }                        // dropped before chaining (extend.cpp:205-213)
	if (n > PLAN_MAX_HITS) { grp.n_bands = PLAN_ON_HOST; a.groups[g] = grp; return; }
	const int64_t t0 = a.tlimits[grp.target];
	const int tlen = (int)(a.tlimits[grp.target + 1] - t0 - 1);
	const uint32_t query = a.hits[b].query;
	const int qlen = (int)(a.qlimits[query + 1] - a.qlimits[query] - 1);
	// the group's hits sorted by (diagonal, j) -- they arrive sorted by (j, i)
	int hi[PLAN_MAX_HITS], hj[PLAN_MAX_HITS], hs[PLAN_MAX_HITS];
	for (uint32_t x = 0; x < n; ++x) {
		const dmnd_seed_hit h = a.hits[b + x];
		const int i = h.seed_offset, j = (int)(h.subject - t0), d = i - j;
		int p = (int)x;
		while (p > 0 && (hi[p - 1] - hj[p - 1] > d || (hi[p - 1] - hj[p - 1] == d && hj[p - 1] > j))) { hi[p] = hi[p - 1]; hj[p] = hj[p - 1]; hs[p] = hs[p - 1]; --p; }
		hi[p] = i; hj[p] = j; hs[p] = (int)(b + x);
	}
	// a hit inside the last kept segment of its diagonal is skipped; segments need a positive score (ungapped.cpp:96-113)
	Seg sg[PLAN_MAX_HITS];
	int ns = 0;
	for (uint32_t x = 0; x < n; ++x) {
		if (ns > 0 && sg[ns - 1].diag() == hi[x] - hj[x] && sg[ns - 1].j_end() >= hj[x]) continue;
		const XdropSeg xs = a.xd[hs[x]];
		if (xs.score > 0) sg[ns++] = Seg{ hi[x] - xs.left, hj[x] - xs.left, xs.left + xs.right, xs.score };
	}
	if (ns == 0) { a.groups[g] = grp; return; }
	if (ns == 1) {
		const Chain c{ sg[0].diag(), sg[0].diag(), sg[0].score, 0, 0, 0, 0 };
		BandOut out{ a.band_slots + b, (int)n, 0, false };
		merge_bands(&c, 1, band_for_dev(qlen, a.band_fast != 0), qlen, tlen, out);
		grp.n_bands = (uint8_t)out.n;
		a.groups[g] = grp;
		return;
	}
	if (ns > PLAN_MAX_SEGS) { grp.n_bands = PLAN_ON_HOST; a.groups[g] = grp; return; }
	// stable by (diagonal, segment start): the x-drop walk to the left may carry a later hit's segment in front of an earlier one's
	for (int x = 1; x < ns; ++x) {
		const Seg v = sg[x];
		int p = x;
		while (p > 0 && (sg[p - 1].diag() > v.diag() || (sg[p - 1].diag() == v.diag() && sg[p - 1].j > v.j))) { sg[p] = sg[p - 1]; --p; }
		sg[p] = v;
	}
	int32_t* dst = a.segs + 4 * (size_t)b;
	for (int x = 0; x < ns; ++x) { dst[4 * x] = sg[x].i; dst[4 * x + 1] = sg[x].j; dst[4 * x + 2] = sg[x].len; dst[4 * x + 3] = sg[x].score; }
	grp.n_bands = PLAN_NEED_CHAIN;
	grp.band_begin = (uint32_t)ns;
	a.groups[g] = grp;
}

typedef ChainWorkspaceT<FixedChainPolicy, PLAN_MAX_SEGS, 96, 16> DevChain;

__global__ __launch_bounds__(64) void plan_chain_kernel(PlanArgs a)
{
	__shared__ ScoreTable S;
	for (int x = threadIdx.x; x < 32 * 32; x += blockDim.x) S.m[x] = a.matrix[x];
	if (threadIdx.x == 0) { S.gap_open = a.gap_open; S.gap_extend = a.gap_extend; }
	__syncthreads();
	const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= a.counters->n_groups) return;
	PlanGroup grp = a.groups[g];
	if (grp.n_bands != PLAN_NEED_CHAIN) return;
	const int ns = (int)grp.band_begin;
	const uint32_t b = grp.hit_begin;
	const int64_t t0 = a.tlimits[grp.target];
	const int tlen = (int)(a.tlimits[grp.target + 1] - t0 - 1);
	const uint32_t query = a.hits[b].query;
	const int64_t q0 = a.qlimits[query];
	const int qlen = (int)(a.qlimits[query + 1] - q0 - 1);
	Seg sg[PLAN_MAX_SEGS];
	const int32_t* src = a.segs + 4 * (size_t)b;
	for (int x = 0; x < ns; ++x) sg[x] = Seg{ src[4 * x], src[4 * x + 1], src[4 * x + 2], src[4 * x + 3] };
	DevChain ws;
	FixedVec<Chain, 16> chains;
	ws.run_segs(S, SeqRef{ a.qblock + q0, qlen }, SeqRef{ a.tblock + t0, tlen }, sg, (size_t)ns, chains);
	grp.band_begin = 0;
	if (ws.overflowed() || chains.overflow) { grp.n_bands = PLAN_ON_HOST; a.groups[g] = grp; return; }
	insertion_sort(chains.begin(), chains.end(), [](const Chain& x, const Chain& y) { return x.d_min < y.d_min; });      // std::stable_sort by d_min
	BandOut out{ a.band_slots + b, (int)grp.n_hits, 0, false };
	merge_bands(chains, (int)chains.size(), band_for_dev(qlen, a.band_fast != 0), qlen, tlen, out);
	grp.n_bands = out.overflow || out.n >= PLAN_NEED_CHAIN ? (uint8_t)PLAN_ON_HOST : (uint8_t)out.n;
	a.groups[g] = grp;
}

__global__ __launch_bounds__(256) void plan_count_kernel(PlanArgs a)
{
	const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t G = a.counters->n_groups;
	if (g > G) return;
	uint32_t n = 0;
	if (g < G) {
		const uint8_t nb = a.groups[g].n_bands;
		if (nb == PLAN_ON_HOST) atomicAdd(&a.counters->n_on_host, 1u); else n = nb;
	}
	a.band_count[g] = n;                                             // (entry G: the scan leaves the total there)
}

__global__ __launch_bounds__(256) void plan_gather_kernel(PlanArgs a)
{
	const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t G = a.counters->n_groups;
	if (g > G) return;
	if (g == G) { a.counters->n_bands = a.band_off[G]; return; }
	PlanGroup grp = a.groups[g];
	const uint32_t off = a.band_off[g];
	if (grp.n_bands != PLAN_ON_HOST)
		for (uint32_t k = 0; k < grp.n_bands; ++k) a.bands[off + k] = a.band_slots[grp.hit_begin + k];
	a.groups[g].band_begin = off;
}

hipError_t ensure_tmp(void** tmp, size_t* have, size_t need)
{
	if (need <= *have) return hipSuccess;
	if (*tmp) (void)hipFree(*tmp);
	*tmp = nullptr; *have = 0;
	const hipError_t e = hipMalloc(tmp, need);
	if (e == hipSuccess) *have = need;
	return e;
}

}  // namespace

hipError_t launch_plan(const PlanArgs& a, hipStream_t st)
{
	if (a.n_hits <= 0) return hipSuccess;
	const size_t n = (size_t)a.n_hits;
	hipError_t e = hipMemsetAsync(a.counters, 0, sizeof(PlanCounters), st);
	if (e != hipSuccess) return e;
	const unsigned b256 = (unsigned)((n + 255) / 256), b64 = (unsigned)((n + 63) / 64), b256g = (unsigned)((n + 1 + 255) / 256);
	hipLaunchKernelGGL(plan_mark_kernel, dim3(b256), dim3(256), 0, st, a);
	size_t need = 0, need2 = 0;
	e = rocprim::inclusive_scan(nullptr, need, a.heads, a.head_scan, n, rocprim::plus<uint64_t>(), st);
	if (e != hipSuccess) return e;
	e = rocprim::exclusive_scan(nullptr, need2, a.band_count, a.band_off, 0u, n + 1, rocprim::plus<uint32_t>(), st);
	if (e != hipSuccess) return e;
	e = ensure_tmp(a.scan_tmp, a.scan_tmp_bytes, need > need2 ? need : need2);
	if (e != hipSuccess) return e;
	e = rocprim::inclusive_scan(*a.scan_tmp, need, a.heads, a.head_scan, n, rocprim::plus<uint64_t>(), st);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(plan_fill_kernel, dim3(b256), dim3(256), 0, st, a);
	// groups <= hits: the per-group kernels are launched over the hit count and return beyond the group count (read on the device)
	hipLaunchKernelGGL(plan_segments_kernel, dim3(b64), dim3(64), 0, st, a);
	hipLaunchKernelGGL(plan_chain_kernel, dim3(b64), dim3(64), 0, st, a);
	hipLaunchKernelGGL(plan_count_kernel, dim3(b256g), dim3(256), 0, st, a);
	// (the scan runs over n + 1 entries whatever the group count: entries beyond it are never read)
	e = rocprim::exclusive_scan(*a.scan_tmp, need2, a.band_count, a.band_off, 0u, n + 1, rocprim::plus<uint32_t>(), st);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(plan_gather_kernel, dim3(b256g), dim3(256), 0, st, a);
	return hipGetLastError();
}

}  // namespace dmnd

// dmnd_init: the first launch of a kernel of this translation unit loads its code object onto the device
namespace { __global__ void touch_plan_kernel() {} }
extern "C" hipError_t dmnd_touch_plan(hipStream_t st) { hipLaunchKernelGGL(touch_plan_kernel, dim3(1), dim3(64), 0, st); return hipGetLastError(); }
