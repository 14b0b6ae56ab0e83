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
//                        instantiated here over fixed arrays in LDS (two workspace sizes, see plan_chain_kernel)
//   add_dp_targets       band = [d_min - b, d_max + 1 + b) clipped to the matrix, overlapping bands merged (align/gapped_score.cpp:107-180)
// Kernels (all one thread per hit or per group; the work per group is a few hundred scalar operations, the launch is latency-bound):
//   plan_mark_kernel     target of every hit (binary search in the block's limits), group / query head flags, order check
//   rocPRIM inclusive scan of the packed head flags  -> group and query numbers
//   plan_fill_kernel     group and query records
//   plan_segments_kernel per group: score, filter flag, sorted segments; single-segment groups are finished here
//   plan_chain_list_kernel, plan_chain_kernel<small> / <large>    per multi-segment group: chaining + band merge
//   rocPRIM exclusive scan of the band counts, plan_gather_kernel: dense band list
// A group with more than PLAN_MAX_HITS hits or PLAN_MAX_SEGS segments, or whose chaining outgrows the fixed arrays, is marked
// PLAN_ON_HOST and planned by the host as before (extend_host.hip plan_groups): same result either way.
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

// per-lane arrays of plan_segments_kernel, in LDS (entry [x][lane]: a lane's accesses never conflict with its neighbours'). In
// private memory they were 528 bytes of scratch per lane behind 240 registers -- and the first kernel of a process that needs
// scratch makes the runtime set the scratch arena of its queue up, milliseconds on the first dmnd_extend of a run.
// The segments take the hits' places: segment ns is written when hit x >= ns has been read, (i, j, source) -> (i, j, length), + score.
struct SegLds {
	int hi[PLAN_MAX_HITS][64], hj[PLAN_MAX_HITS][64], hs[PLAN_MAX_HITS][64], ss[PLAN_MAX_HITS][64];
};

__global__ __launch_bounds__(64) void plan_segments_kernel(PlanArgs a)
{
	__shared__ SegLds L;
	const int lane = threadIdx.x;
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
	for (uint32_t x = 0; x < n; ++x) {
		const dmnd_seed_hit h = a.hits[b + x];
		const int i = h.seed_offset, j = (int)(h.subject - t0), d = i - j;
		int p = (int)x;
		while (p > 0 && (L.hi[p - 1][lane] - L.hj[p - 1][lane] > d || (L.hi[p - 1][lane] - L.hj[p - 1][lane] == d && L.hj[p - 1][lane] > j))) {
			L.hi[p][lane] = L.hi[p - 1][lane]; L.hj[p][lane] = L.hj[p - 1][lane]; L.hs[p][lane] = L.hs[p - 1][lane]; --p;
		}
		L.hi[p][lane] = i; L.hj[p][lane] = j; L.hs[p][lane] = (int)(b + x);
	}
	// a hit inside the last kept segment of its diagonal is skipped; segments need a positive score (ungapped.cpp:96-113)
	int ns = 0;
	for (uint32_t x = 0; x < n; ++x) {
		const int i = L.hi[x][lane], j = L.hj[x][lane];
		if (ns > 0 && L.hi[ns - 1][lane] - L.hj[ns - 1][lane] == i - j && L.hj[ns - 1][lane] + L.hs[ns - 1][lane] >= j) continue;
		const XdropSeg xs = a.xd[L.hs[x][lane]];
		if (xs.score > 0) { L.hi[ns][lane] = i - xs.left; L.hj[ns][lane] = j - xs.left; L.hs[ns][lane] = xs.left + xs.right; L.ss[ns][lane] = xs.score; ++ns; }
	}
	if (ns == 0) { a.groups[g] = grp; return; }
	if (ns == 1) {
		const int d = L.hi[0][lane] - L.hj[0][lane];
		const Chain c{ d, d, L.ss[0][lane], 0, 0, 0, 0 };
		BandOut out{ a.band_slots + b, (int)n, 0, false };
		merge_bands(&c, 1, band_for_dev(qlen, a.band_fast != 0), qlen, tlen, out);
		grp.n_bands = (uint8_t)out.n;
		a.groups[g] = grp;
		return;
	}
	if (ns > PLAN_MAX_SEGS) { grp.n_bands = PLAN_ON_HOST; a.groups[g] = grp; return; }
	// stable by (diagonal, segment start): the x-drop walk to the left may carry a later hit's segment in front of an earlier one's
	for (int x = 1; x < ns; ++x) {
		const int vi = L.hi[x][lane], vj = L.hj[x][lane], vl = L.hs[x][lane], vs = L.ss[x][lane];
		int p = x;
		while (p > 0 && (L.hi[p - 1][lane] - L.hj[p - 1][lane] > vi - vj || (L.hi[p - 1][lane] - L.hj[p - 1][lane] == vi - vj && L.hj[p - 1][lane] > vj))) {
			L.hi[p][lane] = L.hi[p - 1][lane]; L.hj[p][lane] = L.hj[p - 1][lane]; L.hs[p][lane] = L.hs[p - 1][lane]; L.ss[p][lane] = L.ss[p - 1][lane]; --p;
		}
		L.hi[p][lane] = vi; L.hj[p][lane] = vj; L.hs[p][lane] = vl; L.ss[p][lane] = vs;
	}
	int32_t* dst = a.segs + 4 * (size_t)b;
	for (int x = 0; x < ns; ++x) { dst[4 * x] = L.hi[x][lane]; dst[4 * x + 1] = L.hj[x][lane]; dst[4 * x + 2] = L.hs[x][lane]; dst[4 * x + 3] = L.ss[x][lane]; }
	grp.n_bands = PLAN_NEED_CHAIN;
	grp.band_begin = (uint32_t)ns;
	a.groups[g] = grp;
}

// Everything a lane's chaining works on, in LDS: in private memory this was 6 KB of scratch per lane (one lane per group, 64 groups a
// wavefront, most of them with nothing to chain). plan_chain_list_kernel lists the groups that need chaining -- those with at most
// PLAN_SMALL_SEGS segments from the front of the list, the others from its back -- and two instantiations of plan_chain_kernel take
// them: a 1 KB workspace for the small ones (most: two or three segments; as many lanes of a workgroup as 48 KB hold), the
// 6 KB workspace for the rest and for a small one whose links did not fit (it is appended to the other list). A call with few
// hits (PlanArgs::small_segs = 0) uses the large form only: each kernel's time is the one longest chain in it (~0.1 ms: the junction
// search reads letters one dependent load at a time), and two kernels would pay that twice.
enum { PLAN_SMALL_SEGS = 4 };
template<int NODES, int LINKS, int CHAINS>
struct ChainLdsT {
	ChainWorkspaceT<FixedChainPolicy, NODES, LINKS, CHAINS> ws;
	Seg sg[NODES];
	FixedVec<Chain, CHAINS> chains;
};
template<int NODES, int LINKS, int CHAINS>
struct ChainLanes { enum { bytes = (int)sizeof(ChainLdsT<NODES, LINKS, CHAINS>), fit = 48 * 1024 / bytes, value = fit > 64 ? 64 : fit }; };

__global__ __launch_bounds__(256) void plan_chain_list_kernel(PlanArgs a)
{
	__shared__ uint32_t n_small, n_big, base_small, base_big;
	if (threadIdx.x == 0) { n_small = 0; n_big = 0; }
	__syncthreads();
	const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
	const bool in = g < a.counters->n_groups;
	const PlanGroup grp = in ? a.groups[g] : PlanGroup{ 0, 0, 0, 0, 0, 0, 0 };
	const bool need = in && grp.n_bands == PLAN_NEED_CHAIN, small = need && (int)grp.band_begin <= a.small_segs, big = need && !small;
	// places inside the workgroup from LDS counters, one global atomic per workgroup and list (one per wavefront cost 0.2 ms at 10^6 groups)
	uint32_t at = 0;
	if (small) at = atomicAdd(&n_small, 1u);
	if (big) at = atomicAdd(&n_big, 1u);
	__syncthreads();
	if (threadIdx.x == 0) {
		if (n_small) base_small = atomicAdd(&a.counters->n_chain, n_small);
		if (n_big) base_big = atomicAdd(&a.counters->n_chain_big, n_big);
	}
	__syncthreads();
	if (small) a.chain_list[base_small + at] = g;
	if (big) a.chain_list[a.chain_cap - 1 - (base_big + at)] = g;
}

template<int NODES, int LINKS, int CHAINS, bool SMALL>
__global__ __launch_bounds__(64) void plan_chain_kernel(PlanArgs a)
{
	typedef ChainLdsT<NODES, LINKS, CHAINS> Lds;
	constexpr int LANES = ChainLanes<NODES, LINKS, CHAINS>::value;
	__shared__ ScoreTable S;
	__shared__ __align__(16) char raw[LANES * sizeof(Lds)];
	for (int x = threadIdx.x; x < 32 * 32; x += blockDim.x) S.m[x] = a.matrix[x];
	if (threadIdx.x == 0) { S.gap_open = a.gap_open; S.gap_extend = a.gap_extend; }
	__syncthreads();
	if ((int)threadIdx.x >= LANES) return;
	const uint32_t k = blockIdx.x * LANES + threadIdx.x;
	if (k >= (SMALL ? a.counters->n_chain : a.counters->n_chain_big)) return;
	const uint32_t g = SMALL ? a.chain_list[k] : a.chain_list[a.chain_cap - 1 - k];
	Lds& L = *reinterpret_cast<Lds*>(raw + threadIdx.x * sizeof(Lds));
	PlanGroup grp = a.groups[g];
	const int ns = (int)grp.band_begin;
	const uint32_t b = grp.hit_begin;
	const int64_t t0 = a.tlimits[grp.target];
	const int tlen = (int)(a.tlimits[grp.target + 1] - t0 - 1);
	const uint32_t query = a.hits[b].query;
	const int64_t q0 = a.qlimits[query];
	const int qlen = (int)(a.qlimits[query + 1] - q0 - 1);
	const int32_t* src = a.segs + 4 * (size_t)b;
	for (int x = 0; x < ns; ++x) L.sg[x] = Seg{ src[4 * x], src[4 * x + 1], src[4 * x + 2], src[4 * x + 3] };
	L.ws.reset_fixed();
	L.chains.reset();
	L.ws.run_segs(S, SeqRef{ a.qblock + q0, qlen }, SeqRef{ a.tblock + t0, tlen }, L.sg, (size_t)ns, L.chains);
	if (L.ws.overflowed() || L.chains.overflow) {
		if (SMALL) {      // once more with the large workspace (the group keeps its PLAN_NEED_CHAIN state)
			a.chain_list[a.chain_cap - 1 - atomicAdd(&a.counters->n_chain_big, 1u)] = g;
			return;
		}
		grp.band_begin = 0; grp.n_bands = PLAN_ON_HOST; a.groups[g] = grp;
		return;
	}
	grp.band_begin = 0;
	insertion_sort(L.chains.begin(), L.chains.end(), [](const Chain& x, const Chain& y) { return x.d_min < y.d_min; });      // std::stable_sort by d_min
	BandOut out{ a.band_slots + b, (int)grp.n_hits, 0, false };
	merge_bands(L.chains, (int)L.chains.size(), band_for_dev(qlen, a.band_fast != 0), qlen, tlen, out);
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

void plan_kernel_table(const void** out, int* n)
{
	const void* k[] = { (const void*)plan_mark_kernel, (const void*)plan_fill_kernel, (const void*)plan_segments_kernel, (const void*)plan_chain_list_kernel,
		(const void*)plan_chain_kernel<PLAN_SMALL_SEGS, 16, 4, true>, (const void*)plan_chain_kernel<PLAN_MAX_SEGS, 96, 16, false>, (const void*)plan_count_kernel, (const void*)plan_gather_kernel };
	*n = (int)(sizeof(k) / sizeof(k[0]));
	for (int i = 0; i < *n; ++i) out[i] = k[i];
}

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
	// (the chaining kernels are launched over the upper bound of one group per two hits -- a group that needs chaining has at least
	// two -- and read the counts of their lists on the device)
	hipLaunchKernelGGL(plan_chain_list_kernel, dim3(b256), dim3(256), 0, st, a);
	{
		constexpr int LS = ChainLanes<PLAN_SMALL_SEGS, 16, 4>::value, LB = ChainLanes<PLAN_MAX_SEGS, 96, 16>::value;
		if (a.small_segs > 0) hipLaunchKernelGGL((plan_chain_kernel<PLAN_SMALL_SEGS, 16, 4, true>), dim3((unsigned)((n / 2 + LS) / LS)), dim3(64), 0, st, a);
		hipLaunchKernelGGL((plan_chain_kernel<PLAN_MAX_SEGS, 96, 16, false>), dim3((unsigned)((n / 2 + LB) / LB)), dim3(64), 0, st, a);
	}
	hipLaunchKernelGGL(plan_count_kernel, dim3(b256g), dim3(256), 0, st, a);
	// (the scan runs over n + 1 entries whatever the group count: entries beyond it are never read)
	e = rocprim::exclusive_scan(*a.scan_tmp, need2, a.band_count, a.band_off, 0u, n + 1, rocprim::plus<uint32_t>(), st);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(plan_gather_kernel, dim3(b256g), dim3(256), 0, st, a);
	return hipGetLastError();
}

}  // namespace dmnd

// dmnd_init: the first launch of a kernel of this translation unit loads its code object onto the device; and the first launch of
// EVERY kernel costs a function look-up of its own (measured, round 6: the planner's first call spent 6.4 ms of host time launching its
// nine kernels and two scans against 0.3 ms later) -- asked for here, on dmnd_init's side thread, with hipFuncGetAttributes and a
// one-element scan
namespace { __global__ void touch_plan_kernel() {} }
extern "C" hipError_t dmnd_touch_plan(hipStream_t st)
{
	hipLaunchKernelGGL(touch_plan_kernel, dim3(1), dim3(64), 0, st);
	hipError_t e = hipGetLastError();
	hipFuncAttributes attr;
	const void* kernels[16];
	int n_kernels = 0;
	dmnd::plan_kernel_table(kernels, &n_kernels);
	for (int i = 0; i < n_kernels; ++i) if (e == hipSuccess) e = hipFuncGetAttributes(&attr, kernels[i]);
	if (e != hipSuccess) return e;
	void* buf = nullptr;
	if (hipMalloc(&buf, 4096) != hipSuccess) return hipGetLastError();
	uint64_t* a = static_cast<uint64_t*>(buf);
	uint32_t* b = reinterpret_cast<uint32_t*>(a + 16);
	size_t need = 0, need2 = 0;
	(void)rocprim::inclusive_scan(nullptr, need, a, a + 4, (size_t)2, rocprim::plus<uint64_t>(), st);
	(void)rocprim::exclusive_scan(nullptr, need2, b, b + 4, 0u, (size_t)2, rocprim::plus<uint32_t>(), st);
	void* tmp = nullptr;
	if (hipMalloc(&tmp, (need > need2 ? need : need2) + 256) == hipSuccess) {
		(void)hipMemsetAsync(buf, 0, 4096, st);
		(void)rocprim::inclusive_scan(tmp, need, a, a + 4, (size_t)2, rocprim::plus<uint64_t>(), st);
		(void)rocprim::exclusive_scan(tmp, need2, b, b + 4, 0u, (size_t)2, rocprim::plus<uint32_t>(), st);
		(void)hipStreamSynchronize(st);
		(void)hipFree(tmp);
	}
	(void)hipFree(buf);
	return hipGetLastError();
}
