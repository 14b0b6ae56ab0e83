// extend_kernels.hip -- device half of the extension stage behind the planner (see extend_kernels.h; round 6, SURVEY.md 8 rows a14,
// a18 and the host preparation of a15 / a16 moved off the host for the queries whose targets fit one ranking chunk).
//
// Kernels, all latency-bound integer work on lists of 10^5 - 10^6 entries (a handful of bytes per entry, one pass each):
//   ext_mark_kernel      one wavefront per query: does the query run here? (at most chunk_size groups, none left to the host by the
//                        planner, every band something the traceback-mode sweeps take); item count per group
//   rocPRIM exclusive scan of the counts -> first item of every group
//   ext_items_kernel     one thread per group: the DpTargets of its bands, band class, sweep steps, trace bytes, sort key; class
//                        histogram and DP cells (block-aggregated atomics)
//   rocPRIM radix sort (15 key bits) -> launch order: band class ascending, longest items first (api.hip order_slots)
//   ext_slots_kernel, rocPRIM scan, ext_offsets_kernel: trace offset of every item, item pairs of the packed 16-bit launches
//   (the sweeps: swipe16_kernels.hip / swipe_kernels.hip, launched by the host from the class histogram)
//   ext_select_kernel    one wavefront per query: best HSP per target past the report cutoff, rank of every target by
//                        (e-value, score desc, target) by counting, the first k survive; ambiguity detection
//   rocPRIM scan of the survivors, ext_round2_kernel: the list the trace walk runs over
//   (traceback_kernel, swipe_kernels.hip)
//   ext_records_kernel   one wavefront per query: its match records in output order
// Compiled with -ffp-contract=off like plan_kernels.hip (the e-value below follows evalue.h operation by operation).
#include <hip/hip_runtime.h>
#include <cfloat>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include "extend_kernels.h"
#include "swipe_core.h"

namespace dmnd {

namespace {

__device__ inline double ext_normal_cdf(double x) { return 0.5 * erfc(-0.70710678118654752440 * x); }

// Evaluer::evalue (evalue.h:34-52) with the device's exp / erfc / sqrt
__device__ inline double ext_evalue(const ExtEvalue& p, int raw_score, int qlen, int slen)
{
	const double inv_sqrt_2pi = 1.0 / sqrt(2.0 * 3.1415926535897932384626433832795);
	const double y = (double)raw_score, len1 = (double)(unsigned)qlen, len2 = (double)(unsigned)slen;
	const double m_l = len2 - (p.a * y + p.b), n_l = len1 - (p.a * y + p.b);
	const double v = fmax(p.v_thr, p.alpha * y + p.beta), sv = sqrt(v);
	const double mF = sv == 0.0 ? 1e100 : m_l / sv, nF = sv == 0.0 ? 1e100 : n_l / sv;
	const double PmF = ext_normal_cdf(mF), PnF = ext_normal_cdf(nF);
	const double EmF = -inv_sqrt_2pi * exp(-0.5 * mF * mF), EnF = -inv_sqrt_2pi * exp(-0.5 * nF * nF);
	const double p1 = m_l * PmF - sv * EmF, p2 = n_l * PnF - sv * EnF;
	const double c = fmax(p.c_thr, p.sigma * y + p.tau);
	const double area = p1 * p2 + c * (PmF * PnF);
	return area * (p.K * exp(-p.lambda * y)) * p.db_letters / (double)slen;
}

// two device e-values whose order the host's own values could reverse (or a value that close to the cutoff)
__device__ inline bool ext_near(double x, double y) { return fabs(x - y) <= 1e-9 * fmax(fabs(x), fabs(y)); }

__device__ inline int64_t ext_cells(const dmnd_dp_target& d)       // DpTarget::cells of a banded target (dp/dp.h:47-52, 121-124)
{
	const int pos = imax(d.d_end - 1, 0) - (d.d_end - 1);
	const int j1 = imin(d.query_len - 1 - d.d_begin, d.target_len - 1) + 1;
	return (int64_t)(j1 - pos) * (int64_t)(d.d_end - d.d_begin);
}

__global__ __launch_bounds__(64) void ext_mark_kernel(ExtArgs a)
{
	const uint32_t q = blockIdx.x, lane = threadIdx.x;
	const uint32_t g0 = a.queries[q].group_begin, g1 = a.queries[q + 1].group_begin, ng = g1 - g0;
	const uint32_t query = a.queries[q].query;
	const int qlen = (int)(a.qlimits[query + 1] - a.qlimits[query] - 1);
	bool bad = ng > a.chunk_size || qlen <= 0;
	if (!bad)
		for (uint32_t g = g0 + lane; g < g1; g += 64) {
			const PlanGroup grp = a.groups[g];
			if (!grp.pass) continue;
			if (grp.n_bands == PLAN_ON_HOST) { bad = true; break; }
			const int tlen = (int)(a.tlimits[grp.target + 1] - a.tlimits[grp.target] - 1);
			if (tlen <= 0) { bad = true; break; }
			for (uint32_t k = 0; k < grp.n_bands; ++k) {
				const PlanBand b = a.bands[grp.band_begin + k];
				const int band = b.d_end - b.d_begin;
				const dmnd_dp_target d{ 0, 0, 0, qlen, tlen, b.d_begin, b.d_end };
				if (band <= 0 || band_class(band) > 32 || ext_cells(d) > a.max_swipe_dp) bad = true;
			}
		}
	const bool ok = __ballot(bad) == 0;
	for (uint32_t g = g0 + lane; g < g1; g += 64) {
		const PlanGroup grp = a.groups[g];
		a.gq[g] = ok ? q : 0xffffffffu;
		a.cnt[g] = ok && grp.pass ? grp.n_bands : 0u;
	}
	if (lane == 0) {
		a.qstate[q] = ok ? EXT_Q_DEVICE : EXT_Q_HOST;      // (no counter here: thousands of atomics on one address cost this kernel 70 us)
		if (q == 0) a.cnt[a.n_groups] = 0;
	}
}

__global__ __launch_bounds__(256) void ext_init_kernel(ExtArgs a)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i > a.n_bands) return;
	a.r2_tr[i] = 0;
	if (i == a.n_bands) return;
	a.idx[i] = i; a.keys[i] = 0x7fffu; a.rows[i] = 0;
}

__global__ __launch_bounds__(256) void ext_items_kernel(ExtArgs a)
{
	__shared__ uint32_t h_count[EXT_CLASSES], h_steps[EXT_CLASSES];
	__shared__ unsigned long long h_cells;
	if (threadIdx.x < EXT_CLASSES) { h_count[threadIdx.x] = 0; h_steps[threadIdx.x] = 0; }
	if (threadIdx.x == 0) h_cells = 0;
	__syncthreads();
	const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g < a.n_groups) {
		const uint32_t n = a.cnt[g];
		if (n) {
			const PlanGroup grp = a.groups[g];
			const uint32_t query = a.queries[a.gq[g]].query;
			const int64_t q0 = a.qlimits[query], t0 = a.tlimits[grp.target];
			const int qlen = (int)(a.qlimits[query + 1] - q0 - 1), tlen = (int)(a.tlimits[grp.target + 1] - t0 - 1);
			const uint32_t first = a.item_off[g];
			unsigned long long cells = 0;
			for (uint32_t k = 0; k < n; ++k) {
				const PlanBand b = a.bands[grp.band_begin + k];
				const dmnd_dp_target d{ q0, t0, a.use_cbs ? q0 : (int64_t)-1, qlen, tlen, b.d_begin, b.d_end };
				const uint32_t i = first + k;
				a.items[i] = d;
				a.item_group[i] = g;
				const int P = band_class(b.d_end - b.d_begin);
				const Geom geom = make_geom(qlen, tlen, b.d_begin, b.d_end);
				const int64_t steps = n_steps(geom);
				const int c = 31 - __clz(P);
				a.p_of_item[i] = P;
				a.rows[i] = trace_bytes(geom, P);
				const int64_t s16 = steps >> 4;
				a.keys[i] = ((uint32_t)c << 10) | (uint32_t)(1023 - (s16 < 1023 ? s16 : 1023));
				atomicAdd(&h_count[c], 1u);
				atomicMax(&h_steps[c], (uint32_t)(steps < 0x7fffffff ? steps : 0x7fffffff));
				cells += (unsigned long long)ext_cells(d);
			}
			atomicAdd(&h_cells, cells);
		}
		if (g == 0) a.ctr->n_items = a.item_off[a.n_groups];
	}
	__syncthreads();
	if (threadIdx.x < EXT_CLASSES && h_count[threadIdx.x]) {
		atomicAdd(&a.ctr->class_count[threadIdx.x], h_count[threadIdx.x]);
		atomicMax(&a.ctr->class_max_steps[threadIdx.x], h_steps[threadIdx.x]);
	}
	if (threadIdx.x == 0 && h_cells) atomicAdd(&a.ctr->cells1, h_cells);
}

__global__ __launch_bounds__(256) void ext_slots_kernel(ExtArgs a)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s > a.n_bands) return;
	a.rows_slot[s] = s < a.ctr->n_items ? a.rows[a.order[s]] : 0;
}

__global__ __launch_bounds__(256) void ext_offsets_kernel(ExtArgs a)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t n = a.ctr->n_items;
	if (s == 0) a.ctr->total_rows = (unsigned long long)a.off_slot[n];
	if (s >= n) return;
	const uint32_t item = a.order[s];
	a.off_item[item] = a.off_slot[s];
	// the packed 16-bit launches take their items in pairs of neighbours of the launch order; every class starts a new pair
	const uint32_t c = a.keys_sorted[s] >> 10;
	uint32_t s0 = 0, pair0 = 0;
	for (uint32_t x = 0; x < c; ++x) { s0 += a.ctr->class_count[x]; pair0 += (a.ctr->class_count[x] + 1) / 2; }
	const uint32_t count = a.ctr->class_count[c];
	a.pairs[2 * pair0 + (s - s0)] = (int32_t)item;
	if (s - s0 == count - 1 && (count & 1u)) a.pairs[2 * pair0 + count] = -1;
}

struct SelSlot { double ev; int score; uint32_t target; int tlen; int valid; };

__device__ inline bool sel_less(const SelSlot& x, const SelSlot& y)      // Target::comp_evalue (target.h:123-129)
{
	return x.ev < y.ev || (x.ev == y.ev && (x.score > y.score || (x.score == y.score && x.target < y.target)));
}

__global__ __launch_bounds__(64) void ext_select_kernel(ExtArgs a)
{
	extern __shared__ SelSlot sel[];
	const uint32_t q = blockIdx.x, lane = threadIdx.x;
	if (a.qstate[q] != EXT_Q_DEVICE) return;
	const uint32_t g0 = a.queries[q].group_begin, g1 = a.queries[q + 1].group_begin, ng = g1 - g0;
	const uint32_t query = a.queries[q].query;
	const int qlen = (int)(a.qlimits[query + 1] - a.qlimits[query] - 1);
	bool amb = false, sat = false;
	uint32_t n_valid_mine = 0;
	for (uint32_t gi = lane; gi < ng; gi += 64) {
		const uint32_t g = g0 + gi, n = a.cnt[g], first = a.item_off[g];
		const uint32_t target = a.groups[g].target;
		const int tlen = (int)(a.tlimits[target + 1] - a.tlimits[target] - 1);
		// best HSP of the target among its bands that pass the report cutoff (gapped_score.cpp:182-268; Target::add_hit + inner_culling:
		// the highest score, of equal ones the band that starts first)
		bool have = false;
		int best = 0; uint32_t bi = 0; double bev = 0;
		for (uint32_t k = 0; k < n; ++k) {
			const SwipeEnd e = a.ends[first + k];
			if (e.pad[0]) sat = true;
			if (e.score <= 0) continue;
			const double ev = ext_evalue(a.ev, e.score, qlen, tlen);
			if (ext_near(ev, a.ev.max_evalue)) amb = true;
			if (!(ev <= a.ev.max_evalue)) continue;
			if (!have || e.score > best || (e.score == best && a.items[first + k].d_begin < a.items[bi].d_begin)) { have = true; best = e.score; bi = first + k; bev = ev; }
		}
		sel[gi] = SelSlot{ bev, best, target, tlen, have ? 1 : 0 };
		a.cand_item[g] = bi;
		a.cand_ev[g] = bev;
		n_valid_mine += have ? 1u : 0u;
	}
	__syncthreads();
	uint32_t n_valid = n_valid_mine;
	for (int o = 32; o > 0; o >>= 1) n_valid += __shfl_xor(n_valid, o);
	uint32_t keep_mask_lo = 0;       // (bit gi / 64 of this lane's groups)
	for (uint32_t gi = lane, r = 0; gi < ng; gi += 64, ++r) {
		const SelSlot me = sel[gi];
		bool keep = me.valid != 0;
		if (keep && n_valid > (uint32_t)a.k) {
			// culling(targets) (culling.cpp:189-193): sort, first k. Rank by counting; a pair the host's e-values could order the other way
			// round makes the query ambiguous (equal inputs give equal values on both sides)
			uint32_t rank = 0;
			for (uint32_t gj = 0; gj < ng; ++gj) {
				const SelSlot o = sel[gj];
				if (!o.valid || gj == gi) continue;
				if (ext_near(o.ev, me.ev) && !(o.score == me.score && o.tlen == me.tlen)) amb = true;
				rank += sel_less(o, me) ? 1u : 0u;
			}
			keep = rank < (uint32_t)a.k;
		}
		if (keep && r < 32) keep_mask_lo |= 1u << r;
	}
	const bool any_amb = __ballot(amb) != 0, any_sat = __ballot(sat) != 0;
	for (uint32_t gi = lane, r = 0; gi < ng; gi += 64, ++r)
		a.kept[g0 + gi] = !any_amb && !any_sat && ((keep_mask_lo >> r) & 1u) ? 1u : 0u;
	if (lane == 0) {
		if (any_amb || any_sat) a.qstate[q] = EXT_Q_AMBIGUOUS;
		if (any_amb) atomicAdd(&a.ctr->n_ambiguous, 1u);
		if (any_sat) atomicAdd(&a.ctr->n_saturated, 1u);
	}
}

__global__ __launch_bounds__(256) void ext_round2_kernel(ExtArgs a)
{
	__shared__ unsigned long long h_cells;
	if (threadIdx.x == 0) h_cells = 0;
	__syncthreads();
	const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g < a.n_groups) {
		if (g == 0) a.ctr->n_kept = a.kept_pos[a.n_groups];
		if (a.kept[g]) {
			const uint32_t k = a.kept_pos[g], item = a.cand_item[g];
			a.r2_order[k] = (int32_t)item;
			a.r2_p[k] = a.p_of_item[item];
			a.r2_off[k] = a.off_item[item];
			atomicAdd(&h_cells, (unsigned long long)ext_cells(a.items[item]));
		}
	}
	__syncthreads();
	if (threadIdx.x == 0 && h_cells) atomicAdd(&a.ctr->cells2, h_cells);
}

__global__ __launch_bounds__(64) void ext_records_kernel(ExtArgs a)
{
	extern __shared__ SelSlot sel[];
	const uint32_t q = blockIdx.x, lane = threadIdx.x;
	if (a.qstate[q] != EXT_Q_DEVICE) return;
	const uint32_t g0 = a.queries[q].group_begin, g1 = a.queries[q + 1].group_begin, ng = g1 - g0;
	const uint32_t first = a.kept_pos[g0];
	if (a.kept_pos[g1] == first) return;
	const uint32_t query = a.queries[q].query;
	for (uint32_t gi = lane; gi < ng; gi += 64) {
		const uint32_t g = g0 + gi;
		SelSlot s{ 0.0, 0, 0, 0, 0 };
		if (a.kept[g]) s = SelSlot{ a.cand_ev[g], a.ends[a.cand_item[g]].score, a.groups[g].target, 0, 1 };
		sel[gi] = s;
	}
	__syncthreads();
	for (uint32_t gi = lane; gi < ng; gi += 64) {
		const SelSlot me = sel[gi];
		if (!me.valid) continue;
		uint32_t rank = 0;
		for (uint32_t gj = 0; gj < ng; ++gj) rank += sel[gj].valid && gj != gi && sel_less(sel[gj], me) ? 1u : 0u;
		const uint32_t g = g0 + gi, item = a.cand_item[g];
		const dmnd_dp_target d = a.items[item];
		dmnd_match& m = a.records[first + rank];             // (field by field into HBM: a local record would live in scratch memory)
		m.query = query; m.target = me.target;
		m.ungapped_score = (int32_t)a.groups[g].score; m.d_begin = d.d_begin; m.d_end = d.d_end;
		m.frame = 0; m.read_begin = 0; m.read_end = 0;
		m.evalue = me.ev; m.bit_score = 0.0;                  // the host writes its own e-value and the bit score
		const dmnd_hsp hsp = a.hsps[item];
		m.hsp.score = hsp.score; m.hsp.q_begin = hsp.q_begin; m.hsp.q_end = hsp.q_end; m.hsp.s_begin = hsp.s_begin; m.hsp.s_end = hsp.s_end;
		m.hsp.length = hsp.length; m.hsp.identities = hsp.identities; m.hsp.mismatches = hsp.mismatches; m.hsp.positives = hsp.positives;
		m.hsp.gap_openings = hsp.gap_openings; m.hsp.gaps = hsp.gaps; m.hsp.transcript_len = hsp.transcript_len;
		m.hsp.transcript_off = -1;
	}
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

hipError_t launch_ext_prepare(const ExtArgs& a, hipStream_t st)
{
	hipError_t e = hipMemsetAsync(a.ctr, 0, sizeof(ExtCounters), st);
	if (e != hipSuccess) return e;
	const unsigned nb1 = a.n_bands + 1, bB = (nb1 + 255) / 256, bG = (a.n_groups + 255) / 256;
	size_t need_a = 0, need_b = 0, need_c = 0;
	e = rocprim::exclusive_scan(nullptr, need_a, a.cnt, a.item_off, 0u, (size_t)a.n_groups + 1, rocprim::plus<uint32_t>(), st);
	if (e != hipSuccess) return e;
	e = rocprim::radix_sort_pairs(nullptr, need_b, a.keys, a.keys_sorted, a.idx, a.order, (size_t)a.n_bands, 0, 15, st);
	if (e != hipSuccess) return e;
	e = rocprim::exclusive_scan(nullptr, need_c, a.rows_slot, a.off_slot, (int64_t)0, (size_t)nb1, rocprim::plus<int64_t>(), st);
	if (e != hipSuccess) return e;
	const size_t need = need_a > need_b ? (need_a > need_c ? need_a : need_c) : (need_b > need_c ? need_b : need_c);
	e = ensure_tmp(a.scan_tmp, a.scan_tmp_bytes, need);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(ext_mark_kernel, dim3(a.n_queries), dim3(64), 0, st, a);
	hipLaunchKernelGGL(ext_init_kernel, dim3(bB), dim3(256), 0, st, a);
	e = rocprim::exclusive_scan(*a.scan_tmp, need_a, a.cnt, a.item_off, 0u, (size_t)a.n_groups + 1, rocprim::plus<uint32_t>(), st);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(ext_items_kernel, dim3(bG), dim3(256), 0, st, a);
	e = rocprim::radix_sort_pairs(*a.scan_tmp, need_b, a.keys, a.keys_sorted, a.idx, a.order, (size_t)a.n_bands, 0, 15, st);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(ext_slots_kernel, dim3(bB), dim3(256), 0, st, a);
	e = rocprim::exclusive_scan(*a.scan_tmp, need_c, a.rows_slot, a.off_slot, (int64_t)0, (size_t)nb1, rocprim::plus<int64_t>(), st);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(ext_offsets_kernel, dim3(bB), dim3(256), 0, st, a);
	return hipGetLastError();
}

hipError_t launch_ext_select(const ExtArgs& a, hipStream_t st)
{
	hipError_t e = hipMemsetAsync(a.kept, 0, ((size_t)a.n_groups + 1) * sizeof(uint32_t), st);
	if (e != hipSuccess) return e;
	size_t need = 0;
	e = rocprim::exclusive_scan(nullptr, need, a.kept, a.kept_pos, 0u, (size_t)a.n_groups + 1, rocprim::plus<uint32_t>(), st);
	if (e != hipSuccess) return e;
	e = ensure_tmp(a.scan_tmp, a.scan_tmp_bytes, need);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(ext_select_kernel, dim3(a.n_queries), dim3(64), (size_t)a.chunk_size * sizeof(SelSlot), st, a);
	e = rocprim::exclusive_scan(*a.scan_tmp, need, a.kept, a.kept_pos, 0u, (size_t)a.n_groups + 1, rocprim::plus<uint32_t>(), st);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(ext_round2_kernel, dim3((a.n_groups + 255) / 256), dim3(256), 0, st, a);
	return hipGetLastError();
}

hipError_t launch_ext_records(const ExtArgs& a, uint32_t n_kept, hipStream_t st)
{
	if (n_kept == 0) return hipSuccess;
	hipLaunchKernelGGL(ext_records_kernel, dim3(a.n_queries), dim3(64), (size_t)a.chunk_size * sizeof(SelSlot), st, a);
	return hipGetLastError();
}

}  // namespace dmnd

// dmnd_init: the first launch of a kernel of this translation unit loads its code object onto the device
namespace { __global__ void touch_extend_kernel() {} }
extern "C" hipError_t dmnd_touch_extend(hipStream_t st) { hipLaunchKernelGGL(touch_extend_kernel, dim3(1), dim3(64), 0, st); return hipGetLastError(); }
