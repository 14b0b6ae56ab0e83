// extend_kernels.hip -- device half of the extension stage behind the planner (see extend_kernels.h; round 6, SURVEY.md 8 rows a14,
// a18 and the host preparation of a15 / a16 moved off the host for the queries whose targets fit one ranking chunk).
//
// Kernels, all latency-bound integer work on lists of 10^5 - 10^6 entries (a handful of bytes per entry, one pass each):
//   ext_mark_kernel      one wavefront per query: does the query run here? (no group left to the host by the planner, every band
//                        something the traceback-mode sweeps take); ranking state, sort keys of the ranking order
//   rocPRIM radix sort   -> every query's groups by (seed-hit score descending, load order)
//   per ranking-chunk iteration (most calls: one):
//     ext_window_kernel    item count of every group in an active query's current chunk; rocPRIM scan -> first item
//     ext_items_kernel     one thread per group: the DpTargets of its bands, band class, sweep steps, trace bytes, sort key; class
//                          histogram and DP cells (block-aggregated atomics)
//     rocPRIM radix sort (15 key bits) -> launch order: band class ascending, longest items first (api.hip order_slots)
//     ext_slots_kernel, rocPRIM scan, ext_offsets_kernel: trace offset of every item, item pairs of the packed 16-bit launches
//     (the sweeps: swipe16_kernels.hip / swipe_kernels.hip, launched by the host from the class histogram)
//     ext_append_kernel    one wavefront per query: best HSP per target past the report cutoff, append_hits (rank by counting when the
//                          aligned targets must be cut to k), next window, tail rule; ambiguity detection
//   ext_final_kernel     the first k of a query's aligned targets by (e-value, score desc, target); rocPRIM scan, ext_round2_kernel:
//                        the list the trace walk runs over (launched behind every iteration, used when no query is left ranking)
//   (traceback_kernel, swipe_kernels.hip)
//   ext_records_kernel   one wavefront per query: its match records in output order
// Compiled with -ffp-contract=off like plan_kernels.hip (the e-value below follows evalue.h operation by operation).
#include <hip/hip_runtime.h>
#include <cfloat>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include "extend_kernels.h"
#include "swipe16_core.h"
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

// bit score of a raw score (Evaluer::bitscore, evalue.h): plain double arithmetic, no library function -- the host's value bit for bit
__device__ inline double ext_bitscore(const ExtEvalue& p, int raw_score) { return (p.lambda * (double)raw_score - p.ln_k) / 0.69314718055994530941723212145818; }

__global__ __launch_bounds__(64) void ext_mark_kernel(ExtArgs a)
{
	const uint32_t q = blockIdx.x, lane = threadIdx.x;
	const uint32_t g0 = a.queries[q].group_begin, g1 = a.queries[q + 1].group_begin, ng = g1 - g0;
	const uint32_t query = a.queries[q].query;
	const int qlen = (int)(a.qlimits[query + 1] - a.qlimits[query] - 1);
	// more groups than a chunk: ranked in chunks (extend.cpp:289-336). A first chunk smaller than -k would grow by the e-value of
	// the seed-hit scores (extend.cpp:262-268): those queries stay on the host
	bool bad = ng > EXT_MAX_GROUPS || qlen <= 0 || (ng > a.chunk_size && (uint32_t)a.k > a.chunk_size);
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
		// ranking order = (score descending, load order): a stable sort by (query, 0xffff - score) of the groups as they lie
		a.okeys[g] = ((uint64_t)q << 16) | (uint64_t)(0xffffu - (uint32_t)a.groups[g].score);
		a.oidx[g] = g;
		a.aligned[g] = 0; a.g_cnt[g] = 0; a.g_first[g] = 0;
	}
	if (lane == 0) {
		a.qstate[q] = ok ? EXT_Q_DEVICE : EXT_Q_HOST;
		a.q_active[q] = ok ? 1 : 0;
		a.q_i0[q] = 0; a.q_i1[q] = ng < a.chunk_size ? ng : a.chunk_size;
		a.q_tail[q] = 0; a.q_prev[q] = 0;
	}
}

// the items of the current iteration: the bands of the passing groups of every active query's window. Every group gets its count
// (0 outside the windows) and a cleared `kept` flag here, and workgroup 0 resets the iteration's counters -- seven memset launches per
// iteration otherwise (each a kernel of its own on the stream)
__global__ __launch_bounds__(64) void ext_window_kernel(ExtArgs a)
{
	const uint32_t q = blockIdx.x, lane = threadIdx.x;
	const uint32_t g0 = a.queries[q].group_begin, g1 = a.queries[q + 1].group_begin;
	const bool active = a.q_active[q] != 0;
	const uint32_t i0 = active ? a.q_i0[q] : 0, i1 = active ? a.q_i1[q] : 0;
	uint32_t swept = 0;                        // targets of the window that are swept (they passed the filters before)
	for (uint32_t w = lane; w < g1 - g0; w += 64) {
		const uint32_t g = a.gorder[g0 + w];
		uint32_t n = 0;
		if (w >= i0 && w < i1) { const PlanGroup grp = a.groups[g]; n = grp.pass ? grp.n_bands : 0u; }
		a.cnt[g] = n;
		a.kept[g] = 0;
		swept += n ? 1u : 0u;
	}
	for (int off = 32; off >= 1; off >>= 1) swept += __shfl_xor(swept, off);
	if (lane == 0) a.q_swept[q] = swept;       // summed by ext_items_kernel (an atomic per query here is 10 ns each, one after the other)
	if (q == 0 && lane < EXT_CLASSES) { a.ctr->class_count[lane] = 0; a.ctr->class_max_steps[lane] = 0; }
	if (q == 0 && lane == 0) {
		a.cnt[a.n_groups] = 0; a.kept[a.n_groups] = 0;
		a.ctr->n_items = 0; a.ctr->n_active = 0; a.ctr->n_resweep = 0; a.ctr->total_rows = 0; a.ctr->cells2 = 0;
		a.ctr->window_targets = 0; a.ctr->window_bound = 0;
	}
}

__global__ __launch_bounds__(256) void ext_init_kernel(ExtArgs a)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, n = a.item_cap - a.item_base;
	if (i >= n) return;
	a.idx[i] = i; a.keys[i] = 0x7fffu; a.rows[i] = 0;
}

// launch class of an item (api.hip sweep_class: the row classes take what the packed 16-bit kernels take)
__device__ inline int ext_class(bool rows, int band, int64_t steps)
{
	return rows && steps <= 2 * (int64_t)SW16_MAX_PAIRS ? band_class_rows(band) : band_class(band);
}

__global__ __launch_bounds__(256) void ext_items_kernel(ExtArgs a)
{
	__shared__ uint32_t h_count[EXT_CLASSES], h_steps[EXT_CLASSES];
	__shared__ unsigned long long h_cells, h_diag, h_lane;
	__shared__ uint32_t h_targets, h_bound;
	if (threadIdx.x < EXT_CLASSES) { h_count[threadIdx.x] = 0; h_steps[threadIdx.x] = 0; }
	if (threadIdx.x == 0) { h_cells = 0; h_diag = 0; h_lane = 0; h_targets = 0; h_bound = 0; }
	__syncthreads();
	const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
	for (uint32_t q = g; q < a.n_queries; q += gridDim.x * blockDim.x) {      // the windows' swept targets, and how many of them can survive the culling
		const uint32_t swept = a.q_swept[q];
		if (swept) { atomicAdd(&h_targets, swept); atomicAdd(&h_bound, swept < (uint32_t)a.k ? swept : (uint32_t)a.k); }
	}
	const bool rows = a.item_off[a.n_groups] >= a.row_min_items;      // by the items of the whole iteration
	if (g < a.n_groups) {
		const uint32_t n = a.cnt[g];
		if (n) {
			const PlanGroup grp = a.groups[g];
			const uint32_t query = a.hits[grp.hit_begin].query;
			const int64_t q0 = a.qlimits[query], t0 = a.tlimits[grp.target];
			const int qlen = (int)(a.qlimits[query + 1] - q0 - 1), tlen = (int)(a.tlimits[grp.target + 1] - t0 - 1);
			const uint32_t local = a.item_off[g], first = a.item_base + local;
			a.g_first[g] = first; a.g_cnt[g] = n;
			unsigned long long cells = 0, diag = 0, lanes = 0;
			for (uint32_t k = 0; k < n; ++k) {
				const PlanBand b = a.bands[grp.band_begin + k];
				const dmnd_dp_target d{ q0, t0, a.use_cbs ? q0 : (int64_t)-1, qlen, tlen, b.d_begin, b.d_end };
				a.items[first + k] = d;
				const Geom geom = make_geom(qlen, tlen, b.d_begin, b.d_end);
				const int64_t steps = n_steps(geom);
				const int P = ext_class(rows, b.d_end - b.d_begin, steps);
				const int c = class_index(P);
				a.p_of_item[first + k] = P;
				a.rows[local + k] = trace_bytes(geom, P);
				const int64_t s16 = steps >> 4;
				a.keys[local + k] = ((uint32_t)c << 10) | (uint32_t)(1023 - (s16 < 1023 ? s16 : 1023));
				atomicAdd(&h_count[c], 1u);
				atomicMax(&h_steps[c], (uint32_t)(steps < 0x7fffffff ? steps : 0x7fffffff));
				cells += (unsigned long long)ext_cells(d);
				diag += (unsigned long long)(b.d_end - b.d_begin) * (unsigned long long)steps;
				lanes += (unsigned long long)(2 * P * class_lanes(P)) * (unsigned long long)steps;
			}
			atomicAdd(&h_cells, cells); atomicAdd(&h_diag, diag); atomicAdd(&h_lane, lanes);
		}
		if (g == 0) a.ctr->n_items = a.item_off[a.n_groups];
	}
	__syncthreads();
	if (threadIdx.x < EXT_CLASSES && h_count[threadIdx.x]) {
		atomicAdd(&a.ctr->class_count[threadIdx.x], h_count[threadIdx.x]);
		atomicMax(&a.ctr->class_max_steps[threadIdx.x], h_steps[threadIdx.x]);
	}
	if (threadIdx.x == 0 && h_cells) { atomicAdd(&a.ctr->cells1, h_cells); atomicAdd(&a.ctr->diag_steps, h_diag); atomicAdd(&a.ctr->lane_steps, h_lane); }
	if (threadIdx.x == 0 && h_targets) { atomicAdd(&a.ctr->window_targets, (unsigned long long)h_targets); atomicAdd(&a.ctr->window_bound, (unsigned long long)h_bound); }
}

__global__ __launch_bounds__(256) void ext_slots_kernel(ExtArgs a)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s > a.item_cap - a.item_base) return;
	a.rows_slot[s] = s < a.ctr->n_items ? a.rows[a.order[s]] : 0;
}

__global__ __launch_bounds__(256) void ext_offsets_kernel(ExtArgs a)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t n = a.ctr->n_items;
	if (s == 0) a.ctr->total_rows = (unsigned long long)a.off_slot[n];
	if (s >= n) return;
	const uint32_t item = a.order[s];                   // (relative to item_base, as the sweeps see the item arrays)
	a.off_item[a.item_base + item] = a.off_slot[s];
	// the packed 16-bit launches take their items in pairs (eights for a row class) of neighbours of the launch order; every class
	// starts a new wavefront, and the last wavefront of a class is filled up with -1 (dmnd_sweep_classes reads the same layout)
	const uint32_t c = a.keys_sorted[s] >> 10;
	uint32_t s0 = 0, pair0 = 0;
	for (uint32_t x = 0; x < c; ++x) {
		const uint32_t per = (uint32_t)class_items_per_wave16(class_of_index((int)x));
		s0 += a.ctr->class_count[x]; pair0 += (a.ctr->class_count[x] + per - 1) / per * per;
	}
	const uint32_t count = a.ctr->class_count[c], per = (uint32_t)class_items_per_wave16(class_of_index((int)c));
	a.pairs[pair0 + (s - s0)] = (int32_t)item;
	if (s - s0 == count - 1)
		for (uint32_t x = count; x < (count + per - 1) / per * per; ++x) a.pairs[pair0 + x] = -1;
}

// behind an iteration's sweeps: its items' trace offsets from the first arena on (the walk of round 2 reads all arenas from one
// base), or -1 where the sweeps ran without keeping trace rows
__global__ __launch_bounds__(256) void ext_rebase_kernel(ExtArgs a, uint32_t n_items, int64_t rel, int kept)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_items) a.off_item[a.item_base + i] = kept ? a.off_item[a.item_base + i] + rel : (int64_t)-1;
}

struct SelSlot { double ev; int score; uint32_t target; int tlen; uint32_t g; };

__device__ inline bool sel_less(const SelSlot& x, const SelSlot& y)      // Target::comp_evalue (target.h:123-129)
{
	return x.ev < y.ev || (x.ev == y.ev && (x.score > y.score || (x.score == y.score && x.target < y.target)));
}
// the host's own e-values could order the two the other way round (equal inputs give equal values on both sides)
__device__ inline bool sel_ambiguous(const SelSlot& x, const SelSlot& y) { return ext_near(x.ev, y.ev) && !(x.score == y.score && x.tlen == y.tlen); }

// the query's aligned targets into an LDS list (any order); returns their number (all lanes), cap + 1 if they do not fit
__device__ inline uint32_t gather_flagged(const ExtArgs& a, const uint8_t* flag8, const uint32_t* flag32, uint32_t g0, uint32_t ng, SelSlot* list, uint32_t cap, uint32_t* counter, uint32_t lane)
{
	if (lane == 0) *counter = 0;
	__syncthreads();
	for (uint32_t gi = lane; gi < ng; gi += 64) {
		const uint32_t g = g0 + gi;
		if (!(flag8 ? flag8[g] != 0 : flag32[g] != 0)) continue;
		const uint32_t x = atomicAdd(counter, 1u);
		if (x >= cap) continue;
		const uint32_t target = a.groups[g].target;
		list[x] = SelSlot{ a.cand_ev[g], a.ends[a.cand_item[g]].score, target, (int)(a.tlimits[target + 1] - a.tlimits[target] - 1), g };
	}
	__syncthreads();
	const uint32_t n = *counter;
	return n > cap ? cap + 1 : n;
}

// One ranking-chunk iteration of a query behind its sweeps (extend.cpp:289-336 with default options, one wavefront per query):
// the chunk's targets with a reported HSP (v), append_hits (culling.cpp:115-145) into the aligned targets, the next window, the
// tail rule (ranking_terminate, extend.cpp:111-119).
__global__ __launch_bounds__(64) void ext_append_kernel(ExtArgs a)
{
	extern __shared__ SelSlot lds[];
	__shared__ uint32_t n_v_sh, n_al_sh, flags_sh;
	__shared__ double kth_sh;
	__shared__ SelSlot kth_slot;
	const uint32_t q = blockIdx.x, lane = threadIdx.x;
	if (!a.q_active[q]) return;
	SelSlot* vs = lds;                                   // chunk_size entries
	SelSlot* al = lds + a.chunk_size;                    // k + chunk_size entries
	const uint32_t cap_al = (uint32_t)a.k + a.chunk_size;
	const uint32_t g0 = a.queries[q].group_begin, g1 = a.queries[q + 1].group_begin, ng = g1 - g0;
	const uint32_t i0 = a.q_i0[q], i1 = a.q_i1[q];
	const uint32_t query = a.queries[q].query;
	const int qlen = (int)(a.qlimits[query + 1] - a.qlimits[query] - 1);
	if (lane == 0) { n_v_sh = 0; flags_sh = 0; }
	__syncthreads();
	bool amb = false, sat = false;
	for (uint32_t w = i0 + lane; w < i1; w += 64) {
		const uint32_t g = a.gorder[g0 + w], n = a.g_cnt[g], first = a.g_first[g];
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
		a.cand_item[g] = bi;
		a.cand_ev[g] = bev;
		if (have) vs[atomicAdd(&n_v_sh, 1u)] = SelSlot{ bev, best, target, tlen, g };
	}
	__syncthreads();
	const uint32_t n_v = n_v_sh;
	bool new_hits = false;
	if (n_v > 0) {
		const uint32_t na = gather_flagged(a, a.aligned, nullptr, g0, ng, al, cap_al, &n_al_sh, lane);
		if (na > cap_al) amb = true;                       // (cannot happen: the list is cut to k before it grows by a chunk)
		new_hits = na < (uint32_t)a.k;
		const bool lost = __ballot(amb) != 0;              // (uniform: the branch below holds barriers)
		if (!new_hits && !lost) {
			// culling(targets, sort_only = false): the first k stay; the chunk is appended if its best e-value reaches the k-th's
			for (uint32_t x = lane; x < na; x += 64) {
				const SelSlot me = al[x];
				uint32_t rank = 0;
				for (uint32_t y = 0; y < na; ++y) {
					if (y == x) continue;
					if (sel_ambiguous(al[y], me)) amb = true;
					rank += sel_less(al[y], me) ? 1u : 0u;
				}
				if (rank >= (uint32_t)a.k) a.aligned[me.g] = 0;
				if (rank == (uint32_t)a.k - 1) { kth_sh = me.ev; kth_slot = me; }
			}
			__syncthreads();
			const double kth = kth_sh;
			const SelSlot ks = kth_slot;
			bool reach = false;
			for (uint32_t x = lane; x < n_v; x += 64) {
				if (sel_ambiguous(vs[x], ks)) amb = true;
				reach |= vs[x].ev <= kth;
			}
			if (__ballot(reach) != 0) new_hits = true;
		}
		if (new_hits) for (uint32_t x = lane; x < n_v; x += 64) a.aligned[vs[x].g] = 1;
	}
	const bool any_amb = __ballot(amb) != 0, any_sat = __ballot(sat) != 0;
	if (any_amb || any_sat) {
		// the host redoes the query: nothing of it stays here
		for (uint32_t gi = lane; gi < ng; gi += 64) a.aligned[g0 + gi] = 0;
		if (lane == 0) {
			a.qstate[q] = EXT_Q_AMBIGUOUS; a.q_active[q] = 0;
			if (any_amb) atomicAdd(&a.ctr->n_ambiguous, 1u);
			if (any_sat) atomicAdd(&a.ctr->n_saturated, 1u);
		}
		return;
	}
	if (lane == 0) {
		// the next window and whether the ranking goes on (extend.cpp:325-336)
		const uint32_t n0 = i1, n1 = i1 + (a.chunk_size < ng - i1 ? a.chunk_size : ng - i1);
		const int prev = a.q_tail[q];
		const int next_tail = (int)a.groups[a.gorder[g0 + n1 - 1]].score;
		a.q_prev[q] = prev;
		if (new_hits) a.q_tail[q] = next_tail;
		const bool terminate = !new_hits && (prev == 0 || (double)next_tail / (double)prev <= 0.95 || ext_bitscore(a.ev, next_tail) < 25.0);
		const bool go_on = n0 < ng && !terminate;
		a.q_i0[q] = n0; a.q_i1[q] = n1;
		a.q_active[q] = go_on ? 1 : 0;
		if (go_on) atomicAdd(&a.ctr->n_active, 1u);
	}
}

// culling(aligned_targets) once the ranking is over (extend.cpp:331): sort by (e-value, score, target), the first k survive
__global__ __launch_bounds__(64) void ext_final_kernel(ExtArgs a)
{
	extern __shared__ SelSlot lds[];
	__shared__ uint32_t n_sh;
	const uint32_t q = blockIdx.x, lane = threadIdx.x;
	if (a.qstate[q] != EXT_Q_DEVICE || a.q_active[q]) return;
	const uint32_t g0 = a.queries[q].group_begin, g1 = a.queries[q + 1].group_begin, ng = g1 - g0;
	const uint32_t cap = (uint32_t)a.k + a.chunk_size;
	const uint32_t na = gather_flagged(a, a.aligned, nullptr, g0, ng, lds, cap, &n_sh, lane);
	bool amb = na > cap;
	if (!amb)
		for (uint32_t x = lane; x < na; x += 64) {
			const SelSlot me = lds[x];
			bool keep = true;
			if (na > (uint32_t)a.k) {
				uint32_t rank = 0;
				for (uint32_t y = 0; y < na; ++y) {
					if (y == x) continue;
					if (sel_ambiguous(lds[y], me)) amb = true;
					rank += sel_less(lds[y], me) ? 1u : 0u;
				}
				keep = rank < (uint32_t)a.k;
			}
			a.kept[me.g] = keep ? 1u : 0u;
		}
	if (__ballot(amb) != 0) {
		for (uint32_t gi = lane; gi < ng; gi += 64) { a.kept[g0 + gi] = 0; a.aligned[g0 + gi] = 0; }
		if (lane == 0) { a.qstate[q] = EXT_Q_AMBIGUOUS; atomicAdd(&a.ctr->n_ambiguous, 1u); }
	}
}

__global__ __launch_bounds__(256) void ext_round2_kernel(ExtArgs a)
{
	__shared__ unsigned long long h_cells;
	__shared__ uint32_t h_lost;
	if (threadIdx.x == 0) { h_cells = 0; h_lost = 0; }
	__syncthreads();
	const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g < a.n_groups) {
		if (g == 0) a.ctr->n_kept = a.kept_pos[a.n_groups];
		if (a.kept[g]) {
			const uint32_t k = a.kept_pos[g], item = a.cand_item[g];
			a.r2_order[k] = (int32_t)item;
			a.r2_p[k] = a.p_of_item[item];
			a.r2_off[k] = a.off_item[item];
			a.r2_group[k] = g;
			if (a.off_item[item] < 0) atomicAdd(&h_lost, 1u);
			atomicAdd(&h_cells, (unsigned long long)ext_cells(a.items[item]));
		}
	}
	__syncthreads();
	if (threadIdx.x == 0 && h_cells) atomicAdd(&a.ctr->cells2, h_cells);
	if (threadIdx.x == 0 && h_lost) atomicAdd(&a.ctr->n_resweep, h_lost);
}

// Round 2 for the survivors whose round-1 sweep kept no trace rows (the reference's round 2 sweeps every survivor again with
// traceback, gapped_final.cpp:66-160): a copy of the item behind all others, as one more iteration; the group's best item is the copy
__global__ __launch_bounds__(256) void ext_resweep_kernel(ExtArgs a, uint32_t n_kept)
{
	__shared__ uint32_t h_count[EXT_CLASSES], h_steps[EXT_CLASSES];
	__shared__ unsigned long long h_cells;
	if (threadIdx.x < EXT_CLASSES) { h_count[threadIdx.x] = 0; h_steps[threadIdx.x] = 0; }
	if (threadIdx.x == 0) h_cells = 0;
	__syncthreads();
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k < n_kept && a.r2_off[k] < 0) {
		const uint32_t local = atomicAdd(&a.ctr->n_items, 1u), copy = a.item_base + local;
		const dmnd_dp_target d = a.items[a.r2_order[k]];
		a.items[copy] = d;
		const Geom geom = make_geom(d.query_len, d.target_len, d.d_begin, d.d_end);
		const int64_t steps = n_steps(geom);
		const int P = ext_class(a.ctr->n_resweep >= a.row_min_items, d.d_end - d.d_begin, steps);
		const int c = class_index(P);
		a.p_of_item[copy] = P;
		a.rows[local] = trace_bytes(geom, P);
		const int64_t s16 = steps >> 4;
		a.keys[local] = ((uint32_t)c << 10) | (uint32_t)(1023 - (s16 < 1023 ? s16 : 1023));
		atomicAdd(&h_count[c], 1u);
		atomicMax(&h_steps[c], (uint32_t)(steps < 0x7fffffff ? steps : 0x7fffffff));
		a.r2_order[k] = (int32_t)copy;
		a.r2_p[k] = P;                            // (the copy's class: the first sweep may have been in another one, ext_class)
		a.cand_item[a.r2_group[k]] = copy;
		atomicAdd(&h_cells, (unsigned long long)ext_cells(d));
	}
	__syncthreads();
	if (threadIdx.x < EXT_CLASSES && h_count[threadIdx.x]) {
		atomicAdd(&a.ctr->class_count[threadIdx.x], h_count[threadIdx.x]);
		atomicMax(&a.ctr->class_max_steps[threadIdx.x], h_steps[threadIdx.x]);
	}
	if (threadIdx.x == 0 && h_cells) atomicAdd(&a.ctr->cells_again, h_cells);
}

// ... and behind the sweeps of the copies: the walk's trace offsets of those slots
__global__ __launch_bounds__(256) void ext_rewalk_kernel(ExtArgs a, uint32_t n_kept)
{
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k < n_kept && a.r2_off[k] < 0) a.r2_off[k] = a.off_item[a.r2_order[k]];
}

__global__ __launch_bounds__(64) void ext_records_kernel(ExtArgs a)
{
	extern __shared__ SelSlot lds[];
	__shared__ uint32_t n_sh;
	const uint32_t q = blockIdx.x, lane = threadIdx.x;
	if (a.qstate[q] != EXT_Q_DEVICE) return;
	const uint32_t g0 = a.queries[q].group_begin, g1 = a.queries[q + 1].group_begin, ng = g1 - g0;
	const uint32_t first = a.kept_pos[g0];
	if (a.kept_pos[g1] == first) return;
	const uint32_t query = a.queries[q].query;
	const uint32_t nk = gather_flagged(a, nullptr, a.kept, g0, ng, lds, (uint32_t)a.k, &n_sh, lane);
	for (uint32_t x = lane; x < nk && x < (uint32_t)a.k; x += 64) {
		const SelSlot me = lds[x];
		uint32_t rank = 0;
		for (uint32_t y = 0; y < nk; ++y) rank += y != x && sel_less(lds[y], me) ? 1u : 0u;
		const uint32_t g = me.g, item = a.cand_item[g];
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

int bits_for(uint32_t n) { int b = 1; while (b < 32 && ((uint32_t)1 << b) <= n) ++b; return b; }

}  // namespace

hipError_t launch_ext_begin(const ExtArgs& a, hipStream_t st)
{
	hipError_t e = hipMemsetAsync(a.ctr, 0, sizeof(ExtCounters), st);
	if (e != hipSuccess) return e;
	e = hipMemsetAsync(a.r2_tr, 0, ((size_t)a.n_bands + 1) * sizeof(int64_t), st);
	if (e != hipSuccess) return e;
	const int key_bits = 16 + bits_for(a.n_queries);
	size_t need = 0;
	e = rocprim::radix_sort_pairs(nullptr, need, a.okeys, a.okeys_sorted, a.oidx, a.gorder, (size_t)a.n_groups, 0, key_bits, st);
	if (e != hipSuccess) return e;
	e = ensure_tmp(a.scan_tmp, a.scan_tmp_bytes, need);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(ext_mark_kernel, dim3(a.n_queries), dim3(64), 0, st, a);
	e = rocprim::radix_sort_pairs(*a.scan_tmp, need, a.okeys, a.okeys_sorted, a.oidx, a.gorder, (size_t)a.n_groups, 0, key_bits, st);
	if (e != hipSuccess) return e;
	return hipGetLastError();
}

namespace {

// the iteration's counters (the cell counts and the flags of the call stay)
hipError_t reset_iteration(const ExtArgs& a, hipStream_t st)
{
	hipError_t e = hipMemsetAsync(&a.ctr->n_items, 0, 2 * sizeof(uint32_t), st);
	if (e != hipSuccess) return e;
	return hipMemsetAsync(a.ctr->class_count, 0, 2 * EXT_CLASSES * sizeof(uint32_t) + sizeof(unsigned long long), st);
}

// launch order, trace offsets and pairs of the items the iteration has written (keys, rows, class histogram)
hipError_t order_items(const ExtArgs& a, hipStream_t st)
{
	const uint32_t n_left = a.item_cap - a.item_base;
	const unsigned bB = (n_left + 1 + 255) / 256;
	size_t need_b = 0, need_c = 0;
	hipError_t e = rocprim::radix_sort_pairs(nullptr, need_b, a.keys, a.keys_sorted, a.idx, a.order, (size_t)n_left, 0, 15, st);
	if (e != hipSuccess) return e;
	e = rocprim::exclusive_scan(nullptr, need_c, a.rows_slot, a.off_slot, (int64_t)0, (size_t)n_left + 1, rocprim::plus<int64_t>(), st);
	if (e != hipSuccess) return e;
	e = ensure_tmp(a.scan_tmp, a.scan_tmp_bytes, need_b > need_c ? need_b : need_c);
	if (e != hipSuccess) return e;
	if (n_left > 0) {
		e = rocprim::radix_sort_pairs(*a.scan_tmp, need_b, a.keys, a.keys_sorted, a.idx, a.order, (size_t)n_left, 0, 15, st);
		if (e != hipSuccess) return e;
	}
	hipLaunchKernelGGL(ext_slots_kernel, dim3(bB), dim3(256), 0, st, a);
	e = rocprim::exclusive_scan(*a.scan_tmp, need_c, a.rows_slot, a.off_slot, (int64_t)0, (size_t)n_left + 1, rocprim::plus<int64_t>(), st);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(ext_offsets_kernel, dim3(bB), dim3(256), 0, st, a);
	return hipGetLastError();
}

}  // namespace

hipError_t launch_ext_prepare(const ExtArgs& a, hipStream_t st)
{
	const uint32_t n_left = a.item_cap - a.item_base;
	size_t need_a = 0;
	hipError_t e = rocprim::exclusive_scan(nullptr, need_a, a.cnt, a.item_off, 0u, (size_t)a.n_groups + 1, rocprim::plus<uint32_t>(), st);
	if (e != hipSuccess) return e;
	e = ensure_tmp(a.scan_tmp, a.scan_tmp_bytes, need_a);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(ext_window_kernel, dim3(a.n_queries), dim3(64), 0, st, a);
	if (n_left > 0) hipLaunchKernelGGL(ext_init_kernel, dim3((n_left + 255) / 256), dim3(256), 0, st, a);
	e = rocprim::exclusive_scan(*a.scan_tmp, need_a, a.cnt, a.item_off, 0u, (size_t)a.n_groups + 1, rocprim::plus<uint32_t>(), st);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(ext_items_kernel, dim3((a.n_groups + 255) / 256), dim3(256), 0, st, a);
	return order_items(a, st);
}

hipError_t launch_ext_append(const ExtArgs& a, uint32_t n_items, bool kept, int64_t rel, hipStream_t st)
{
	if (n_items > 0 && (!kept || rel != 0)) hipLaunchKernelGGL(ext_rebase_kernel, dim3((n_items + 255) / 256), dim3(256), 0, st, a, n_items, rel, kept ? 1 : 0);
	// (n_active, n_resweep, cells2 and the kept flags were cleared by this iteration's ext_window_kernel: the round-2 list below is
	// rebuilt behind every iteration)
	hipError_t e = hipSuccess;
	const size_t lds_append = ((size_t)a.k + 2 * (size_t)a.chunk_size) * sizeof(SelSlot), lds_final = ((size_t)a.k + (size_t)a.chunk_size) * sizeof(SelSlot);
	hipLaunchKernelGGL(ext_append_kernel, dim3(a.n_queries), dim3(64), lds_append, st, a);
	// speculatively (the host only uses it when no query is left ranking): final culling, record slots, the round-2 list
	size_t need = 0;
	e = rocprim::exclusive_scan(nullptr, need, a.kept, a.kept_pos, 0u, (size_t)a.n_groups + 1, rocprim::plus<uint32_t>(), st);
	if (e != hipSuccess) return e;
	e = ensure_tmp(a.scan_tmp, a.scan_tmp_bytes, need);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(ext_final_kernel, dim3(a.n_queries), dim3(64), lds_final, st, a);
	e = rocprim::exclusive_scan(*a.scan_tmp, need, a.kept, a.kept_pos, 0u, (size_t)a.n_groups + 1, rocprim::plus<uint32_t>(), st);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(ext_round2_kernel, dim3((a.n_groups + 255) / 256), dim3(256), 0, st, a);
	return hipGetLastError();
}

hipError_t launch_ext_resweep(const ExtArgs& a, uint32_t n_kept, hipStream_t st)
{
	hipError_t e = reset_iteration(a, st);
	if (e != hipSuccess) return e;
	const uint32_t n_left = a.item_cap - a.item_base;
	if (n_left > 0) hipLaunchKernelGGL(ext_init_kernel, dim3((n_left + 255) / 256), dim3(256), 0, st, a);
	hipLaunchKernelGGL(ext_resweep_kernel, dim3((n_kept + 255) / 256), dim3(256), 0, st, a, n_kept);
	return order_items(a, st);
}

hipError_t launch_ext_rewalk(const ExtArgs& a, uint32_t n_items, uint32_t n_kept, int64_t rel, hipStream_t st)
{
	if (n_items > 0 && rel != 0) hipLaunchKernelGGL(ext_rebase_kernel, dim3((n_items + 255) / 256), dim3(256), 0, st, a, n_items, rel, 1);
	hipLaunchKernelGGL(ext_rewalk_kernel, dim3((n_kept + 255) / 256), dim3(256), 0, st, a, n_kept);
	return hipGetLastError();
}

namespace {
// the host's e-value and bit score into the device copy of the records (two doubles per record, in record order)
__global__ __launch_bounds__(256) void ext_patch_kernel(dmnd_match* records, const double* ev_bits, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	records[i].evalue = ev_bits[2 * i]; records[i].bit_score = ev_bits[2 * i + 1];
}
// block-local target ids -> database-wide ordinals while the records are gathered for a join
__global__ __launch_bounds__(256) void ext_gather_kernel(dmnd_match* dst, const dmnd_match* src, uint32_t n, uint32_t target_offset)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	dmnd_match m = src[i];
	m.target += target_offset;
	dst[i] = m;
}
}

hipError_t launch_ext_patch(dmnd_match* records, const double* ev_bits, uint32_t n, hipStream_t st)
{
	if (n == 0) return hipSuccess;
	hipLaunchKernelGGL(ext_patch_kernel, dim3((n + 255) / 256), dim3(256), 0, st, records, ev_bits, n);
	return hipGetLastError();
}

hipError_t launch_ext_gather(dmnd_match* dst, const dmnd_match* src, uint32_t n, uint32_t target_offset, hipStream_t st)
{
	if (n == 0) return hipSuccess;
	hipLaunchKernelGGL(ext_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dst, src, n, target_offset);
	return hipGetLastError();
}

hipError_t launch_ext_records(const ExtArgs& a, uint32_t n_kept, hipStream_t st)
{
	if (n_kept == 0) return hipSuccess;
	hipLaunchKernelGGL(ext_records_kernel, dim3(a.n_queries), dim3(64), (size_t)a.k * sizeof(SelSlot), st, a);
	return hipGetLastError();
}

}  // namespace dmnd

// dmnd_init: the first launch of a kernel of this translation unit loads its code object onto the device
namespace { __global__ void touch_extend_kernel() {} }
extern "C" hipError_t dmnd_touch_extend(hipStream_t st) { hipLaunchKernelGGL(touch_extend_kernel, dim3(1), dim3(64), 0, st); return hipGetLastError(); }
