// swipe16_kernels.hip -- gfx950 (MI355X) banded Smith-Waterman sweep on the packed 16-bit VALU, two work items per wavefront.
//
// Replaces, for bands of up to 128 * SW16_MAX_P diagonals, the 32-bit kernels of swipe_kernels.hip behind
// DP::BandedSwipe::swipe (/root/reference/src/dp/swipe/banded_swipe.h:189-351; the reference's 16-bit vectors with overflow
// escalation: score_vector_int16.h, swipe_wrapper.cpp:317-360). Per-lane arithmetic and its correctness argument:
// swipe16_core.h. What this file adds is the wavefront schedule:
//   * one wavefront per PAIR of work items (low / high halves of every DP register), 4 wavefronts per workgroup;
//   * per pair-step one DPP shift each for F (even step) and E (odd step), and three for the packed letter windows (query
//     letters, their bias, target letters of both items) -- only lane 63 / lane 0 take new letters, from an edge record that
//     the wavefront prepared in LDS for the next 256 pair-steps (one broadcast ds_read_b128 per pair-step, issued a whole
//     pair-step before use; the inner loop has no global loads and next to no scalar-unit work);
//   * the scores of pair-step t + 1 are looked up in LDS (32x32 table of 16-bit entries with the sentinel's row and column
//     at -128) BEFORE the cells of pair-step t are computed, so LDS latency hides behind ~40 packed VALU instructions
//     instead of stalling the first cell of every step;
//   * trace (TRACE): one nibble per cell, collected in registers over a group of 16 / P pair-steps and written as ONE 16-byte
//     record per lane, item and group (swipe_core.h, trace_byte_index) -- half the bytes and a sixteenth of the store
//     instructions of a row per anti-diagonal step, and the walk (traceback_kernel) finds 16 / P consecutive columns of a
//     diagonal in one record instead of one 128-byte line per column.
//   * ROW classes (P = 3 / 5, banded_swipe16_rows_kernel): the same sweep with an item pair per 16-lane DPP ROW -- 2 P = 6 / 10
//     diagonals per lane, 96 / 160 per item, EIGHT items per wavefront. The shifts are row shifts (row_shr:1 / row_shl:1 stop at
//     the row's ends, which are the bands' edges), the items' geometry and the edge records are per row, and the wavefront runs
//     until its longest item is done. Bands of 61-81 / 131-161 diagonals -- most of them -- fill 0.63-0.84 / 0.82-1.0 of their lanes
//     there against 0.48-0.63 / 0.51-0.63 in the classes 1 / 2. That is the whole gain: a cell costs the same 31 (18, 15.9 without
//     end cells) VALU instructions in every class -- the shifts and window moves are a few per cent of a pair-step --, so the sweeps'
//     instructions fall by the lane use (measured 1.42 x on C2skew). A wavefront of eight items is a longer dependency chain per
//     anti-diagonal step, though: the row classes are taken when a launch set fills the chip (api.hip sweep_rows_min_items).
//   * COORDS = false (DMND_SWIPE_SCORE, the device half's scores-first pass): one packed max per cell instead of the two end-cell keys.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>
#include "swipe16_core.h"
#include "swipe_kernels.h"

namespace dmnd {

// wave_shr:1 -> lane l reads lane l-1, wave_shl:1 -> lane l reads lane l+1. *_z: the edge lane gets 0; *_in: it keeps `edge`.
__device__ __forceinline__ uint32_t shr1_z(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t shl1_z(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t shr1_in(uint32_t edge, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t shl1_in(uint32_t edge, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130, 0xf, 0xf, false); }

// row_shr:1 / row_shl:1: the same inside every row of 16 lanes (lane 0 / 15 of a row is its edge lane)
__device__ __forceinline__ uint32_t rshr1_z(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t rshl1_z(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t rshr1_in(uint32_t edge, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x111, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t rshl1_in(uint32_t edge, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x101, 0xf, 0xf, false); }

__device__ __forceinline__ int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ int64_t rfl64(int64_t x)
{
	const uint32_t lo = (uint32_t)rfl((int)(uint32_t)x), hi = (uint32_t)rfl((int)(uint32_t)((uint64_t)x >> 32));
	return (int64_t)(((uint64_t)hi << 32) | lo);
}

// one work item of the pair with every field in scalar registers (it is the same for all 64 lanes)
struct Item16 {
	Geom g;
	SeqView v;
	int pairs;                      // pair-steps
};

__device__ __forceinline__ Item16 load_item(const dmnd_dp_target* items, int idx, const int8_t* qblock, const int8_t* tblock, const int8_t* cbs)
{
	const dmnd_dp_target it = items[idx];
	Item16 r;
	r.g = make_geom(rfl(it.query_len), rfl(it.target_len), rfl(it.d_begin), rfl(it.d_end));
	const int64_t c = rfl64(it.cbs_off);
	r.v = SeqView{ qblock + rfl64(it.query_off), tblock + rfl64(it.target_off), c >= 0 ? cbs + c : nullptr, nullptr };
	r.pairs = sw16_pairs(r.g);
	return r;
}

// ... and with every field per lane (the same within a row: the row classes)
__device__ __forceinline__ Item16 load_item_row(const dmnd_dp_target* items, int idx, const int8_t* qblock, const int8_t* tblock, const int8_t* cbs)
{
	const dmnd_dp_target it = items[idx];
	Item16 r;
	r.g = make_geom(it.query_len, it.target_len, it.d_begin, it.d_end);
	r.v = SeqView{ qblock + it.query_off, tblock + it.target_off, it.cbs_off >= 0 ? cbs + it.cbs_off : nullptr, nullptr };
	r.pairs = sw16_pairs(r.g);
	return r;
}

// f(integral_constant<0>) ... f(integral_constant<N-1>): a loop whose index is a compile-time constant in every iteration
template<int... R, typename F>
__device__ __forceinline__ void unrolled(std::integer_sequence<int, R...>, F&& f) { (f(std::integral_constant<int, R>()), ...); }

enum { EDGE_CHUNK = 256 };            // pair-steps per refill of a wavefront's edge records (4 records per lane)

template<int P, bool TRACE, bool COORDS = true>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64)
void banded_swipe16_kernel(const int8_t* __restrict__ qblock, const int8_t* __restrict__ tblock, const int8_t* __restrict__ cbs,
	const int8_t* __restrict__ matrix, const dmnd_dp_target* __restrict__ items, const int32_t* __restrict__ pairs,
	const int64_t* __restrict__ trace_off, uint8_t* __restrict__ trace, SwipeEnd* __restrict__ ends, int64_t n_pairs, int gap_open, int gap_extend)
{
	__shared__ uint16_t table[32 * 32];
	__shared__ uint4 edges[WAVES_PER_BLOCK][EDGE_CHUNK];      // per wavefront: Edge16 (qq, tt, cc) of the next EDGE_CHUNK pair-steps
	for (int x = threadIdx.x; x < 32 * 32; x += blockDim.x)
		table[x] = sw16_table_entry(matrix, x);
	__syncthreads();

	const int lane = threadIdx.x & 63, wave = rfl((int)(threadIdx.x >> 6));
	const int64_t slot = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave;
	if (slot >= n_pairs)
		return;
	const int idxA = rfl(pairs[2 * slot]), idxB_raw = rfl(pairs[2 * slot + 1]);
	const bool hasB = idxB_raw >= 0;                   // the odd item of a class shares its wavefront with a copy of itself
	const int idxB = hasB ? idxB_raw : idxA;
	const Item16 A = load_item(items, idxA, qblock, tblock, cbs), B = load_item(items, idxB, qblock, tblock, cbs);
	const pk16 go = pk_both(gap_open + gap_extend), ge = pk_both(gap_extend);
	const int nA = A.pairs, nB = hasB ? B.pairs : 0;

	Lane16<P> st;
	lane16_init(st, A.g, A.v, B.g, B.v, lane);
	constexpr int G = Sw16Group<P>::G;                 // pair-steps per trace record (16 bytes per lane and item)
	uint8_t *recA = nullptr, *recB = nullptr;          // this lane's record of the current group
	if (TRACE) {
		recA = trace + rfl64(trace_off[idxA]) + lane * 16;
		recB = trace + rfl64(trace_off[idxB]) + lane * 16;
	}
	uint4* const my_edges = edges[wave];

	pk16 S0[P], S1[P];
	lane16_scores(st, table, S0, S1);
	Edge16 e = sw16_edge(A.g, A.v, B.g, B.v, P, 0);        // enters at the end of pair-step 0
	Trace16Group<P> acc;

	// One pair-step; R = its position inside the group of G (compile time: the trace bytes go to fixed places of the record)
	auto pair_step = [&](auto r, int t, int tc) {
		constexpr int R = decltype(r)::value;
		// edge record of the NEXT pair-step (LDS broadcast read, in flight during this one)
		const uint4 nx = my_edges[t - tc];
		// windows of pair-step t + 1 and its scores: the LDS reads are in flight while the cells of pair-step t are computed
		lane16_advance(st, shl1_in(e.qq, st.QQ[1]), shl1_in(e.cc, st.CC[1]), shr1_in(e.tt, st.TT[P - 1]));
		pk16 N0[P], N1[P];
		lane16_scores(st, table, N0, N1);
		const uint32_t revt = 0xffffu - (uint32_t)t;
		pk16 tb0[P], tb1[P];
		lane16_step<P, TRACE, 0, COORDS>(st, S0, shr1_z(st.F[2 * P - 1]), go, ge, revt, tb0);
		lane16_step<P, TRACE, 1, COORDS>(st, S1, shl1_z(st.E[0]), go, ge, revt, tb1);
		if (TRACE) acc.template put<R>(tb0, tb1);
#pragma unroll
		for (int p = 0; p < P; ++p) { S0[p] = N0[p]; S1[p] = N1[p]; }
		e.qq = nx.x; e.tt = nx.y; e.cc = nx.z;
		// the group is unrolled: without a fence the scheduler hoists the LDS reads of all its pair-steps to the top (16 edge
		// records = 64 registers at P = 1) and halves the occupancy
		if (R % 2 == 1) __builtin_amdgcn_sched_barrier(0);
	};
	// The sweep runs in whole groups of G pair-steps, fully unrolled (the window registers rotate with period P + 1 and the
	// score registers with period 2, so the copies at the end of a pair-step become renames; the DPP moves are convergent
	// operations, which keeps the compiler from unrolling a loop with a run-time trip count by itself). The last group of an
	// item may run past its last pair-step: those cells lie behind the matrix, where nothing can reach the best-score record
	// (swipe16_core.h), and their trace bytes land in the padding of the item's last record.
	const int nMax = nA > nB ? nA : nB, T = (nMax + G - 1) / G * G;
	for (int tc = 0; tc < T; tc += EDGE_CHUNK) {
		// records of pair-steps tc + 1 .. tc + EDGE_CHUNK (slot r holds pair-step tc + 1 + r), built by the lanes in parallel
		// (adjacent lanes read adjacent letters); only this wavefront reads them back, so a wave-level barrier orders the
		// writes before the reads
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int r = lane; r < EDGE_CHUNK; r += 64) {
			const Edge16 x = sw16_edge(A.g, A.v, B.g, B.v, P, tc + 1 + r);
			my_edges[r] = make_uint4(x.qq, x.tt, x.cc, 0u);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		const int te = tc + EDGE_CHUNK < T ? tc + EDGE_CHUNK : T;      // EDGE_CHUNK is a multiple of every G
		for (int t = tc; t < te; t += G) {
			unrolled(std::make_integer_sequence<int, G>(), [&](auto r) { pair_step(r, t + decltype(r)::value, tc); });
			if (TRACE) {
				// one 16-byte store per lane and item: the 64 lanes write 1 KiB of consecutive bytes
				if (t < nA) *reinterpret_cast<uint4*>(recA) = make_uint4(acc.a[0], acc.a[1], acc.a[2], acc.a[3]);
				if (t < nB) *reinterpret_cast<uint4*>(recB) = make_uint4(acc.b[0], acc.b[1], acc.b[2], acc.b[3]);
				recA += 1024; recB += 1024;
			}
		}
	}

	// end cells of both items: per lane from its diagonal keys, then a wave reduction
#pragma unroll
	for (int item = 0; item < 2; ++item) {
		if (item == 1 && !hasB) break;
		int bs, bi, bj;
		lane16_finish<P, COORDS>(st, item ? B.g : A.g, item == 1, lane, bs, bi, bj);
#pragma unroll
		for (int off = 32; off >= 1; off >>= 1) {
			const int os = __shfl_xor(bs, off), oi = __shfl_xor(bi, off), oj = __shfl_xor(bj, off);
			if (better_end(os, oj, oi, bs, bj, bi)) { bs = os; bi = oi; bj = oj; }
		}
		if (lane == 0) {
			SwipeEnd x;
			x.score = bs; x.end_i = bi; x.end_j = bj; x.stat_a = 0; x.stat_b = 0;
			x.pad[0] = bs >= SW16_MAX_SCORE ? 1 : 0;      // saturated: the host re-runs the item in the 32-bit kernel
			x.pad[1] = x.pad[2] = 0;
			ends[item ? idxB : idxA] = x;
		}
	}
}

// ---- row classes ----------------------------------------------------------------------------------------------------------
// pairs[8 w + 2 r], pairs[8 w + 2 r + 1]: the items A / B of row r of wavefront w (-1: none -- the tail of a class; the first is
// always there). A row without an item sweeps a copy of another one and stores nothing.
enum { ROW_EDGE_CHUNK = 60 };         // pair-steps per refill of a row's edge records: a multiple of both G = 5 and G = 3 (16 KB per workgroup: the registers, not the LDS, bound the wavefronts per SIMD)

template<int P, bool TRACE, bool COORDS = true>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64)
void banded_swipe16_rows_kernel(const int8_t* __restrict__ qblock, const int8_t* __restrict__ tblock, const int8_t* __restrict__ cbs,
	const int8_t* __restrict__ matrix, const dmnd_dp_target* __restrict__ items, const int32_t* __restrict__ pairs,
	const int64_t* __restrict__ trace_off, uint8_t* __restrict__ trace, SwipeEnd* __restrict__ ends, int64_t n_waves, int gap_open, int gap_extend)
{
	__shared__ uint16_t table[32 * 32];
	// per wavefront and row: Edge16 (qq, tt, cc) of the next ROW_EDGE_CHUNK pair-steps (+ 1: the four rows of a wavefront read the
	// same record number at the same time, from different banks)
	__shared__ uint4 edges[WAVES_PER_BLOCK][4][ROW_EDGE_CHUNK + 1];
	for (int x = threadIdx.x; x < 32 * 32; x += blockDim.x)
		table[x] = sw16_table_entry(matrix, x);
	__syncthreads();

	const int lane = threadIdx.x & 63, wave = rfl((int)(threadIdx.x >> 6)), row = lane >> 4, rl = lane & 15;
	const int64_t slot = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave;
	if (slot >= n_waves)
		return;
	const int idx0 = rfl(pairs[8 * slot]);
	const int idxA_raw = pairs[8 * slot + 2 * row], idxB_raw = pairs[8 * slot + 2 * row + 1];
	const bool hasA = idxA_raw >= 0, hasB = idxB_raw >= 0;
	const int idxA = hasA ? idxA_raw : idx0, idxB = hasB ? idxB_raw : idxA;
	// The items' geometry and pointers are per row, i.e. in vector registers: they are loaded where they are needed (start, every
	// refill of the edge records, end) instead of living through the sweep -- `again` hides the index from the compiler so that it
	// does not keep the first load's 30 registers alive
	auto again = [](int idx) { asm volatile("" : "+v"(idx)); return idx; };
	const pk16 go = pk_both(gap_open + gap_extend), ge = pk_both(gap_extend);
	int nA, nB;
	Lane16<P> st;
	pk16 S0[P], S1[P];
	Edge16 e;
	{
		const Item16 A = load_item_row(items, idxA, qblock, tblock, cbs), B = load_item_row(items, idxB, qblock, tblock, cbs);
		nA = hasA ? A.pairs : 0; nB = hasB ? B.pairs : 0;
		lane16_init(st, A.g, A.v, B.g, B.v, rl);
		lane16_scores(st, table, S0, S1);
		e = sw16_edge(A.g, A.v, B.g, B.v, P, 0, 15);
	}
	constexpr int G = Sw16Group<P>::G;
	uint8_t *recA = nullptr, *recB = nullptr;
	if (TRACE) {
		recA = trace + trace_off[idxA] + rl * 16;
		recB = trace + trace_off[idxB] + rl * 16;
	}
	uint4* const my_edges = edges[wave][row];
	Trace16Group<P> acc;

	auto pair_step = [&](auto r, int t, int tc) {
		constexpr int R = decltype(r)::value;
		const uint4 nx = my_edges[t - tc];
		lane16_advance(st, rshl1_in(e.qq, st.QQ[1]), rshl1_in(e.cc, st.CC[1]), rshr1_in(e.tt, st.TT[P - 1]));
		pk16 N0[P], N1[P];
		lane16_scores(st, table, N0, N1);
		const uint32_t revt = 0xffffu - (uint32_t)t;
		pk16 tb0[P], tb1[P];
		lane16_step<P, TRACE, 0, COORDS>(st, S0, rshr1_z(st.F[2 * P - 1]), go, ge, revt, tb0);
		lane16_step<P, TRACE, 1, COORDS>(st, S1, rshl1_z(st.E[0]), go, ge, revt, tb1);
		if (TRACE) acc.template put<R>(tb0, tb1);
#pragma unroll
		for (int p = 0; p < P; ++p) { S0[p] = N0[p]; S1[p] = N1[p]; }
		e.qq = nx.x; e.tt = nx.y; e.cc = nx.z;
		__builtin_amdgcn_sched_barrier(0);
	};
	// the wavefront runs until its longest item is done (the launch order puts items of similar length next to each other); what
	// a shorter one computes behind its matrix is harmless (swipe16_core.h) and is not stored
	int nMax = nA > nB ? nA : nB;
	nMax = max(max(__builtin_amdgcn_readlane(nMax, 0), __builtin_amdgcn_readlane(nMax, 16)), max(__builtin_amdgcn_readlane(nMax, 32), __builtin_amdgcn_readlane(nMax, 48)));
	const int T = (nMax + G - 1) / G * G;
	for (int tc = 0; tc < T; tc += ROW_EDGE_CHUNK) {
		__builtin_amdgcn_wave_barrier();
		{
			const Item16 A = load_item_row(items, again(idxA), qblock, tblock, cbs), B = load_item_row(items, again(idxB), qblock, tblock, cbs);
			for (int r = rl; r < ROW_EDGE_CHUNK; r += 16) {
				const Edge16 x = sw16_edge(A.g, A.v, B.g, B.v, P, tc + 1 + r, 15);
				my_edges[r] = make_uint4(x.qq, x.tt, x.cc, 0u);
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		const int te = tc + ROW_EDGE_CHUNK < T ? tc + ROW_EDGE_CHUNK : T;
		for (int t = tc; t < te; t += G) {
			unrolled(std::make_integer_sequence<int, G>(), [&](auto r) { pair_step(r, t + decltype(r)::value, tc); });
			if (TRACE) {
				// one 16-byte store per lane and item: the 16 lanes of a row write 256 consecutive bytes
				if (t < nA) *reinterpret_cast<uint4*>(recA) = make_uint4(acc.a[0], acc.a[1], acc.a[2], acc.a[3]);
				if (t < nB) *reinterpret_cast<uint4*>(recB) = make_uint4(acc.b[0], acc.b[1], acc.b[2], acc.b[3]);
				recA += 256; recB += 256;
			}
		}
	}

#pragma unroll
	for (int item = 0; item < 2; ++item) {
		const dmnd_dp_target it = items[again(item ? idxB : idxA)];
		const Geom g = make_geom(it.query_len, it.target_len, it.d_begin, it.d_end);
		int bs, bi, bj;
		lane16_finish<P, COORDS>(st, g, item == 1, rl, bs, bi, bj);
#pragma unroll
		for (int off = 8; off >= 1; off >>= 1) {
			const int os = __shfl_xor(bs, off), oi = __shfl_xor(bi, off), oj = __shfl_xor(bj, off);
			if (better_end(os, oj, oi, bs, bj, bi)) { bs = os; bi = oi; bj = oj; }
		}
		if (rl == 0 && (item ? hasB : hasA)) {
			SwipeEnd x;
			x.score = bs; x.end_i = bi; x.end_j = bj; x.stat_a = 0; x.stat_b = 0;
			x.pad[0] = bs >= SW16_MAX_SCORE ? 1 : 0;
			x.pad[1] = x.pad[2] = 0;
			ends[item ? idxB : idxA] = x;
		}
	}
}

template<int P>
static hipError_t launch16_rows(bool trace, const Swipe16Args& a, hipStream_t stream)
{
	const unsigned blocks = (unsigned)((a.n_pairs + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
	if (blocks == 0)
		return hipSuccess;
	const dim3 grid(blocks), block(WAVES_PER_BLOCK * 64);
	if (trace) hipLaunchKernelGGL((banded_swipe16_rows_kernel<P, true>), grid, block, 0, stream, a.qblock, a.tblock, a.cbs, a.matrix, a.items, a.pairs, a.trace_off, a.trace, a.ends, a.n_pairs, a.gap_open, a.gap_extend);
	else if (a.score_only) hipLaunchKernelGGL((banded_swipe16_rows_kernel<P, false, false>), grid, block, 0, stream, a.qblock, a.tblock, a.cbs, a.matrix, a.items, a.pairs, a.trace_off, a.trace, a.ends, a.n_pairs, a.gap_open, a.gap_extend);
	else hipLaunchKernelGGL((banded_swipe16_rows_kernel<P, false>), grid, block, 0, stream, a.qblock, a.tblock, a.cbs, a.matrix, a.items, a.pairs, a.trace_off, a.trace, a.ends, a.n_pairs, a.gap_open, a.gap_extend);
	return hipGetLastError();
}

template<int P>
static hipError_t launch16_p(bool trace, const Swipe16Args& a, hipStream_t stream)
{
	const unsigned blocks = (unsigned)((a.n_pairs + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
	if (blocks == 0)
		return hipSuccess;
	const dim3 grid(blocks), block(WAVES_PER_BLOCK * 64);
	if (trace) hipLaunchKernelGGL((banded_swipe16_kernel<P, true>), grid, block, 0, stream, a.qblock, a.tblock, a.cbs, a.matrix, a.items, a.pairs, a.trace_off, a.trace, a.ends, a.n_pairs, a.gap_open, a.gap_extend);
	else if (a.score_only) hipLaunchKernelGGL((banded_swipe16_kernel<P, false, false>), grid, block, 0, stream, a.qblock, a.tblock, a.cbs, a.matrix, a.items, a.pairs, a.trace_off, a.trace, a.ends, a.n_pairs, a.gap_open, a.gap_extend);
	else hipLaunchKernelGGL((banded_swipe16_kernel<P, false>), grid, block, 0, stream, a.qblock, a.tblock, a.cbs, a.matrix, a.items, a.pairs, a.trace_off, a.trace, a.ends, a.n_pairs, a.gap_open, a.gap_extend);
	return hipGetLastError();
}

hipError_t launch_banded_swipe16(int P, bool trace, const Swipe16Args& a, hipStream_t stream)
{
	switch (P) {
	case 1: return launch16_p<1>(trace, a, stream);
	case 2: return launch16_p<2>(trace, a, stream);
	case 4: return launch16_p<4>(trace, a, stream);
	case 3: return launch16_rows<3>(trace, a, stream);      // a.n_pairs = wavefronts, a.pairs = eight items each
	case 5: return launch16_rows<5>(trace, a, stream);
	default: return hipErrorInvalidValue;
	}
}

}  // namespace dmnd

// dmnd_init: the first launch of a kernel of this translation unit loads its code object onto the device
namespace { __global__ void touch_swipe16_kernel() {} }
extern "C" hipError_t dmnd_touch_swipe16(hipStream_t st) { hipLaunchKernelGGL(touch_swipe16_kernel, dim3(1), dim3(64), 0, st); return hipGetLastError(); }
