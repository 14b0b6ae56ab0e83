// swipe_kernels.hip -- gfx950 (MI355X) kernels for the banded Smith-Waterman extension.
//
// Replaces the reference's SIMD kernels behind DP::BandedSwipe::swipe
// (/root/reference/src/dp/swipe/banded_swipe.h:189-351, swipe_wrapper.cpp:446-470).
// Design (see swipe_core.h for the per-lane arithmetic and DESIGN.md for the numbers):
//   * one 64-lane wavefront per work item (DpTarget), anti-diagonal sweep; lane l owns 2*P
//     consecutive band diagonals, P chosen per item so that 128*P >= band;
//   * H/E/F live in VGPRs; the only cross-lane traffic is ONE DPP wave shift per step
//     (v_mov_b32_dpp wave_shr:1 / wave_shl:1 -- no LDS, no ds_bpermute);
//   * the item's 32x32 int8 substitution matrix -- the context's, or the item's own composition-adjusted one
//     (--comp-based-stats 2..5: dp/swipe/target_iterator.h:124-134 hands every SIMD channel its target's matrix) -- is staged in
//     LDS by the item's wavefront;
//   * letters: a lane's cells of one (even, odd) step pair need P + 1 consecutive query letters and P consecutive target
//     letters, and the next pair the same windows moved by one, so the windows live in VGPRs and every pair fetches ONE
//     query letter, ONE bias byte and ONE target letter per lane (scalar base + lane offset, adjacent lanes adjacent bytes),
//     issued a whole pair before their first use; cells are branch-free (invalid cells are zeroed by v_cndmask), the end
//     cell is kept as one (best score, first anti-diagonal) record per diagonal and turned into coordinates after the sweep;
//   * TRACEBACK mode streams one trace nibble per cell to an HBM arena in the layout of swipe_core.h (trace_byte_index:
//     a lane's bytes of 16 / P consecutive pair-steps are one 16-byte record); a second kernel walks it with one wavefront
//     per item (64 columns per round trip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include "swipe_core.h"
#include "swipe_kernels.h"

namespace dmnd {

// wave_shr:1 -> lane l reads lane l-1, lane 0 keeps `old` (0);  wave_shl:1 -> lane l reads lane l+1, lane 63 gets 0
__device__ __forceinline__ int wave_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int wave_shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }

__device__ __forceinline__ int64_t uniform64(int64_t x)
{
	const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)x >> 32));
	return (int64_t)(((uint64_t)hi << 32) | lo);
}

// MULTI (bands wider than one wavefront's 128 * P diagonals, P at its maximum): the item is swept by all the wavefronts of the
// workgroup, wavefront w owning the diagonals of the virtual lanes 64 w .. 64 w + 63. The cross-lane shifts continue over the
// wavefront boundaries through LDS with one workgroup barrier per anti-diagonal step -- slow next to the single-wavefront sweep,
// and meant for the rare long repeat proteins whose merged band exceeds 4096 (2048 with statistics) diagonals; the reference
// escalates its row counters the same way (RowCounter, banded_swipe.h:204). The trace is that of an item of band class
// P * wavefronts (one row of 64 * P * wavefronts bytes per pair-step).
constexpr int MULTI_MAX_WAVES = 16;

// MAXW: the largest workgroup the instantiation is launched with, in wavefronts. The register budget of a lane halves with every
// doubling (512 VGPRs per SIMD lane slot / wavefronts per SIMD), so the common 2- and 4-wavefront cases get their own build that
// keeps H/E/F in registers, and only the 8- and 16-wavefront build spills.
template<int P, bool COORDS, bool TRACE, int STAT = STAT_NONE, bool REV = false, bool MULTI = false, int MAXW = MULTI_MAX_WAVES>
__global__ __launch_bounds__(MULTI ? MAXW * 64 : WAVES_PER_BLOCK * 64)
void banded_swipe_kernel(SwipeArgs args)
{
	// one scoring matrix per item: the context's, or the item's own composition-adjusted one (cbs_off <= -2), staged by the
	// item's wavefront(s)
	__shared__ int8_t matrices[MULTI ? 1 : WAVES_PER_BLOCK][32 * 32];
	__shared__ int edge[2][3][MULTI ? MULTI_MAX_WAVES : 1];
	__shared__ int red[MULTI ? MULTI_MAX_WAVES : 1][5];

	const int wave = threadIdx.x >> 6, n_waves = MULTI ? (int)(blockDim.x >> 6) : 1;
	const int lane = MULTI ? (int)threadIdx.x : (int)(threadIdx.x & 63);          // MULTI: the virtual lane, 0 .. 64 * n_waves - 1
	const int64_t slot = MULTI ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave;
	const bool live = slot < args.n;
	int32_t item_idx = 0;
	dmnd_dp_target it = { 0, 0, -1, 1, 1, 0, 1 };
	if (live) { item_idx = args.order[slot]; it = args.items[item_idx]; }
	const int64_t q_off = uniform64(it.query_off), t_off = uniform64(it.target_off), c_off = uniform64(it.cbs_off);
	int8_t* const matrix = matrices[MULTI ? 0 : wave];
	{
		const int32_t* src = reinterpret_cast<const int32_t*>(c_off <= -2 ? args.matrices + own_matrix_number(c_off) * (32 * 32) : args.matrix);
		for (int x = lane; x < 32 * 32 / 4; x += MULTI ? (int)blockDim.x : 64)
			reinterpret_cast<int32_t*>(matrix)[x] = src[x];
	}
	__syncthreads();
	if (!live)
		return;
	const Geom g = make_geom(it.query_len, it.target_len, it.d_begin, it.d_end);
	SeqView v{ args.qblock + q_off, args.tblock + t_off, c_off >= 0 ? args.cbs + c_off : own_matrix_biased(c_off) ? args.cbs + q_off : nullptr, matrix };
	if (REV) { v.rev_q = it.query_len - 1; v.rev_t = it.target_len - 1; }
	const int go = args.gap_open + args.gap_extend, ge = args.gap_extend;
	int bs, bi, bj, ba = 0, bb = 0;
	if constexpr (STAT == STAT_NONE) {
		// register-window sweep (swipe_core.h): per step pair one query letter, one bias byte and one target letter are
		// fetched, a pair ahead of their use; the cells themselves touch only VGPRs and the matrix in LDS
		WinLane<P, COORDS> st;
		win_init(st, g, v, lane);
		const int d0 = g.d_begin + 2 * P * lane;
		uint8_t* tbase = nullptr;
		if (TRACE)
			tbase = args.trace + args.trace_off[slot];
		const int PC = P * n_waves;                           // the item's band class: its trace layout (trace_byte_index)
		const bool has_cbs = v.cbs != nullptr;
		uint32_t tb_even[(P + 3) / 4], tb_odd[(P + 3) / 4];
		int t = 0;
		for (int a = g.a_first; a <= g.a_last; a += 2, ++t) {
			const uint32_t xi = (uint32_t)clampi(st.iq, g.qlen - 1), xj = (uint32_t)clampi(st.jt, g.tlen - 1);
			const int nq = v.q[xi], nt = v.t[xj], nc = has_cbs ? v.cbs[xi] : 0;      // consumed by win_advance at the end of the pair
			int nb = wave_shr1(st.F[2 * P - 1]);
			if (MULTI) {
				if ((threadIdx.x & 63) == 63) edge[0][0][wave] = st.F[2 * P - 1];
				__syncthreads();
				if ((threadIdx.x & 63) == 0 && wave > 0) nb = edge[0][0][wave - 1];
			}
			win_step<P, COORDS, TRACE, 0>(st, matrix, nb, go, ge, a, d0, tb_even);
			// the odd step may lie past a_last: all its cells are then invalid, and its trace row is allocated
			nb = wave_shl1(st.E[0]);
			if (MULTI) {
				if ((threadIdx.x & 63) == 0) edge[1][0][wave] = st.E[0];
				__syncthreads();
				if ((threadIdx.x & 63) == 63 && wave + 1 < n_waves) nb = edge[1][0][wave + 1];
			}
			win_step<P, COORDS, TRACE, 1>(st, matrix, nb, go, ge, a + 1, d0, tb_odd);
			if (TRACE) win_store_trace<P>(tbase + trace_byte_index(PC, t, lane * P), tb_even, tb_odd);
			win_advance(st, nq, nc, nt);
		}
		win_finish(st, d0);
		bs = st.best; bi = st.best_i; bj = st.best_j;
	}
	else {
		Lane<P, COORDS, STAT> st;
		st.init(g, lane);
		for (int a = g.a_first; a <= g.a_last; a += 2) {
			int nb = wave_shr1(st.F[2 * P - 1]), na = wave_shr1(st.st.Fa[2 * P - 1]), nbb = wave_shr1(st.st.Fb[2 * P - 1]);
			if (MULTI) {
				if ((threadIdx.x & 63) == 63) { edge[0][0][wave] = st.F[2 * P - 1]; edge[0][1][wave] = st.st.Fa[2 * P - 1]; edge[0][2][wave] = st.st.Fb[2 * P - 1]; }
				__syncthreads();
				if ((threadIdx.x & 63) == 0 && wave > 0) { nb = edge[0][0][wave - 1]; na = edge[0][1][wave - 1]; nbb = edge[0][2][wave - 1]; }
			}
			lane_step<P, COORDS, false, 0, STAT>(st, g, v, lane, a, nb, go, ge, nullptr, na, nbb);
			nb = wave_shl1(st.E[0]); na = wave_shl1(st.st.Ea[0]); nbb = wave_shl1(st.st.Eb[0]);
			if (MULTI) {
				if ((threadIdx.x & 63) == 0) { edge[1][0][wave] = st.E[0]; edge[1][1][wave] = st.st.Ea[0]; edge[1][2][wave] = st.st.Eb[0]; }
				__syncthreads();
				if ((threadIdx.x & 63) == 63 && wave + 1 < n_waves) { nb = edge[1][0][wave + 1]; na = edge[1][1][wave + 1]; nbb = edge[1][2][wave + 1]; }
			}
			lane_step<P, COORDS, false, 1, STAT>(st, g, v, lane, a + 1, nb, go, ge, nullptr, na, nbb);
		}
		bs = st.best; bi = st.best_i; bj = st.best_j; ba = st.best_a; bb = st.best_b;
	}

	// wave reduction of the end cell
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		const int os = __shfl_xor(bs, off), oi = __shfl_xor(bi, off), oj = __shfl_xor(bj, off);
		int oa = 0, ob = 0;
		if constexpr (STAT != STAT_NONE) { oa = __shfl_xor(ba, off); ob = __shfl_xor(bb, off); }
		if (COORDS ? better_end(os, oj, oi, bs, bj, bi) : os > bs) { bs = os; bi = oi; bj = oj; ba = oa; bb = ob; }
	}
	if (MULTI) {                                          // ... and over the wavefronts of the item
		if ((threadIdx.x & 63) == 0) { red[wave][0] = bs; red[wave][1] = bi; red[wave][2] = bj; red[wave][3] = ba; red[wave][4] = bb; }
		__syncthreads();
		if (threadIdx.x != 0) return;
		for (int w = 1; w < n_waves; ++w) {
			const int os = red[w][0], oi = red[w][1], oj = red[w][2];
			if (COORDS ? better_end(os, oj, oi, bs, bj, bi) : os > bs) { bs = os; bi = oi; bj = oj; ba = red[w][3]; bb = red[w][4]; }
		}
	}
	if ((threadIdx.x & 63) == 0) {
		SwipeEnd e;
		e.score = bs; e.end_i = bi; e.end_j = bj; e.stat_a = ba; e.stat_b = bb; e.pad[0] = e.pad[1] = e.pad[2] = 0;
		args.ends[item_idx] = e;
	}
}

// ---- traceback: one WAVEFRONT per item ---------------------------------------------------------------------------
// The serial walk (traceback_walk, swipe_core.h) is one dependent HBM/L2 load per alignment column. Alignments are
// mostly runs of (mis)matches along one diagonal, so the wave speculates: lane L loads the trace byte and scores the
// cell L diagonal steps back from the current cell; a ballot finds the length of the leading match run, a wave prefix
// sum of the cell scores finds the step at which the running score reaches the alignment score (the walk's stop
// condition, banded_swipe.h:128-183), and up to 64 columns are consumed per round trip. Gaps (rare, short) are walked
// serially by the whole wave in lockstep. Same outputs as traceback_walk, byte for byte.
__device__ __forceinline__ int wave_prefix_sum(int x, int lane)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const int y = __shfl_up(x, d);
		if (lane >= d) x += y;
	}
	return x;
}

__device__ __forceinline__ int popc64(unsigned long long x) { return __popcll(x); }

__device__ WalkResult traceback_walk_wave(const uint8_t* trace, const Geom& g, int P, const SeqView& v, int gap_open, int gap_extend,
	int best, int end_i, int end_j, uint8_t* transcript, int cap, int lane)
{
	WalkResult r;
	r.length = r.identities = r.mismatches = r.positives = r.gap_openings = r.gaps = 0;
	r.status = 0;
	int i = end_i, j = end_j, sc = 0, n = 0;                  // wave-uniform
	while (i >= 0 && j >= 0 && sc < best) {
		const int ii = i - lane, jj = j - lane;
		const bool valid = ii >= 0 && jj >= 0;
		const uint8_t m = valid ? trace_at(trace, g, P, ii, jj) : (uint8_t)TB_GAP_V;
		const bool is_match = valid && (m & (TB_GAP_V | TB_GAP_H)) == 0;
		int s = 0, ql = 0, tl = 1;
		bool positive = false;
		if (is_match) {
			ql = v.q[ii] & LETTER_MASK; tl = v.t[jj] & LETTER_MASK;
			s = v.M[tl * 32 + ql];
			positive = s > 0;
			if (v.cbs) s += v.cbs[ii];
		}
		const unsigned long long mm = __ballot(is_match);
		const int run = ~mm ? __builtin_ctzll(~mm) : 64;
		if (run > 0) {
			const int ps = wave_prefix_sum(lane < run ? s : 0, lane);
			// the walk tests `sc < best` before every column: column L is consumed iff the score before it is still below best
			const unsigned long long sm = __ballot(lane < run && sc + (ps - s) >= best);
			const int steps = sm ? imin(run, __builtin_ctzll(sm)) : run;       // >= 1: lane 0 sees sc < best
			const bool mine = lane < steps;
			if (transcript && mine && n + lane < cap - 1)
				transcript[cap - 2 - (n + lane)] = (uint8_t)(ql == tl ? (OP_MATCH << OP_COUNT_BITS) | 1 : (OP_SUBSTITUTION << OP_COUNT_BITS) | tl);
			const int ident = popc64(__ballot(mine && ql == tl)), pos_mm = popc64(__ballot(mine && ql != tl && positive));
			r.identities += ident; r.positives += ident + pos_mm; r.mismatches += steps - ident;
			r.length += steps;
			sc += __shfl(ps, steps - 1);
			n += steps; i -= steps; j -= steps;
			continue;
		}
		// gap at (i, j): lane 0's byte, walked by all lanes in lockstep (wave-uniform addresses)
		const uint8_t m0 = (uint8_t)__shfl((int)m, 0);
		int l = 0;
		if (m0 & TB_GAP_V) {
			do { ++l; --i; } while (i > 0 && (trace_at(trace, g, P, i, j) & TB_OPEN_V) == 0);
			int c = l;
			while (c > 0) {
				const int kk = imin(c, (int)OP_MAX_COUNT);
				if (transcript && lane == 0 && n < cap - 1) transcript[cap - 2 - n] = (uint8_t)((OP_INSERTION << OP_COUNT_BITS) | kk);
				++n; c -= kk;
			}
		}
		else {
			const int j_before = j;
			do { ++l; --j; } while (j > 0 && (trace_at(trace, g, P, i, j) & TB_OPEN_H) == 0);
			for (int x = lane; transcript && x < l; x += 64)
				if (n + x < cap - 1) transcript[cap - 2 - (n + x)] = (uint8_t)((OP_DELETION << OP_COUNT_BITS) | (v.t[j_before - x] & LETTER_MASK));
			n += l;
		}
		++r.gap_openings;
		r.length += l;
		r.gaps += l;
		sc -= gap_open + l * gap_extend;
	}
	if (sc != best) r.status = -6;           // DMND_E_TRACEBACK
	r.q_begin = i + 1; r.s_begin = j + 1; r.transcript_len = n;
	if (!transcript) return r;               // statistics only (the caller did not ask for transcripts)
	if (n > cap - 1) { r.status = -5; n = 0; }    // DMND_E_CAP
	// move to the front of the slot: 64 bytes per pass, every pass reads before it writes, destinations trail sources
	for (int x0 = 0; x0 < n; x0 += 64) {
		const int x = x0 + lane;
		uint8_t b = 0;
		if (x < n) b = transcript[cap - 1 - n + x];
		__builtin_amdgcn_wave_barrier();
		if (x < n) transcript[x] = b;
	}
	if (cap > 0 && lane == 0) transcript[n] = 0;
	r.q_begin = i + 1; r.s_begin = j + 1; r.transcript_len = n;
	return r;
}

__global__ __launch_bounds__(256) void traceback_kernel(TracebackArgs args)
{
	__shared__ int8_t matrix[32 * 32];
	for (int x = threadIdx.x; x < 32 * 32 / 4; x += blockDim.x)
		reinterpret_cast<int32_t*>(matrix)[x] = reinterpret_cast<const int32_t*>(args.matrix)[x];
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int64_t slot = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	if (slot >= args.n)
		return;
	const int32_t item_idx = args.order[slot];
	const dmnd_dp_target it = args.items[item_idx];
	const SwipeEnd e = args.ends[item_idx];
	dmnd_hsp h;
	h.score = e.score;
	h.q_begin = h.q_end = h.s_begin = h.s_end = 0;
	h.length = h.identities = h.mismatches = h.positives = h.gap_openings = h.gaps = 0;
	h.transcript_len = 0;
	h.transcript_off = args.transcript_off[slot];
	if (e.pad[0]) h.transcript_len = -1;          // saturated 16-bit sweep: nothing to walk, the host re-runs the item
	if (e.score > 0 && !e.pad[0]) {
		const Geom g = make_geom(it.query_len, it.target_len, it.d_begin, it.d_end);
		// an item with its own adjusted matrix is walked on that matrix, read where it lies in HBM (a few hundred look-ups)
		const SeqView v{ args.qblock + it.query_off, args.tblock + it.target_off,
			it.cbs_off >= 0 ? args.cbs + it.cbs_off : own_matrix_biased(it.cbs_off) ? args.cbs + it.query_off : nullptr,
			it.cbs_off <= -2 ? args.matrices + own_matrix_number(it.cbs_off) * (32 * 32) : matrix };
		const int PC = args.p_of_slot[slot];
		const int cap = (int)(args.transcript_off[slot + 1] - args.transcript_off[slot]);
		const WalkResult r = traceback_walk_wave(args.trace + args.trace_off[slot], g, PC, v, args.gap_open, args.gap_extend,
			e.score, e.end_i, e.end_j, args.transcript ? args.transcript + args.transcript_off[slot] : nullptr, cap, lane);
		h.q_begin = r.q_begin; h.s_begin = r.s_begin; h.q_end = e.end_i + 1; h.s_end = e.end_j + 1;
		h.length = r.length; h.identities = r.identities; h.mismatches = r.mismatches; h.positives = r.positives;
		h.gap_openings = r.gap_openings; h.gaps = r.gaps; h.transcript_len = r.transcript_len;
		if (r.status != 0 && lane == 0)
			atomicMin(args.status, r.status);
	}
	else if (args.transcript && lane == 0 && args.transcript_off[slot + 1] > args.transcript_off[slot])
		args.transcript[args.transcript_off[slot]] = 0;
	if (lane == 0) args.hsps[item_idx] = h;
}

template<int P>
static hipError_t launch_p(int mode, const SwipeArgs& a, hipStream_t stream)
{
	const unsigned blocks = (unsigned)((a.n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
	if (blocks == 0)
		return hipSuccess;
	const dim3 grid(blocks), block(WAVES_PER_BLOCK * 64);
	switch (mode) {
	case K_SCORE: hipLaunchKernelGGL((banded_swipe_kernel<P, false, false>), grid, block, 0, stream, a); break;
	case K_COORDS: hipLaunchKernelGGL((banded_swipe_kernel<P, true, false>), grid, block, 0, stream, a); break;
	case K_TRACE: hipLaunchKernelGGL((banded_swipe_kernel<P, true, true>), grid, block, 0, stream, a); break;
	case K_STATS_FWD:
		if constexpr (P <= 16) hipLaunchKernelGGL((banded_swipe_kernel<P, true, false, STAT_FWD, false>), grid, block, 0, stream, a);
		else return hipErrorInvalidValue;
		break;
	case K_STATS_BWD_REV:
		if constexpr (P <= 16) hipLaunchKernelGGL((banded_swipe_kernel<P, true, false, STAT_BWD, true>), grid, block, 0, stream, a);
		else return hipErrorInvalidValue;
		break;
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

// band classes above the single-wavefront maximum: P = 32 * waves (16 * waves with statistics), waves = 2, 4, 8, 16
static hipError_t launch_multi(int P, int mode, const SwipeArgs& a, hipStream_t stream)
{
	if (a.n == 0) return hipSuccess;
	const bool stats = mode == K_STATS_FWD || mode == K_STATS_BWD_REV;
	const int waves = P / (stats ? 16 : 32);
	if (waves < 2 || waves > MULTI_MAX_WAVES || (waves & (waves - 1))) return hipErrorInvalidValue;
	const dim3 grid((unsigned)a.n), block((unsigned)waves * 64);
	if (std::getenv("DMND_TRACE")) std::fprintf(stderr, "banded swipe: %lld item(s) of band class %d on %d wavefronts each (mode %d)\n", (long long)a.n, P, waves, mode);
#define DMND_MULTI(MAXW)                                                                                                                        \
	switch (mode) {                                                                                                                             \
	case K_SCORE: hipLaunchKernelGGL((banded_swipe_kernel<32, false, false, STAT_NONE, false, true, MAXW>), grid, block, 0, stream, a); break;  \
	case K_COORDS: hipLaunchKernelGGL((banded_swipe_kernel<32, true, false, STAT_NONE, false, true, MAXW>), grid, block, 0, stream, a); break;  \
	case K_TRACE: hipLaunchKernelGGL((banded_swipe_kernel<32, true, true, STAT_NONE, false, true, MAXW>), grid, block, 0, stream, a); break;    \
	case K_STATS_FWD: hipLaunchKernelGGL((banded_swipe_kernel<16, true, false, STAT_FWD, false, true, MAXW>), grid, block, 0, stream, a); break; \
	case K_STATS_BWD_REV: hipLaunchKernelGGL((banded_swipe_kernel<16, true, false, STAT_BWD, true, true, MAXW>), grid, block, 0, stream, a); break; \
	default: return hipErrorInvalidValue;                                                                                                       \
	}
	if (waves <= 4) { DMND_MULTI(4) }
	else { DMND_MULTI(MULTI_MAX_WAVES) }
#undef DMND_MULTI
	return hipGetLastError();
}

hipError_t launch_banded_swipe(int P, int mode, const SwipeArgs& a, hipStream_t stream)
{
	const bool stats = mode == K_STATS_FWD || mode == K_STATS_BWD_REV;
	if (P > (stats ? 16 : 32)) return launch_multi(P, mode, a, stream);
	switch (P) {
	case 1: return launch_p<1>(mode, a, stream);
	case 2: return launch_p<2>(mode, a, stream);
	case 4: return launch_p<4>(mode, a, stream);
	case 8: return launch_p<8>(mode, a, stream);
	case 16: return launch_p<16>(mode, a, stream);
	case 32: return launch_p<32>(mode, a, stream);   // statistics variants exist up to P = 16 per wavefront
	default: return hipErrorInvalidValue;
	}
}

hipError_t launch_traceback(const TracebackArgs& a, hipStream_t stream)
{
	if (a.n == 0)
		return hipSuccess;
	const unsigned waves = 4, blocks = (unsigned)((a.n + waves - 1) / waves);
	hipLaunchKernelGGL(traceback_kernel, dim3(blocks), dim3(waves * 64), 0, stream, a);
	return hipGetLastError();
}

}  // namespace dmnd

// dmnd_init: the first launch of a kernel of this translation unit loads its code object onto the device
namespace { __global__ void touch_swipe_kernel() {} }
extern "C" hipError_t dmnd_touch_swipe(hipStream_t st) { hipLaunchKernelGGL(touch_swipe_kernel, dim3(1), dim3(64), 0, st); return hipGetLastError(); }
