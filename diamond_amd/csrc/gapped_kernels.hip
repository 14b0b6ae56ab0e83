// gapped_kernels.hip -- gapped filter on the MI355X: one 64-lane wavefront per seed hit, one lane per diagonal.
//
// Stage 1 scans 64 diagonals (lane k = diagonal d_begin + k) over <= 200 target columns, stage 2 (only for hits whose
// stage-1 value beats its cutoff) 128 diagonals (two per lane) over <= 2*window columns. The running scores live in
// VGPRs; the target letter of a column is wave-uniform (scalar load), the query letter and bias are byte loads that are
// consecutive across lanes. The 1-D diagonal combination (diag_alignment) runs on the scalar unit via v_readlane.
// The 32x32 matrix sits in LDS. Integer work, bounded by VALU issue; HBM traffic is the hit list (24 B in, 1 B out).
//
// Round 3: the hits arrive sorted by query (hundreds per query in the sensitive modes), so a workgroup takes a UNIT of up to 64
// consecutive hits of one query and first builds that query's score profile in LDS -- profile[target letter][query position],
// matrix score + Hauser bias, clamped, -1 outside the query: DP::make_profile8, what the reference's scan reads too. A cell is then
// one LDS byte read (consecutive across the lanes of a diagonal band) and four VALU operations, where the matrix path above spends
// about fifteen on letter extraction, bias, clamps and the range test. Queries too long for the LDS budget keep the matrix path.
#include <algorithm>
#include "gapped_kernels.h"

namespace dmnd {

namespace {

__device__ __forceinline__ int sread(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

__device__ __forceinline__ uint32_t load4(const int8_t* p) { uint32_t w; __builtin_memcpy(&w, p, 4); return w; }

__device__ __forceinline__ void diag_step(const int8_t* M, int l, uint32_t qw, uint32_t cw, int s, int i, int qlen, int& v, int& best)
{
	int sc = M[(l << 5) + ((qw >> (8 * s)) & 31)];
	const int bias = (int)(int8_t)(cw >> (8 * s));
	if (l < 20) sc = imax(imin(sc + bias, 127), -128);
	if ((unsigned)i >= (unsigned)qlen) sc = -1;              // profile padding
	v = imin(imax(v + sc, 0), 255);
	best = imax(best, v);
}

// NDIAG diagonals per lane (dg, dg + 64), matrix in LDS. Four columns per iteration: the target letters are one
// wave-uniform dword, the lane's query letters and bias bytes one (unaligned) dword each. Reads up to 127 bytes outside
// the query are inside the block's 256-byte perimeter padding (and the bias buffer's slack); their values are never
// used (the range test substitutes the padding score).
template<int NDIAG>
__device__ __forceinline__ void scan_diags_lds(const int8_t* M, const int8_t* __restrict__ q, int qlen, const int8_t* __restrict__ cbs,
	const int8_t* __restrict__ t, int dg, int j0, int j1, int& best0, int& best1)
{
	int v0 = 0, v1 = 0;
	best0 = 0; best1 = 0;
	int j = j0;
	for (; j + 4 <= j1; j += 4) {
		const uint32_t tw = load4(t + j);
		const uint32_t qw0 = load4(q + dg + j), cw0 = cbs ? load4(cbs + dg + j) : 0u;
		uint32_t qw1 = 0, cw1 = 0;
		if (NDIAG == 2) { qw1 = load4(q + dg + 64 + j); cw1 = cbs ? load4(cbs + dg + 64 + j) : 0u; }
#pragma unroll
		for (int s = 0; s < 4; ++s) {
			const int l = (int)((tw >> (8 * s)) & 0xff);
			diag_step(M, l, qw0, cw0, s, dg + j + s, qlen, v0, best0);
			if (NDIAG == 2) diag_step(M, l, qw1, cw1, s, dg + 64 + j + s, qlen, v1, best1);
		}
	}
	for (; j < j1; ++j) {
		const int l = (int)(uint8_t)t[j];
		const int i0 = dg + j;
		diag_step(M, l, (uint32_t)(uint8_t)q[i0], cbs ? (uint32_t)(uint8_t)cbs[i0] : 0u, 0, i0, qlen, v0, best0);
		if (NDIAG == 2) diag_step(M, l, (uint32_t)(uint8_t)q[i0 + 64], cbs ? (uint32_t)(uint8_t)cbs[i0 + 64] : 0u, 0, i0 + 64, qlen, v1, best1);
	}
}

__device__ __forceinline__ int64_t uniform64(int64_t x)
{
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)x >> 32));
	return (int64_t)(((uint64_t)hi << 32) | lo);
}

__global__ __launch_bounds__(256) void gapped_filter_kernel(const GfArgs a)
{
	__shared__ int8_t M[1024];
	for (int i = threadIdx.x; i < 1024; i += blockDim.x) M[i] = a.matrix[i];
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
	for (int64_t h = wave; h < a.n_hits; h += n_waves) {
		// every lane of the wave works on the same hit: pin its fields to scalar registers
		const uint32_t query = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.hits[h].query);
		const int hit_i = __builtin_amdgcn_readfirstlane(a.hits[h].seed_offset);
		const int64_t subject = uniform64(a.hits[h].subject);
		// target of the hit: last limits entry <= subject position (wave-uniform binary search)
		int64_t lo = 0, hi = a.n_targets;
		while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (a.tlimits[mid] <= subject) lo = mid; else hi = mid; }
		const int64_t t0 = a.tlimits[lo], q0 = a.qlimits[query];
		const int slen = (int)(a.tlimits[lo + 1] - t0 - 1), qlen = (int)(a.qlimits[query + 1] - q0 - 1);
		const int hit_j = (int)(subject - t0);
		const int8_t* q = a.qdata + q0; const int8_t* t = a.tdata + t0;
		const int8_t* cbs = a.p.use_cbs ? a.cbs + q0 : nullptr;
		// cutoffs use the length of the query's context 0 (gapped_filter.cpp:43: query_profile->length())
		int qlen0 = qlen;
		if (a.p.contexts > 1) {
			const uint32_t c0 = query / (uint32_t)a.p.contexts * (uint32_t)a.p.contexts;
			qlen0 = (int)(a.qlimits[c0 + 1] - a.qlimits[c0] - 1);
			if (qlen0 < 85) {                                     // GAPPED_FILTER_MIN_QLEN, extend.cpp:195,206: filter not applied
				if (lane == 0) { a.flags[h] = 1; if (a.scores) { a.scores[2 * h] = -1; a.scores[2 * h + 1] = -1; } }
				continue;
			}
		}
		const int b1 = bit_length32((uint32_t)qlen0), b2 = bit_length32((uint32_t)slen);
		int d, jb, je, j0, j1;
		hit_window(hit_i, hit_j, slen, 64, 100, d, jb, je);
		scan_range(qlen, d, 64, jb, je, j0, j1);
		int s1, sa, sb;
		scan_diags_lds<1>(M, q, qlen, cbs, t, d + lane, j0, j1, s1, sb);
		DiagAln al;
		al.init(a.p);
		for (int i = 0; i < 64; ++i) al.step(a.p, sread(s1, i), i);
		const int f1 = al.best;
		int f2 = -1;
		const bool stage1_only = a.p.contexts > 1 && qlen0 < 100;       // MIN_STAGE2_QLEN, gapped_filter.cpp:44,54
		if (f1 > a.cutoff1[b1 * 32 + b2] && !stage1_only) {
			hit_window(hit_i, hit_j, slen, 128, a.p.window2, d, jb, je);
			scan_range(qlen, d, 128, jb, je, j0, j1);
			scan_diags_lds<2>(M, q, qlen, cbs, t, d + lane, j0, j1, sa, sb);
			al.init(a.p);
			for (int i = 0; i < 64; ++i) al.step(a.p, sread(sa, i), i);
			for (int i = 0; i < 64; ++i) al.step(a.p, sread(sb, i), 64 + i);
			f2 = al.best;
		}
		if (lane == 0) {
			a.flags[h] = (uint8_t)(stage1_only ? f1 > a.cutoff1[b1 * 32 + b2] : (f2 >= 0 && f2 > a.cutoff2[b1 * 32 + b2]));
			if (a.scores) { a.scores[2 * h] = f1; a.scores[2 * h + 1] = f2; }
		}
	}
}

// NDIAG diagonals per lane (dg, dg + 64) on the LDS profile: row = prof + l * W + GF_PAD, the lane reads row[dg + j]
template<int NDIAG>
__device__ __forceinline__ void scan_diags_profile(const int8_t* prof, int W, const int8_t* __restrict__ t, int dg, int j0, int j1, int& best0, int& best1)
{
	int v0 = 0, v1 = 0;
	best0 = 0; best1 = 0;
	const int8_t* mine = prof + GF_PAD + dg;                        // + l * W + j per column
	int j = j0;
	uint32_t tw = j + 4 <= j1 ? load4(t + j) : 0u;
	for (; j + 4 <= j1; j += 4) {
		const uint32_t tw_next = j + 8 <= j1 ? load4(t + j + 4) : 0u;      // the next four target letters are on their way during this step
		int sc0[4], sc1[4];
#pragma unroll
		for (int s = 0; s < 4; ++s) {
			const uint32_t l = (tw >> (8 * s)) & 31u;
			const int8_t* cell = mine + (__umul24(l, (uint32_t)W) + (uint32_t)(j + s));     // 24-bit multiply: full rate (v_mul_lo_u32 issues at a quarter)
			sc0[s] = cell[0];
			if (NDIAG == 2) sc1[s] = cell[64];
		}
#pragma unroll
		for (int s = 0; s < 4; ++s) {
			v0 = imin(imax(v0 + sc0[s], 0), 255); best0 = imax(best0, v0);
			if (NDIAG == 2) { v1 = imin(imax(v1 + sc1[s], 0), 255); best1 = imax(best1, v1); }
		}
		tw = tw_next;
	}
	for (; j < j1; ++j) {
		const uint32_t l = (uint32_t)(uint8_t)t[j] & 31u;
		const int8_t* cell = mine + (__umul24(l, (uint32_t)W) + (uint32_t)j);
		v0 = imin(imax(v0 + cell[0], 0), 255); best0 = imax(best0, v0);
		if (NDIAG == 2) { v1 = imin(imax(v1 + cell[64], 0), 255); best1 = imax(best1, v1); }
	}
}

// One workgroup per unit = up to 64 consecutive hits of ONE query (units[u] = first hit, end). PROFILE: the query's profile is built
// in LDS (32 rows of W = query length + 2 * GF_PAD bytes) and the scans read it; else the matrix path (queries longer than the LDS
// budget). Same per-hit logic as gapped_filter_kernel.
template<bool PROFILE>
__global__ __launch_bounds__(256) void gapped_filter_unit_kernel(const GfArgs a, const int2* __restrict__ units, int n_units, int W)
{
	extern __shared__ int8_t lds[];                                  // PROFILE: 32 x W profile; else the 32 x 32 matrix
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	if (!PROFILE) {
		for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = a.matrix[i];
		__syncthreads();
	}
	for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
		const int hb = units[u].x, he = units[u].y;
		const uint32_t query = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.hits[hb].query);
		const int64_t q0 = a.qlimits[query];
		const int qlen = (int)(a.qlimits[query + 1] - q0 - 1);
		const int8_t* q = a.qdata + q0;
		const int8_t* cbs = a.p.use_cbs ? a.cbs + q0 : nullptr;
		if (PROFILE) {
			__syncthreads();                                           // the previous unit's scans are done
			for (int x = threadIdx.x; x < W; x += blockDim.x) {
				const int i = x - GF_PAD;
				const bool inside = i >= 0 && i < qlen;
				const int ql = inside ? (q[i] & 31) : 0, bias = inside && cbs ? cbs[i] : 0;
#pragma unroll 8
				for (int l = 0; l < 32; ++l) {
					int sc = a.matrix[(l << 5) + ql];
					if (l < 20) sc = imax(imin(sc + bias, 127), -128);
					lds[l * W + x] = (int8_t)(inside ? sc : -1);
				}
			}
			__syncthreads();
		}
		int qlen0 = qlen;
		bool filter_off = false;
		if (a.p.contexts > 1) {
			const uint32_t c0 = query / (uint32_t)a.p.contexts * (uint32_t)a.p.contexts;
			qlen0 = (int)(a.qlimits[c0 + 1] - a.qlimits[c0] - 1);
			filter_off = qlen0 < 85;                                   // GAPPED_FILTER_MIN_QLEN, extend.cpp:195,206: filter not applied
		}
		const int b1 = bit_length32((uint32_t)qlen0);
		const bool stage1_only = a.p.contexts > 1 && qlen0 < 100;       // MIN_STAGE2_QLEN, gapped_filter.cpp:44,54
		for (int h = hb + wv; h < he; h += 4) {
			if (filter_off) {
				if (lane == 0) { a.flags[h] = 1; if (a.scores) { a.scores[2 * h] = -1; a.scores[2 * h + 1] = -1; } }
				continue;
			}
			const int hit_i = __builtin_amdgcn_readfirstlane(a.hits[h].seed_offset);
			const int64_t subject = uniform64(a.hits[h].subject);
			int64_t lo = 0, hi = a.n_targets;
			while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (a.tlimits[mid] <= subject) lo = mid; else hi = mid; }
			const int64_t t0 = a.tlimits[lo];
			const int slen = (int)(a.tlimits[lo + 1] - t0 - 1);
			const int hit_j = (int)(subject - t0);
			const int8_t* t = a.tdata + t0;
			const int b2 = bit_length32((uint32_t)slen);
			int d, jb, je, j0, j1;
			hit_window(hit_i, hit_j, slen, 64, 100, d, jb, je);
			scan_range(qlen, d, 64, jb, je, j0, j1);
			int s1, sa, sb;
			if (PROFILE) scan_diags_profile<1>(lds, W, t, d + lane, j0, j1, s1, sb);
			else scan_diags_lds<1>(lds, q, qlen, cbs, t, d + lane, j0, j1, s1, sb);
			DiagAln al;
			al.init(a.p);
			for (int i = 0; i < 64; ++i) al.step(a.p, sread(s1, i), i);
			const int f1 = al.best;
			int f2 = -1;
			if (f1 > a.cutoff1[b1 * 32 + b2] && !stage1_only) {
				hit_window(hit_i, hit_j, slen, 128, a.p.window2, d, jb, je);
				scan_range(qlen, d, 128, jb, je, j0, j1);
				if (PROFILE) scan_diags_profile<2>(lds, W, t, d + lane, j0, j1, sa, sb);
				else scan_diags_lds<2>(lds, q, qlen, cbs, t, d + lane, j0, j1, sa, sb);
				al.init(a.p);
				for (int i = 0; i < 64; ++i) al.step(a.p, sread(sa, i), i);
				for (int i = 0; i < 64; ++i) al.step(a.p, sread(sb, i), 64 + i);
				f2 = al.best;
			}
			if (lane == 0) {
				a.flags[h] = (uint8_t)(stage1_only ? f1 > a.cutoff1[b1 * 32 + b2] : (f2 >= 0 && f2 > a.cutoff2[b1 * 32 + b2]));
				if (a.scores) { a.scores[2 * h] = f1; a.scores[2 * h + 1] = f2; }
			}
		}
	}
}

}  // namespace

hipError_t launch_gapped_filter(const GfArgs& a, hipStream_t st)
{
	if (a.n_hits <= 0) return hipSuccess;
	const int64_t blocks = std::min<int64_t>((a.n_hits + 3) / 4, 256 * 32);
	gapped_filter_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(a);
	return hipGetLastError();
}

hipError_t launch_gapped_filter_units(const GfArgs& a, const int2* units, int n_units, int width, hipStream_t st)
{
	if (n_units <= 0) return hipSuccess;
	const unsigned blocks = (unsigned)std::min(n_units, 256 * 64);
	if (width > 0) gapped_filter_unit_kernel<true><<<dim3(blocks), dim3(256), (size_t)32 * (size_t)width, st>>>(a, units, n_units, width);
	else gapped_filter_unit_kernel<false><<<dim3(blocks), dim3(256), 1024, st>>>(a, units, n_units, 0);
	return hipGetLastError();
}

}  // namespace dmnd

// dmnd_init: the first launch of a kernel of this translation unit loads its code object onto the device
namespace { __global__ void touch_gapped_kernel() {} }
extern "C" hipError_t dmnd_touch_gapped(hipStream_t st) { hipLaunchKernelGGL(touch_gapped_kernel, dim3(1), dim3(64), 0, st); return hipGetLastError(); }
