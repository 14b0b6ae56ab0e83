// gapped_kernels.hip -- gapped filter on the MI355X: one 64-lane wavefront per seed hit, one lane per diagonal.
//
// Stage 1 scans 64 diagonals (lane k = diagonal d_begin + k) over <= 200 target columns, stage 2 (only for hits whose
// stage-1 value beats its cutoff) 128 diagonals (two per lane) over <= 2*window columns. The running scores live in
// VGPRs; the target letter of a column is wave-uniform (scalar load), the query letter and bias are byte loads that are
// consecutive across lanes. The 1-D diagonal combination (diag_alignment) runs on the scalar unit via v_readlane.
// The 32x32 matrix sits in LDS. Integer work, bounded by VALU issue; HBM traffic is the hit list (24 B in, 1 B out).
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

}  // namespace

hipError_t launch_gapped_filter(const GfArgs& a, hipStream_t st)
{
	if (a.n_hits <= 0) return hipSuccess;
	const int64_t blocks = std::min<int64_t>((a.n_hits + 3) / 4, 256 * 32);
	gapped_filter_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(a);
	return hipGetLastError();
}

}  // namespace dmnd

// dmnd_init: the first launch of a kernel of this translation unit loads its code object onto the device
namespace { __global__ void touch_gapped_kernel() {} }
extern "C" hipError_t dmnd_touch_gapped(hipStream_t st) { hipLaunchKernelGGL(touch_gapped_kernel, dim3(1), dim3(64), 0, st); return hipGetLastError(); }
