// bias_kernels.hip -- Hauser composition bias of a whole query block on the GPU.
// Replaces HauserCorrection (/root/reference/src/stats/hauser_correction.cpp:28-107) as called per query by
// Extension::extend (src/align/extend.cpp:252-262). One workgroup per sequence, one thread per position: the position's
// window is known in closed form (bias_core.h), so no running sums are carried along the sequence -- every thread sums
// the <= W + 1 matrix entries of its own window (letters through L1, matrix in LDS). 3.0e6 query letters of the C2
// block are ~1.2e8 LDS reads: tens of microseconds, against ~11 ms of host CPU time for the running-sum loop.
#include <hip/hip_runtime.h>
#include "bias_core.h"
#include "bias_kernels.h"
#include "xdrop_core.h"

namespace dmnd {

__global__ __launch_bounds__(256) void hauser_bias_kernel(BiasArgs a)
{
	__shared__ int8_t M[32 * 32];
	__shared__ float bg[20];
	for (int x = threadIdx.x; x < 32 * 32 / 4; x += blockDim.x)
		reinterpret_cast<int32_t*>(M)[x] = reinterpret_cast<const int32_t*>(a.matrix)[x];
	if (threadIdx.x < 20) bg[threadIdx.x] = a.bg[threadIdx.x];
	__syncthreads();
	for (int64_t k = blockIdx.x; k < a.n_seqs; k += gridDim.x) {
		const int64_t s = a.ids ? a.ids[k] : k;
		const int64_t begin = a.limits[s];
		const int l = (int)(a.limits[s + 1] - begin - 1);
		const int8_t* seq = a.block + begin;
		for (int m = threadIdx.x; m < l; m += blockDim.x)
			a.out[begin + m] = hauser_at(seq, l, m, M, bg, a.window);
	}
}

// x-drop ungapped extension of every seed hit of a block pair (ungapped_stage's inner call, align/ungapped.cpp:62-126): one thread
// per hit, both directions along the hit's diagonal; letters and bias come from the resident blocks, the matrix from LDS. The host's
// chaining stage did this per hit with two dependent reads per step into a target block of hundreds of MB (about half of its ~1 us
// per hit); here the walks of all hits of the block run at once while the host is still grouping the hits by target.
// xdrop_walk_core (xdrop_core.h) with the letters fetched sixteen at a time: the byte-by-byte walk is a chain of dependent L2 reads
// (three per step: query letter, target letter, bias), 0.37 ms for the 20 k seed hits of a C2 batch with every wavefront waiting for its
// longest walk. Same sums in the same order; the chunk may reach past the delimiter that ends the walk, never past the block's
// 256-byte perimeter.
__device__ __forceinline__ int xdrop_walk_chunked(const int8_t* M, const int8_t* q, const int8_t* cbs, const int8_t* t, int dir, int best, int xdrop, int& reach)
{
	reach = 0;
	int sum = best, n = 1;
	for (;;) {
		uint32_t qw[4], tw[4], cw[4] = { 0, 0, 0, 0 };
		// the sixteen letters from the current one on in walking direction: memory [p, p + 16) forward, [p - 15, p] backward
		const int back = dir < 0 ? 15 : 0;
		__builtin_memcpy(qw, q - back, 16);
		__builtin_memcpy(tw, t - back, 16);
		if (cbs) __builtin_memcpy(cw, cbs - back, 16);
#pragma unroll
		for (int j = 0; j < 16; ++j) {
			if (!(best - sum < xdrop)) return best;
			const int k = dir < 0 ? 15 - j : j;
			const int a = (int)((qw[k >> 2] >> (8 * (k & 3))) & 31u), b = (int)((tw[k >> 2] >> (8 * (k & 3))) & 31u);
			if (a == 31 || b == 31) return best;
			sum += (int)M[(a << 5) + b] + (int)(int8_t)(cw[k >> 2] >> (8 * (k & 3)));
			if (sum > best) { best = sum; reach = n; }
			++n;
		}
		q += 16 * dir; t += 16 * dir;
		if (cbs) cbs += 16 * dir;
	}
}

__global__ __launch_bounds__(256) void xdrop_seg_kernel(XdropArgs a)
{
	__shared__ int8_t M[32 * 32];
	for (int x = threadIdx.x; x < 32 * 32 / 4; x += blockDim.x)
		reinterpret_cast<int32_t*>(M)[x] = reinterpret_cast<const int32_t*>(a.matrix)[x];
	__syncthreads();
	const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= a.n_hits) return;
	const dmnd_seed_hit h = a.hits[k];
	const int64_t qoff = a.qlimits[h.query] + h.seed_offset;
	const int8_t* q = a.qblock + qoff;
	const int8_t* t = a.tblock + h.subject;
	const int8_t* cbs = a.cbs ? a.cbs + qoff : nullptr;
	int left, right;
	const int s_left = xdrop_walk_chunked(M, q - 1, cbs ? cbs - 1 : nullptr, t - 1, -1, 0, a.xdrop, left);
	const int s_both = xdrop_walk_chunked(M, q, cbs, t, +1, s_left, a.xdrop, right);
	a.out[k] = XdropSeg{ left, right, s_both };
}

hipError_t launch_xdrop_segs(const XdropArgs& a, hipStream_t st)
{
	if (a.n_hits <= 0) return hipSuccess;
	hipLaunchKernelGGL(xdrop_seg_kernel, dim3((unsigned)((a.n_hits + 255) / 256)), dim3(256), 0, st, a);
	return hipGetLastError();
}

hipError_t launch_hauser_bias(const BiasArgs& a, hipStream_t st)
{
	if (a.n_seqs <= 0) return hipSuccess;
	const unsigned blocks = (unsigned)(a.n_seqs < 65536 ? a.n_seqs : 65536);
	hipLaunchKernelGGL(hauser_bias_kernel, dim3(blocks), dim3(256), 0, st, a);
	return hipGetLastError();
}

}  // namespace dmnd

// dmnd_init: the first launch of a kernel of this translation unit loads its code object onto the device
namespace { __global__ void touch_bias_kernel() {} }
extern "C" hipError_t dmnd_touch_bias(hipStream_t st) { hipLaunchKernelGGL(touch_bias_kernel, dim3(1), dim3(64), 0, st); return hipGetLastError(); }
