// xdrop_core.h -- one direction of the x-drop ungapped extension, shared by the host's chaining stage
// (chain_graph.h) and the device kernel that runs it for every seed hit of a block at once (bias_kernels.hip, xdrop_seg_kernel).
// Reference: xdrop_ungapped, /root/reference/src/dp/ungapped_align.cpp:151-199.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DMND_XD __host__ __device__ inline
#else
#define DMND_XD inline
#endif

namespace dmnd {

// From the letters at q / t in steps of dir, letter scores (+ bias) are added to a running sum that starts at `best`; the walk stops
// at a delimiter (letter 31) or when the sum has dropped xdrop below the best. Returns the best sum and, in `reach`, how many
// letters it took to get there. M: 32 x 32 scores, [query letter * 32 + target letter]. Reads one position beyond either end of a
// sequence: the blocks carry delimiters there.
template<typename Score>
DMND_XD int xdrop_walk_core(const Score* M, const int8_t* q, const int8_t* cbs, const int8_t* t, int dir, int best, int xdrop, int& reach)
{
	reach = 0;
	int sum = best;
	for (int n = 1; best - sum < xdrop; ++n, q += dir, t += dir, cbs += cbs ? dir : 0) {
		const int a = *q & 31, b = *t & 31;
		if (a == 31 || b == 31) break;
		sum += (int)M[(a << 5) + b] + (cbs ? (int)*cbs : 0);
		if (sum > best) { best = sum; reach = n; }
	}
	return best;
}

// what the kernel reports per seed hit (i, j): the segment is (i - left, j - left), length left + right, score
struct XdropSeg { int32_t left, right, score; };

}  // namespace dmnd
