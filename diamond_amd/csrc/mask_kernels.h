// mask_kernels.h -- launch interface of the tantan masking kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mask_core.h"

namespace dmnd {

struct TantanArgs {
	TantanParams p;
	int8_t* data;                 // block letters (HBM), masked in place
	const int64_t* limits;        // sequence i = data[limits[i], limits[i+1] - 1)
	int64_t n_seqs;
	const float* lr;              // 32x32 likelihood ratios (HBM)
	float* pb;                    // scratch: one float per block letter (indexed like data)
	float* scale;                 // scratch: limits[i] / 16 + i is the first slot of sequence i
	unsigned long long* n_masked; // out: number of letters masked
};

hipError_t launch_tantan(const TantanArgs& a, hipStream_t st);

}  // namespace dmnd
