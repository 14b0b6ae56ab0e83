// bias_kernels.h -- launch interface of the Hauser composition-bias kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dmnd {

struct BiasArgs {
	const int8_t* block;         // query block letters (HBM)
	const int64_t* limits;       // n_seqs + 1 sequence limits of the block (HBM)
	int64_t n_seqs;              // sequences to process
	const int32_t* ids;          // their block sequence ids (HBM), or NULL = the first n_seqs sequences of the block
	const int8_t* matrix;        // 32x32 int8 (HBM)
	float bg[20];                // ScoreMatrix::background_scores as float
	int window;                  // config.cbs_window
	int8_t* out;                 // bias, indexed like the block
};

hipError_t launch_hauser_bias(const BiasArgs& a, hipStream_t st);

}  // namespace dmnd
