// bias_kernels.h -- launch interface of the Hauser composition-bias kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/diamond_hip.h"
#include "xdrop_core.h"

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

struct XdropArgs {
	const int8_t* qblock;        // resident blocks (HBM)
	const int8_t* tblock;
	const int8_t* cbs;           // Hauser bias parallel to the query block, or NULL
	const int64_t* qlimits;      // sequence limits of the query block (HBM)
	const int8_t* matrix;        // 32x32 int8 (HBM)
	const dmnd_seed_hit* hits;   // the block pair's seed hits (HBM)
	int64_t n_hits;
	int xdrop;                   // config.raw_ungapped_xdrop
	XdropSeg* out;               // one record per hit
};
hipError_t launch_xdrop_segs(const XdropArgs& a, hipStream_t st);

}  // namespace dmnd
