// gapped_kernels.h -- launch interface of the gapped filter kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/diamond_hip.h"
#include "gapped_core.h"

namespace dmnd {

struct GfArgs {
	GfParams p;
	const int8_t* qdata; const int8_t* tdata; const int8_t* cbs;     // blocks + Hauser bias (HBM); cbs is indexed like qdata
	const int64_t* qlimits; const int64_t* tlimits; int64_t n_targets;
	const int8_t* matrix;                                              // 32x32 int8
	const int32_t* cutoff1; const int32_t* cutoff2;                    // CutoffTable2D, [32][32]
	const dmnd_seed_hit* hits; int64_t n_hits;
	uint8_t* flags;                                                    // out: 1 = hit passes both stages
	int32_t* scores;                                                   // out (optional): f1, f2 per hit (f2 = -1 when stage 2 did not run)
};

hipError_t launch_gapped_filter(const GfArgs& a, hipStream_t st);
// The hits cut into units of consecutive hits of one query (units[u] = { first hit, end }): one workgroup per unit. width > 0: the
// query's score profile is built in LDS, 32 rows of `width` bytes (>= the longest query of these units + 2 * GF_PAD); width = 0:
// the matrix path of launch_gapped_filter.
enum { GF_PAD = 128, GF_UNIT_HITS = 64 };
hipError_t launch_gapped_filter_units(const GfArgs& a, const int2* units, int n_units, int width, hipStream_t st);

}  // namespace dmnd
