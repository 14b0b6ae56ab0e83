// mask_core.h -- per-lane arithmetic of tantan repeat masking (SURVEY 8f "masking"), shared by the HIP kernel
// (mask_kernels.hip) and its CPU lane emulator (tests/emu/mask_emu.cpp).
//
// Reference behaviour restated (the reference's AVX2 build, no FMA: CMakeLists.txt:240):
//   Util::tantan::mask, forward_step, backward_step     src/masking/tantan.cpp:45-215
//   SIMD::sum / scale / hsum                            src/util/simd/vector.h:37-67, vector8_avx2.h:132-139
// Single-precision results depend on the operation order, so the order is part of the contract: products and sums are
// rounded separately (compile WITHOUT floating-point contraction), the 50 repeat offsets are summed as six groups of 8
// consecutive offsets, each group as ((v0+v4)+(v1+v5))+((v2+v6)+(v3+v7)), groups accumulated in order, offsets 48 and 49
// added last. One lane owns one offset; SHFL(x, mask) must return x of lane (lane ^ mask).
#pragma once
#include "swipe_core.h"      // DMND_HD

namespace dmnd {

enum { TANTAN_WINDOW = 50 };

struct TantanParams {
	float p_repeat_end, b2b, f2f, p_mask;      // 0.05, 1 - 0.005, 1 - 0.05, config.tantan_minMaskProb (0.9)
	float d[TANTAN_WINDOW];                    // b2f0 * growth^(49 - k), computed on the host exactly as tantan.cpp:130-136
};

// forward_step for one offset: f' = (f * f2f + b_old * d) * e
DMND_HD float tantan_fwd_cell(float f, float f2f, float b_old, float d, float e)
{
	const float t1 = f * f2f, t2 = b_old * d;
	const float tmp = t1 + t2;
	return tmp * e;
}

// backward_step for one offset: vf = f * e; contribution vt = vf * d; f' = vf * f2f + C
DMND_HD float tantan_bwd_cell(float f, float e, float d, float f2f, float C, float& vt)
{
	const float vf = f * e;
	vt = vf * d;
	const float t1 = vf * f2f;
	return t1 + C;
}

// horizontal sum of each aligned group of 8 lanes, result in every lane of the group
template<typename Shfl>
DMND_HD float tantan_group_sum(float v, Shfl shfl)
{
	v = v + shfl(v, 4);
	v = v + shfl(v, 1);
	v = v + shfl(v, 2);
	return v;
}

}  // namespace dmnd
