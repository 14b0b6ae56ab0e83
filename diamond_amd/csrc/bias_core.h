// bias_core.h -- Hauser composition bias of one query position, closed form (shared by bias_kernels.hip and tests/emu).
//
// The reference (HauserCorrection, /root/reference/src/stats/hauser_correction.cpp:28-107) slides a window over the
// sequence and keeps running sums; position m then reads the sum of the scores of residue r = seq[m] against every letter
// of the window that is current when m is emitted. The window as a function of m (l = sequence length, W = config.cbs_window,
// half = min(W / 2, l - 1)), following the five loops of the reference one by one:
//   grow     m <  e2                 [0, half + m + 1)                         e2 = min(W + 1 - half, l - half)
//   slide    e2 <= m < m4            [m - e2 + 1, half + m + 1)                m4 = l - half   (only if the grow phase filled W + 1)
//   shrink   m4 <= m < m4 + e4       [t4 + (m - m4) + 1, l)                    t4 = m4 - e2, e4 = min(l - m4, e2 - 1)
//   rest     m >= m4 + e4            the last window of the shrink phase (or of the phase before it when e4 = 0)
// bias(m) = int8( f < 0 ? f - 0.5 : f + 0.5 ),  f = background(r) - float(sum - score(r, r)) / (n - 1),  n = window length;
// letters other than the 20 standard residues get 0. Integer sums and one float division: bit-identical to the running
// sums of the reference (and of hauser_int8 in extend_host.hip, which restates its loops).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DMND_BIAS_HD __host__ __device__ __forceinline__
#else
#define DMND_BIAS_HD inline
#endif

namespace dmnd {

DMND_BIAS_HD void hauser_window(int l, int W, int m, int& a, int& b)
{
	const int half = W / 2 < l - 1 ? W / 2 : l - 1;
	const int e2 = W + 1 - half < l - half ? W + 1 - half : l - half;      // emits of the grow phase
	const int m4 = l - half;                                                // first position after grow + slide
	const int t4 = m4 - e2;
	int e4 = e2 - 1;                                                        // n after grow = half + e2; shrink runs while n > half + 1
	if (e4 > l - m4) e4 = l - m4;
	if (e4 < 0) e4 = 0;
	if (m < e2) { a = 0; b = half + m + 1; }
	else if (m < m4) { a = m - e2 + 1; b = half + m + 1; }
	else {
		const int k = m - m4 < e4 ? m - m4 + 1 : e4;                        // letters dropped from the left so far
		a = t4 + k; b = l;
	}
}

// seq: the sequence's letters (block encoding), M: 32x32 int8 score matrix, bg: background scores as float
DMND_BIAS_HD int8_t hauser_at(const int8_t* seq, int l, int m, const int8_t* M, const float* bg, int W)
{
	const int r = seq[m] & 31;
	float f = 0.0f;
	if (r < 20 && l > 1) {
		int a, b;
		hauser_window(l, W, m, a, b);
		int sum = 0;
		for (int j = a; j < b; ++j) {
			const int x = seq[j] & 31;
			sum += x < 32 ? M[x * 32 + r] : 0;
		}
		f = bg[r] - float(sum - (int)M[r * 32 + r]) / (unsigned)(b - a - 1);
	}
	return (int8_t)(f < 0.0f ? f - 0.5f : f + 0.5f);
}

}  // namespace dmnd
