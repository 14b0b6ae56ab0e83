// gapped_core.h -- per-diagonal arithmetic of the gapped filter, shared by the HIP kernel
// (gapped_kernels.hip) and the CPU lane emulator (tests/emu/gapped_emu.cpp).
//
// Reference behaviour restated here (AVX2 build of the reference = the build the goldens were minted with):
//   Extension::gapped_filter      src/align/gapped_filter.cpp:33-63   per hit: 64 diagonals x +-100 columns, then 128 x +-window
//   DP::make_profile8             src/dp/score_profile.cpp:33-64      profile = matrix score (+ Hauser bias on rows < 20), padding -1
//   DP::scan_diags64/128          src/dp/scan_diags.cpp:128,30        biased int8: running score floored at 0, capped at 255
//   DP::diag_alignment            src/dp/scan_diags.cpp:277-300       1-D affine-gap combination of the diagonal scores
//   CutoffTable2D                 src/util/scores/cutoff_table.h:50-83
#pragma once
#include <stdint.h>
#include "swipe_core.h"      // DMND_HD, imin/imax

namespace dmnd {

struct GfParams {
	int32_t diag_score;          // config.gapped_filter_diag_score = rawscore(12 bits)
	int32_t gap_open, gap_extend;
	int32_t window2;             // config.gapped_filter_window (200)
	int32_t use_cbs;
	int32_t contexts;            // align_mode.query_contexts; > 1 = translated queries (rules of extend.cpp:206, gapped_filter.cpp:43,54)
};

// Column range the scan runs over, equal for all diagonals of the band (out-of-query cells score the padding value -1)
DMND_HD void scan_range(int qlen, int d_begin, int band, int j_begin, int j_end, int& j0, int& j1)
{
	j0 = imax(j_begin, -(d_begin + band - 1));
	j1 = imin(qlen - d_begin, j_end);
}

// profile value of target letter l at query position i
DMND_HD int profile_score(const int8_t* M, const int8_t* q, int qlen, const int8_t* cbs, int l, int i)
{
	if (i < 0 || i >= qlen) return -1;
	int s = M[(l << 5) + (q[i] & 31)];
	if (cbs && l < 20) s = imax(imin(s + cbs[i], 127), -128);
	return s;
}

// best local ungapped score on diagonal dg over target columns [j0, j1)
DMND_HD int scan_diag(const int8_t* M, const int8_t* q, int qlen, const int8_t* cbs, const int8_t* t, int dg, int j0, int j1)
{
	int v = 0, best = 0;
	for (int j = j0; j < j1; ++j) {
		v = imin(imax(v + profile_score(M, q, qlen, cbs, t[j], dg + j), 0), 255);
		best = imax(best, v);
	}
	return best;
}

struct DiagAln {
	int best, best_gap, d;
	DMND_HD void init(const GfParams& p) { best = 0; best_gap = -p.gap_open; d = -1; }
	DMND_HD void step(const GfParams& p, int s, int i)
	{
		if (s < p.diag_score) return;
		const int gap_score = -p.gap_extend * (i - d) + best_gap;
		int n = s;
		if (gap_score + s > best) best = n = gap_score + s;
		if (s > best) best = n = s;
		const int open_score = -p.gap_open + n;
		if (open_score > gap_score) { best_gap = open_score; d = i; }
	}
};

// geometry of one filter stage for a hit (i, j) on a target of length slen
DMND_HD void hit_window(int hit_i, int hit_j, int slen, int band, int window, int& d_begin, int& j_begin, int& j_end)
{
	d_begin = imax(hit_i - hit_j - band / 2, -(slen - 1));
	j_begin = imax(hit_j - window, 0);
	j_end = imin(hit_j + window, slen);
}

DMND_HD int bit_length32(uint32_t x) { int b = 0; while (x) { ++b; x >>= 1; } return b; }

}  // namespace dmnd
