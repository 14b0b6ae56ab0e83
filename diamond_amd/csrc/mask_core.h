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

// ---- motif soft masking -------------------------------------------------------------------------------------------------
// The reference soft-masks abundant 8-mer motifs while seeds are enumerated (mask_motifs, src/masking/masking.cpp:110-131;
// table: src/masking/motifs.cpp, ~8000 8-mers; Block::soft_mask / remove_soft_masking, src/data/block/block.cpp:164-177;
// enum_seeds, src/search/seed_array/enum_seeds.h:240-271): inside a sequence, every 8-mer of standard amino acids found
// in the table covers its 8 letters; covered stretches (touching ones merge: Mask::Ranges::push_back, masking/def.h:72-78) are
// replaced by the mask letter for the enumeration only -- unless the covered letters make up half of the sequence or more
// (then nothing is masked), and only stretches of at most max_motif_len (30) letters. The letters are restored before the
// filters and the extension run; on the query side the seed positions whose shape window touches a masked stretch keep
// a SEED_MASK bit (MaskingTable::remove, masking.cpp:89-102).
namespace dmnd {

enum { MOTIF_LEN = 8, MOTIF_MAX_RANGE = 30 };

// code of the 8-mer starting at p (Kmer<8>: base-20 polynomial of the letters), false if it holds a non-standard letter
DMND_HD bool motif_code_at(const int8_t* p, uint64_t& code)
{
	uint64_t c = 0;
	for (int i = 0; i < MOTIF_LEN; ++i) {
		const int l = p[i] & 31;
		if (l >= 20) return false;
		c = c * 20 + (uint64_t)l;
	}
	code = c;
	return true;
}

// sorted table lookup
DMND_HD bool motif_in_table(const uint64_t* table, int n, uint64_t code)
{
	int lo = 0, hi = n;
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		if (table[mid] < code) lo = mid + 1; else hi = mid;
	}
	return lo < n && table[lo] == code;
}

// One sequence: hit[x] != 0 iff the 8-mer starting at x is a motif (x + 8 <= len). Writes the mask letter over the covered
// stretches of `seq` that qualify; returns the number of covered letters (what the reference's statistics count).
DMND_HD int motif_mask_sequence(int8_t* seq, const uint8_t* hit, int len, int max_range)
{
	if (len < MOTIF_LEN) return 0;
	// covered[x] = a motif starts in [x - 7, x]
	int covered = 0, since = MOTIF_LEN;                   // since: distance to the last motif start at or before x
	for (int x = 0; x < len; ++x) {
		since = (x + MOTIF_LEN <= len && hit[x]) ? 0 : since + 1;
		covered += since < MOTIF_LEN;
	}
	if (2 * covered >= len) return 0;                      // (double) n / len >= 0.5
	since = MOTIF_LEN;
	int run_begin = -1;
	for (int x = 0; x <= len; ++x) {
		bool cov = false;
		if (x < len) {
			since = (x + MOTIF_LEN <= len && hit[x]) ? 0 : since + 1;
			cov = since < MOTIF_LEN;
		}
		if (cov && run_begin < 0) run_begin = x;
		if (!cov && run_begin >= 0) {
			if (x - run_begin <= max_range)
				for (int y = run_begin; y < x; ++y) seq[y] = 23;
			run_begin = -1;
		}
	}
	return covered;
}

}  // namespace dmnd
