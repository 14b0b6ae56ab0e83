// seed_core.h -- per-thread arithmetic of the MI355X seed stage.
//
// What the stage computes (reference: Search::search_shape, /root/reference/src/search/stage0.cpp:101-217):
// all (query position, reference position) pairs whose spaced seeds are equal (the double-indexed seed-hit
// join), minus low-complexity seeds (seed_complexity.cpp:37-120), filtered by the 48-byte Hamming fingerprint
// (hamming/finger_print.h:59-96, kernel.h:29-75) and by the left-most-seed rule (left_most.h:30-108), emitted
// as Search::Hit records (hit.h:30-47).
//
// GPU-first formulation -- NOT the reference's two partitioned seed arrays + radix/hash
// join: the small side (queries) is indexed in an open-addressing table with per-seed linked lists of query
// positions; the large side (reference block) is streamed once, every position's seed is computed in registers
// and probed. Because every (q, s) decision depends only on the two sequence windows, the seed's index chunk
// and the time at which query letters were masked, all pairs are independent; the reference's sequential
// "index chunk" order is reproduced by comparing chunk ids (chunk of the seed's partition) and by storing,
// per query letter, the (shape, chunk) time at which mask_seeds set its SEED_MASK bit.
//
// These functions are shared by the HIP kernels (seed_kernels.hip) and the CPU emulator in tests/emu.
#pragma once
#include <stdint.h>
#include "swipe_core.h"      // DMND_HD, LETTER_MASK, imin/imax

namespace dmnd {

enum { SEED_MAX_SHAPES = 64, SEED_MAX_WEIGHT = 32, L_MASK = 23, L_STOP = 24, L_DELIM = 31, SEED_NEVER = 255 };
enum { SEED_SPACED = 0, SEED_HASHED = 1 };

// Seed-stage configuration: the globals the reference reads across the seam (shapes, Reduction::instance,
// config.*, Search::Config), SURVEY 8b.
struct SeedParams {
	int32_t n_shapes;
	int32_t shape_len[SEED_MAX_SHAPES], shape_weight[SEED_MAX_SHAPES];
	uint32_t shape_mask[SEED_MAX_SHAPES];                    // Shape::mask_ (bit i = care position i)
	int8_t shape_pos[SEED_MAX_SHAPES][SEED_MAX_WEIGHT];      // Shape::positions_
	int8_t reduction[32];                                    // Reduction::map_ on masked letters (23 = invalid)
	int32_t reduction_size;
	int32_t seedp_bits, index_chunks, hamming_filter_id;
	int32_t ungapped_window, left_most_interval;             // config.ungapped_window (48), config.left_most_interval (32)
	double seed_complexity_cut;
	// stage-2 ungapped window filter (stage2.h:43-63,107-113); use_ungapped = 0 <=> ungapped_evalue == 0
	int32_t use_ungapped, short_query_max_len, short_query_cutoff;
	int32_t cutoff_table[32];                                // CutoffTable::data_[bit_length(query_len)]
	int32_t tile_size, simd_lanes;                           // config.tile_size (1024); int8 lanes of the reference build (AVX2: 32)
	int32_t query_translated;                                // align_mode.query_translated (blastx): short-frame rules of stage2.h:51,58-63
	int32_t seed_encoding;                                   // SEED_SPACED (double-indexed algorithm) or SEED_HASHED (query-indexed)
	int32_t cutoff_table_short[32];                          // CutoffTable(ungapped_evalue_short): translated frames of 61..85 letters (stage2.h:51)
};

DMND_HD bool is_amino_acid(int l) { return l != L_MASK && l != L_DELIM && l != L_STOP; }

// Seed of the window starting at p (Shape::set_seed_reduced on reduced letters, shape.h:114-152): base-`size`
// Horner polynomial; invalid if a care position is X or '*', or if the window [p, p+len) leaves its sequence
// (contains a delimiter/padding byte) -- the reference only enumerates j <= L - len (seed_iterator.h:30-33).
DMND_HD bool seed_at(const SeedParams& c, int sid, const int8_t* p, uint64_t& out)
{
	uint64_t s = 0;
	const int len = c.shape_len[sid];
	for (int i = 0; i < len; ++i)
		if ((p[i] & LETTER_MASK) == L_DELIM) return false;
	for (int k = 0; k < c.shape_weight[sid]; ++k) {
		const int r = c.reduction[p[c.shape_pos[sid][k]] & LETTER_MASK];
		if (r == L_MASK) return false;
		s = s * (uint64_t)c.reduction_size + (uint64_t)r;
	}
	out = s;
	return true;
}

// Table key of a seed on the GPU. For shapes of length <= 16 over an alphabet of <= 15 classes the key is the window's
// class nibbles at the care positions (nibble i = class of letter i, 0 elsewhere): the reference stream obtains it with one
// funnel shift and one AND per window instead of a Horner polynomial over the care positions. It is an injective function
// of the seed, which is all the join needs; the seed VALUE (its low bits select the seed partition / index chunk) is
// recovered with seed_of_key where it matters (mask and pair kernels). Other shapes use the seed value itself as the key.
DMND_HD bool seed_nibble_mode(const SeedParams& c, int sid) { return c.shape_len[sid] <= 16 && c.reduction_size <= 15; }

// Seed of the query-indexed algorithm (HashedSeedIterator, search/seed_array/seed_iterator.h:161-198; the Murmur hash the
// reference applies on top is a bijection and, with the single index chunk of this mode, its partition is never looked at):
// the window's reduced letters under the shape mask, where
//   * the iterator stops at the first window of a sequence and at every window whose LAST letter is an amino acid;
//   * a mask / stop letter inside the window contributes 0 -- except among the first shape_len letters of the sequence, which
//     the iterator's constructor reduces blindly: Reduction::map_[X] = 23 does not fit its 4 bits and spills into the letter
//     before it.
// Same nibble layout as the spaced-seed key (nibble j = letter j of the window), so a window of amino acids has the same key
// in both modes. The position inside the sequence is found by looking back for the delimiter (at most shape_len bytes; blocks
// start with padding delimiters).
DMND_HD bool seed_key_hashed(const SeedParams& c, int sid, const int8_t* p, uint64_t& out)
{
	const int len = c.shape_len[sid];
	for (int i = 0; i < len; ++i)
		if ((p[i] & LETTER_MASK) == L_DELIM) return false;
	int index = len;                                          // position of p in its sequence, capped at len
	for (int k = 1; k <= len; ++k)
		if ((p[-k] & LETTER_MASK) == L_DELIM) { index = k - 1; break; }
	if (index != 0 && !is_amino_acid(p[len - 1] & LETTER_MASK)) return false;
	uint64_t v = 0;
	for (int j = 0; j < len; ++j) {
		const int l = p[j] & LETTER_MASK;
		const int r = (index + j < len || is_amino_acid(l)) ? c.reduction[l] : 0;
		v |= (uint64_t)(r & 15) << (4 * j);
		if ((r & 16) && j > 0) v |= (uint64_t)1 << (4 * (j - 1));
	}
	uint64_t care = 0;
	for (int k = 0; k < c.shape_weight[sid]; ++k) care |= (uint64_t)15 << (4 * c.shape_pos[sid][k]);
	out = v & care;
	return true;
}

DMND_HD bool seed_key_at(const SeedParams& c, int sid, const int8_t* p, uint64_t& out)
{
	if (c.seed_encoding == SEED_HASHED) return seed_key_hashed(c, sid, p, out);
	if (!seed_nibble_mode(c, sid)) return seed_at(c, sid, p, out);
	const int len = c.shape_len[sid];
	for (int i = 0; i < len; ++i)
		if ((p[i] & LETTER_MASK) == L_DELIM) return false;
	uint64_t k = 0;
	for (int j = 0; j < c.shape_weight[sid]; ++j) {
		const int pos = c.shape_pos[sid][j];
		const int r = c.reduction[p[pos] & LETTER_MASK];
		if (r == L_MASK) return false;
		k |= (uint64_t)r << (4 * pos);
	}
	out = k;
	return true;
}

DMND_HD uint64_t seed_of_key(const SeedParams& c, int sid, uint64_t key)
{
	if (!seed_nibble_mode(c, sid)) return key;
	uint64_t s = 0;
	for (int j = 0; j < c.shape_weight[sid]; ++j)
		s = s * (uint64_t)c.reduction_size + ((key >> (4 * c.shape_pos[sid][j])) & 15u);
	return s;
}

// Shape::set_seed on unreduced letters (shape.h:72-96), used by verify_hit
DMND_HD bool seed_unreduced(const SeedParams& c, int sid, const int8_t* p, uint64_t& out)
{
	uint64_t s = 0;
	for (int k = 0; k < c.shape_weight[sid]; ++k) {
		const int l = p[c.shape_pos[sid][k]] & LETTER_MASK;
		if (!is_amino_acid(l)) return false;
		s = s * (uint64_t)c.reduction_size + (uint64_t)c.reduction[l];
	}
	out = s;
	return true;
}

// Index chunk of a seed: Partition<SeedPartition>(2^seedp_bits, index_chunks) over seed & mask
// (basic/seed.h:35-51, util/algo/partition.h:25-55, stage0.cpp:104-121)
DMND_HD int seed_chunk(const SeedParams& c, uint64_t seed)
{
	const int parts = 1 << c.seedp_bits, chunks = c.index_chunks < parts ? c.index_chunks : parts;
	const int part = (int)(seed & (uint64_t)(parts - 1));
	const int size = parts / chunks, rem = parts % chunks;
	const int big = rem * (size + 1);
	return part < big ? part / (size + 1) : rem + (part - big) / size;
}

DMND_HD void chunk_range(const SeedParams& c, int chunk, int& lo, int& hi)
{
	const int parts = 1 << c.seedp_bits, chunks = c.index_chunks < parts ? c.index_chunks : parts;
	const int size = parts / chunks, rem = parts % chunks, b = chunk < rem ? chunk : rem;
	lo = b * (size + 1) + (chunk - b) * size;
	hi = lo + (chunk < rem ? size + 1 : size);
}

// seed_is_complex (seed_complexity.cpp:37-52); lnfact = ln(k!) rounded to 6 decimals (lib/blast/blast_seg.cpp:54)
DMND_HD bool seed_is_complex(const SeedParams& c, int sid, const int8_t* p)
{
	const double LNFACT[20] = { 0.000000, 0.000000, 0.693147, 1.791759, 3.178054, 4.787492, 6.579251, 8.525161, 10.604603,
		12.801827, 15.104413, 17.502308, 19.987214, 22.552164, 25.191221, 27.899271, 30.671860, 33.505073, 36.395445, 39.339884 };
	int count[20];
	for (int i = 0; i < 20; ++i) count[i] = 0;
	for (int k = 0; k < c.shape_weight[sid]; ++k) {
		const int l = p[c.shape_pos[sid][k]] & LETTER_MASK;
		if (l >= 20) return false;
		++count[c.reduction[l]];
	}
	double entropy = LNFACT[c.shape_weight[sid] < 20 ? c.shape_weight[sid] : 19];
	for (int i = 0; i < c.reduction_size; ++i) entropy -= LNFACT[count[i]];
	return entropy >= c.seed_complexity_cut;
}

// FingerPrint::match over [loc-16, loc+32) of masked letters (search/hamming/finger_print.h:59-96): number of equal letters.
// On the device the two 48-byte windows are read with three unaligned 16-byte loads each and compared four letters per
// 32-bit operation: this filter sees every joined (query, reference) position pair -- 1.3e9 of them per --sensitive block
// pair -- and 96 byte-granular gather loads per pair made the pair kernel bound by load issue.
DMND_HD int fingerprint_id(const int8_t* q, const int8_t* s)
{
#if defined(__HIP_DEVICE_COMPILE__)
	int n = 0;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		uint32_t a[4], b[4];
		__builtin_memcpy(a, q - 16 + 16 * k, 16);
		__builtin_memcpy(b, s - 16 + 16 * k, 16);
#pragma unroll
		for (int w = 0; w < 4; ++w) {
			const uint32_t d = (a[w] ^ b[w]) & 0x1f1f1f1fu;               // letter = low 5 bits of every byte
			const uint32_t nz = (d + 0x7f7f7f7fu) & 0x80808080u;         // bit 7 of a byte set <=> the letters differ (no carries: d bytes <= 0x1f)
			n += 4 - __builtin_popcount(nz);
		}
	}
	return n;
#else
	int n = 0;
	for (int i = -16; i < 32; ++i)
		n += (q[i] & LETTER_MASK) == (s[i] & LETTER_MASK);
	return n;
#endif
}

// Util::Seq::clip (util/sequence/sequence.h:30-40): delimiter-free stretch of [seq, seq+len) around seq+anchor
DMND_HD void clip_window(const int8_t* seq, int len, int anchor, int& begin, int& end)
{
	int b = 0, e = len;
	for (int i = 0; i < len; ++i)
		if (seq[i] == L_DELIM) {              // raw byte compare, as memchr does
			if (i >= anchor) { e = i; break; }
			b = i + 1;
		}
	begin = b; end = e;
}

// reduced_match (sse_dist.h:104-153) with Reduction::map8 / map8b (basic.cpp:267-297)
DMND_HD uint64_t reduced_match(const SeedParams& c, const int8_t* q, const int8_t* s, int len)
{
	uint64_t m = 0;
	for (int i = 0; i < len && i < 64; ++i) {
		const int lq = q[i] & LETTER_MASK, ls = s[i] & LETTER_MASK;
		const bool bq = lq == L_MASK || lq == L_STOP || lq == L_DELIM, bs = ls == L_MASK || ls == L_STOP || ls == L_DELIM;
		if (!bq && !bs && c.reduction[lq] == c.reduction[ls]) m |= 1ull << i;
	}
	return m;
}

// PatternMatcher::hit over the shape masks [0, n_patterns) (util/algo/pattern_matcher.h:23-63): bit i of the result is set iff
// some pattern has all its care positions set in h >> i, for i < len - min_len + 1. Evaluated for all i at once: bit i of
// AND_k (h >> pos_k) is exactly "every care position of the pattern is set at offset i" (h carries zeros above its length, as the
// reference's running h >>= 1 does), so a pattern costs one shift + AND per care position instead of a 32-step scan.
DMND_HD uint32_t pattern_hit(const SeedParams& c, int n_patterns, uint32_t h, uint32_t len)
{
	if (n_patterns == 0) return 0;
	uint32_t min_len = 32;
	for (int i = 0; i < n_patterns; ++i) {
		const uint32_t l = (uint32_t)c.shape_len[i];          // = 32 - clz(mask): a shape code starts and ends with '1'
		if (l < min_len) min_len = l;
	}
	if (len < min_len) return 0;
	const uint32_t end = len - min_len + 1;
	uint32_t r = 0;
	for (int p = 0; p < n_patterns; ++p) {
		uint32_t m = 0xffffffffu;
		for (int k = 0; k < c.shape_weight[p]; ++k) m &= h >> c.shape_pos[p][k];
		r |= m;
	}
	return end >= 32 ? r : r & ((1u << end) - 1);
}

DMND_HD bool verify_hit(const SeedParams& c, const int8_t* q, const int8_t* s, bool left, uint32_t match_mask, int sid, bool chunked, int lo, int hi)
{
	if (chunked && (c.shape_mask[sid] & match_mask) == c.shape_mask[sid]) {
		uint64_t seed;
		if (!seed_unreduced(c, sid, s, seed)) return false;
		const int part = (int)(seed & (uint64_t)((1 << c.seedp_bits) - 1));
		if (left && !(part < hi)) return false;          // current_range.lower_or_equal
		if (!left && !(part < lo)) return false;         // current_range.lower
	}
	return fingerprint_id(q, s) >= c.hamming_filter_id;
}

DMND_HD bool verify_hits(const SeedParams& c, uint32_t mask, const int8_t* q, const int8_t* s, bool left, uint32_t match_mask, int sid, bool chunked, int lo, int hi)
{
	for (int pos = 0; mask != 0 && pos < 32; ++pos) {
		if ((mask & 1u) && verify_hit(c, q + pos, s + pos, left, match_mask >> pos, sid, chunked, lo, hi)) return true;
		mask >>= 1;
	}
	return false;
}

// Query letters that carry the SEED_MASK bit at "time" t_now = sid * index_chunks + chunk: mask_time[] holds the
// earliest (shape, chunk) at which mask_seeds set the bit on that letter (SEED_NEVER = never).
DMND_HD uint64_t seed_mask_bits(const uint8_t* mask_time, int len, int t_now)
{
	uint64_t m = 0;
	for (int i = 0; i < len && i < 64; ++i)
		if (mask_time[i] <= t_now) m |= 1ull << i;
	return m;
}

// ungapped_window(query_len) (stage2.h:58-63): frames of translated reads no longer than 85 use their whole length
DMND_HD int stage2_window(const SeedParams& c, int query_len)
{
	return (c.query_translated && query_len <= 85) ? query_len : c.ungapped_window;
}

// search_query_offset + left_most_filter for one (query position, reference position) pair that passed the
// Hamming filter (stage2.h:74-154, left_most.h:62-108). q/s point at the seed positions inside the blocks,
// qmt at the query position's entry of mask_time[]. Returns true if the pair is kept (it is the left-most
// seed hit of its diagonal in index-chunk order).
// How the rule reads its letter windows: byte by byte (this form; the CPU emulator, the oracle-sized cases), or with a few wide
// loads and register arithmetic (seed_kernels.hip, LmWide) -- same values either way.
struct LmBytewise {
	DMND_HD void clip(const int8_t* seq, int len, int anchor, int& b, int& e) const { clip_window(seq, len, anchor, b, e); }
	DMND_HD void masks(const SeedParams& c, const int8_t* qq, const int8_t* ss, const uint8_t* mm, int w, int t_now, uint64_t& match, uint64_t& masked) const
	{
		match = reduced_match(c, qq, ss, w);
		masked = seed_mask_bits(mm, w, t_now);
	}
};

template<typename Windows>
DMND_HD bool left_most_pair_t(const Windows& win, const SeedParams& c, const int8_t* q, const uint8_t* qmt, const int8_t* s, int seed_offset, int sid, int chunk, int query_len)
{
	const int window = stage2_window(c, query_len);
	int cb, ce;
	win.clip(q - window, 2 * window, window, cb, ce);              // query_clipped, stage2.h:94
	const int window_left0 = window - cb, clipped_len = ce - cb;
	const int interval_mod = c.left_most_interval > 0 ? seed_offset % c.left_most_interval : window_left0;
	const int overhang = imax(window_left0 - interval_mod, 0);
	// left_most_filter(query_clipped + overhang, subject + overhang, window_left - overhang, ...)
	const int8_t* qd = q - window_left0 + overhang;                 // query.data()
	const uint8_t* md = qmt - window_left0 + overhang;
	const int8_t* sd = s - window_left0 + overhang;                 // subject
	const int qlen = clipped_len - overhang;
	const int so = window_left0 - overhang;                         // seed_offset inside the clipped window
	const int seed_len = c.shape_len[sid];
	const bool chunked = c.index_chunks > 1, first_shape = sid == 0;
	int lo, hi;
	chunk_range(c, chunk, lo, hi);
	const int t_now = sid * c.index_chunks + chunk;

	int d = imax(so - 16, 0), window_left = imin(16, so);
	const int8_t *qq = qd + d, *ss = sd + d;
	const uint8_t* mm = md + d;
	int w = imin(qlen - d, window_left + 1 + 32);
	int sb, se;
	win.clip(ss, w, window_left, sb, se);                            // subject_clipped
	w -= w - se;
	qq += sb; ss += sb; mm += sb; window_left -= sb; w -= sb;

	uint64_t match_mask, masked_bits;
	win.masks(c, qq, ss, mm, w, t_now, match_mask, masked_bits);
	const uint64_t query_seed_mask = ~masked_bits;
	const uint32_t len_left = (uint32_t)(window_left + seed_len - 1);
	const uint32_t match_mask_left = (uint32_t)(((1ull << len_left) - 1) & match_mask),
		query_mask_left = (uint32_t)(((1ull << len_left) - 1) & query_seed_mask);
	const uint32_t left_hit = pattern_hit(c, sid + 1, match_mask_left, len_left) & query_mask_left;
	if (first_shape && !chunked)
		return left_hit == 0 || !verify_hits(c, left_hit, qq, ss, true, match_mask_left, sid, chunked, lo, hi);
	const uint32_t len_right = (uint32_t)(w - window_left - 1),
		match_mask_right = (uint32_t)(match_mask >> (window_left + 1)), query_mask_right = (uint32_t)(query_seed_mask >> (window_left + 1));
	const uint32_t right_hit = pattern_hit(c, chunked ? sid + 1 : sid, match_mask_right, len_right) & query_mask_right;
	return (left_hit == 0 || !verify_hits(c, left_hit, qq, ss, true, match_mask_left, sid, chunked, lo, hi))
		&& (right_hit == 0 || !verify_hits(c, right_hit, qq + window_left + 1, ss + window_left + 1, false, match_mask_right, sid, chunked, lo, hi));
}

DMND_HD bool left_most_pair(const SeedParams& c, const int8_t* q, const uint8_t* qmt, const int8_t* s, int seed_offset, int sid, int chunk, int query_len)
{
	return left_most_pair_t(LmBytewise(), c, q, qmt, s, seed_offset, sid, chunk, query_len);
}

// Stage-2 ungapped window score: best running local score over `window` aligned letters
// (window_ungapped_best / ungapped_window, src/dp/ungapped_simd.cpp:32-87, dp/ungapped_align.cpp:244-258).
DMND_HD int ungapped_window_score(const int8_t* M, const int8_t* q, const int8_t* s, int window)
{
	int score = 0, st = 0;
	for (int n = 0; n < window; ++n) {
		st += M[(q[n] & LETTER_MASK) * 32 + (s[n] & LETTER_MASK)];
		st = imax(st, 0);
		score = imax(score, st);
	}
	return score;
}

// ungapped_cutoff (stage2.h:43-63): short queries use a fixed bit-score cutoff, others CutoffTable[bit_length(len)]
// (util/scores/cutoff_table.h:26-47). 0 <=> filter off.
DMND_HD int ungapped_cutoff(const SeedParams& c, int query_len)
{
	if (!c.use_ungapped) return 0;
	if (query_len <= c.short_query_max_len) return c.short_query_cutoff;
	int b = 0;
	for (uint32_t x = (uint32_t)query_len; x; x >>= 1) ++b;
	// 60 < len <= 85 of a translated query: CutoffTable(ungapped_evalue_short) (differs from the main table from --very-sensitive up)
	return (c.query_translated && query_len <= 85) ? c.cutoff_table_short[b] : c.cutoff_table[b];
}

// The reference scores Hamming survivors in SIMD batches (search_query_offset, stage2.h:74-154): per subject tile
// of `tile_size` joined positions (ascending position order), survivors are taken `simd_lanes` at a time, and a
// batch of >= 4 runs the int8 kernel whose scores saturate at 255 (ungapped_simd.cpp:69-87) while smaller batches
// run the scalar one. Returns the size of the batch the pair (q, sloc) lands in. Order-free restatement over the seed's
// joined reference positions in ascending order (locs[0..n)): the tile is found by the rank of sloc, the batch by the
// rank among the tile's Hamming survivors. Only needed when the exact score exceeds 255 (rare): such pairs are deferred
// by the pair kernel and resolved in a second pass over a sorted copy of the joined positions of the seeds concerned.
// locs[k] holds the position in its low 40 bits (the sort key of the second pass is seed slot << 40 | position).
DMND_HD int simd_batch_size_sorted(const SeedParams& c, const uint64_t* locs, int64_t n, const int8_t* tdata, const int8_t* q, int64_t sloc)
{
	const uint64_t LOC = ((uint64_t)1 << 40) - 1;
	int64_t lo = 0, hi = n;                                       // rank = number of positions < sloc
	while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)(locs[mid] & LOC) < sloc) lo = mid + 1; else hi = mid; }
	const int64_t rank = lo, T = c.tile_size;
	int64_t t_lo = 0, t_hi = n;
	if (T > 0 && n > T) { t_lo = rank / T * T; t_hi = t_lo + T < n ? t_lo + T : n; }
	int64_t L = 0, r = 0;
	for (int64_t k = t_lo; k < t_hi; ++k) {
		if (fingerprint_id(q, tdata + (int64_t)(locs[k] & LOC)) < c.hamming_filter_id) continue;
		++L; r += k < rank;
	}
	const int64_t lanes = c.simd_lanes, left = L - r / lanes * lanes;
	return (int)(left < lanes ? left : lanes);
}

// Hash of a seed for the query seed table: two independent 32-bit mixes (murmur3-style finalisers on 32-bit lanes).
// a: table slot and level-1 bitmap bits -- evaluated for EVERY reference position by the stream kernel, so it is kept to
// three 32-bit multiplies (v_mul_lo_u32 issues at a quarter of the VALU rate); b: level-2 bitmap, only evaluated for the
// positions that pass level 1. seed_hash = (b << 32) | a.
DMND_HD uint32_t seed_hash_a(uint64_t x)
{
	const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
	uint32_t a = lo * 0x9E3779B1u + hi * 0x85EBCA6Bu;
	a ^= a >> 15; a *= 0x2C1B3C6Du; a ^= a >> 13;
	return a;
}

DMND_HD uint32_t seed_hash_b(uint64_t x)
{
	const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
	uint32_t b = (lo ^ 0x68E31DA4u) * 0xCC9E2D51u + (hi + 0x1B873593u) * 0x27D4EB2Fu;
	b ^= b >> 16; b *= 0x85EBCA6Bu; b ^= b >> 13;
	return b;
}

DMND_HD uint64_t seed_hash(uint64_t x) { return ((uint64_t)seed_hash_b(x) << 32) | seed_hash_a(x); }

// Level-1 filter of the reference stream: a blocked Bloom filter over the query seeds, every seed's K bits inside ONE 32-bit word
// (one L2 request per probe). word = high bits of hash a scaled to the word count (any count, not only powers of two: the filter
// is sized to what an XCD's L2 holds next to the stream); bits = three 5-bit fields of a's low half (K = 2: the first two).
DMND_HD uint32_t bm1_word(uint32_t h, uint32_t words) { return (uint32_t)(((uint64_t)h * words) >> 32); }
// Key classes of the short-seed pipeline (round 4): one of eight, a cheap function of the table key alone. Class c owns the c-th
// eighth of the level-1 filter's words and of the table's slots, and the fused stream kernel runs one workgroup per (tile, class)
// with workgroup w taking class w mod 8 -- the XCD the hardware dispatches it to -- so that an XCD's L2 only ever sees its own eighth
// of the query side (seed_kernels.hip, seed_stream_fast_kernel).
// (an XOR fold of the key's 16 nibbles to three bits: full-rate integer operations only -- the stream evaluates it for every window and class)
DMND_HD uint32_t seed_class(uint64_t key)
{
	uint32_t x = (uint32_t)key ^ (uint32_t)(key >> 32);
	x ^= x >> 16;
	x ^= x >> 8;
	x ^= x >> 4;
	return (x ^ (x >> 3)) & 7u;
}
DMND_HD uint32_t bm1_bits(uint32_t h, uint32_t k3)
{
	const uint32_t two = (1u << (h & 31)) | (1u << ((h >> 5) & 31));
	return k3 ? two | (1u << ((h >> 10) & 31)) : two;
}

}  // namespace dmnd
