/* oracle/seed_search.c -- TEST INFRASTRUCTURE ONLY (parity checker for the seed stage; never used by the product).
 *
 * Plain-C restatement of the reference's double-indexed seed stage for spaced seeds, producing the stage-2 hit
 * multiset that Extension::extend receives (rows a2-a9 of SURVEY.md section 8):
 *   Shape::set_seed_reduced / set_seed                         src/basic/shape.h:72-152
 *   enum_seeds (SPACED_FACTOR encoding, no minimizers)         src/search/seed_array/enum_seeds.h:58-90
 *   seed partitions and index chunks                           src/basic/seed.h:35-51, util/algo/partition.h:25-55,
 *                                                              search/stage0.cpp:104-121
 *   the equi-join of query and reference seeds                 src/util/algo/hash_join.h:175-226 (any join yields the same groups)
 *   Search::mask_seeds / seed_is_complex                       src/search/seed_complexity.cpp:37-52,78-120
 *   FingerPrint (48-byte window, Hamming identity)             src/search/hamming/finger_print.h:59-96, kernel.h:29-75
 *   search_query_offset (window clipping, left-most interval)  src/search/stage2.h:74-154
 *   left_most_filter / verify_hit(s)                           src/search/left_most.h:30-108
 *   reduced_match / seed_mask                                  src/search/sse_dist.h:104-200
 *   PatternMatcher::hit                                        src/util/algo/pattern_matcher.h:23-63
 *   Util::Seq::clip                                            src/util/sequence/sequence.h:30-40
 *   window_ungapped_best / ungapped_window (stage-2 score)      src/dp/ungapped_simd.cpp:32-87, dp/ungapped_align.cpp:244-258
 *   ungapped_cutoff, CutoffTable                                src/search/stage2.h:43-63, util/scores/cutoff_table.h:26-47
 * Covered: with ungapped_evalue == 0 (the --fast family) no window scoring happens and Hit::score_ = 0xFFFF because the
 * reference leaves `scores[]` at INT_MAX (stage2.h:86,112); with ungapped_evalue > 0 (default, sensitive ...) the score
 * is the best local ungapped score of the clipped +-48 window, EXACT when fewer than 4 subjects share the SIMD batch and
 * saturated at 255 otherwise (batches = consecutive groups of `simd_lanes` Hamming survivors of one query position
 * inside one 1024x1024 tile, subjects in ascending location order) -- the AVX2 behaviour SURVEY.md section 7 fixes as canonical.
 * No self mode, no soft masking (run the reference with --masking 0 --motif-masking 0), no translated-query short rules.  Shapes and index chunks are processed in the reference's order so the
 * SEED_MASK bits written by mask_seeds are visible to later chunks exactly as in the reference.
 * Query-indexed algorithm (--algo 1, and what AUTO picks for small query sets against databases of 256 MB and more;
 * run/double_indexed.cpp:267-300), seed_encoding = 1:
 *   HashedSeedIterator                                         src/search/seed_array/seed_iterator.h:161-198
 *   enum_seeds_hashed (complexity filter + mask on the query)  src/search/seed_array/enum_seeds.h:125-153
 *   one index chunk; Search::mask_seeds does nothing            src/search/seed_complexity.cpp:81-82
 * A seed is the window's 4-bit reduced letters under Shape::long_mask (the Murmur hash on top is a bijection and, with one
 * chunk, its partition is irrelevant). Every window ending in an amino acid is enumerated, plus the first window of a
 * sequence; inside a window a mask / stop letter contributes 0, except among the first `length` letters of the sequence,
 * which the iterator's constructor reduces without looking (map_[X] = 23 spills a bit into the letter before).
 * Pinned by tests/test_oracle_seed.py against tests/golden/ext_*.tap (hits tapped at Extension::extend).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define LETTER_MASK 31
#define MASK_LETTER 23
#define STOP_LETTER 24
#define DELIMITER 31
#define SEED_MASK_BIT 0x80

typedef struct { uint64_t seed; int64_t loc; } entry_t;

static int cmp_entry(const void* a, const void* b)
{
	const entry_t* x = (const entry_t*)a; const entry_t* y = (const entry_t*)b;
	if (x->seed != y->seed) return x->seed < y->seed ? -1 : 1;
	return x->loc < y->loc ? -1 : (x->loc > y->loc ? 1 : 0);
}

/* ln(k!) rounded to 6 decimals, the table the reference shares with SEG (src/lib/blast/blast_seg.cpp:54) */
static const double LNFACT[20] = { 0.000000, 0.000000, 0.693147, 1.791759, 3.178054, 4.787492, 6.579251, 8.525161, 10.604603,
	12.801827, 15.104413, 17.502308, 19.987214, 22.552164, 25.191221, 27.899271, 30.671860, 33.505073, 36.395445, 39.339884 };

static int is_amino_acid(int l) { return l != MASK_LETTER && l != DELIMITER && l != STOP_LETTER; }

/* Shape::set_seed_reduced on letters reduced by Reduction::reduce_seq (enum_seeds.h:71): invalid iff a care position
 * reduces to MASK_LETTER (X or '*'); value = base-`size` Horner polynomial of the reduced letters. */
static int seed_at(const oracle_seed_cfg* c, int sid, const int8_t* p, uint64_t* out)
{
	uint64_t s = 0;
	for (int k = 0; k < c->shape_weight[sid]; ++k) {
		const int r = c->reduction[p[c->shape_pos[sid][k]] & LETTER_MASK];
		if (r == MASK_LETTER) return 0;
		s = s * (uint64_t)c->reduction_size + (uint64_t)r;
	}
	*out = s;
	return 1;
}

/* Shape::set_seed on unreduced letters (shape.h:72-96) as used by verify_hit */
static int seed_unreduced(const oracle_seed_cfg* c, int sid, const int8_t* p, uint64_t* out)
{
	uint64_t s = 0;
	for (int k = 0; k < c->shape_weight[sid]; ++k) {
		const int l = p[c->shape_pos[sid][k]] & LETTER_MASK;
		if (!is_amino_acid(l)) return 0;
		s = s * (uint64_t)c->reduction_size + (uint64_t)c->reduction[l];
	}
	*out = s;
	return 1;
}

/* HashedSeedIterator: value of `last_ & long_mask` for the window starting at sequence index j (seq = first letter of the
 * sequence, len its length), 0 if the iterator does not stop at this window */
static int seed_hashed(const oracle_seed_cfg* c, int sid, const int8_t* seq, int64_t len, int64_t j, uint64_t* out)
{
	const int L = c->shape_len[sid];
	if (j + L > len) return 0;
	if (j > 0 && !is_amino_acid(seq[j + L - 1] & LETTER_MASK)) return 0;       /* operator++ only returns on an amino acid */
	uint64_t last = 0, long_mask = 0;
	/* the register holds the sequence from its start; letters before the window are shifted out or masked off, except
	   for the spill of an out-of-range value, which only reaches the letter before it */
	for (int64_t x = j; x < j + L; ++x) {
		const int l = seq[x] & LETTER_MASK;
		last <<= 4;
		if (x < L) last |= (uint64_t)c->reduction[l];                            /* constructor: reduced blindly (X, '*' -> 23) */
		else if (is_amino_acid(l)) last |= (uint64_t)c->reduction[l];
	}
	for (int k = 0; k < c->shape_weight[sid]; ++k) long_mask |= (uint64_t)15 << (4 * (L - 1 - c->shape_pos[sid][k]));
	*out = last & long_mask;
	return 1;
}

static int seed_is_complex(const oracle_seed_cfg* c, int sid, const int8_t* p)
{
	int count[20];
	memset(count, 0, sizeof(count));
	for (int k = 0; k < c->shape_weight[sid]; ++k) {
		const int l = p[c->shape_pos[sid][k]] & LETTER_MASK;
		if (l >= 20) return 0;
		++count[c->reduction[l]];
	}
	double entropy = LNFACT[c->shape_weight[sid]];
	for (int i = 0; i < c->reduction_size; ++i) entropy -= LNFACT[count[i]];
	return entropy >= c->seed_complexity_cut;
}

static int fingerprint_id(const int8_t* q, const int8_t* s)
{
	int n = 0;
	for (int i = -16; i < 32; ++i)
		n += (q[i] & LETTER_MASK) == (s[i] & LETTER_MASK);
	return n;
}

/* Util::Seq::clip: the delimiter-free stretch of [seq, seq+len) around seq+anchor */
static void clip(const int8_t* seq, int len, int anchor, const int8_t** b, const int8_t** e)
{
	const int8_t *a = seq + anchor, *begin = seq, *end = seq + len;
	for (;;) {
		const int8_t* p = (const int8_t*)memchr(begin, DELIMITER, (size_t)(end - begin));
		if (!p) { *b = begin; *e = end; return; }
		if (p >= a) { *b = begin; *e = p; return; }
		begin = p + 1;
	}
}

/* Reduction::map8 / map8b (basic.cpp:267-297): mask, stop and delimiter letters get different sentinels in the two
 * maps so they never compare equal; everything else compares by reduced class. */
static uint64_t reduced_match(const oracle_seed_cfg* c, const int8_t* q, const int8_t* s, int len)
{
	uint64_t m = 0;
	for (int i = 0; i < len && i < 64; ++i) {
		const int lq = q[i] & LETTER_MASK, ls = s[i] & LETTER_MASK;
		const int rq = (lq == MASK_LETTER || lq == STOP_LETTER || lq == DELIMITER) ? c->reduction_size : (c->reduction[lq] == MASK_LETTER ? 0 : c->reduction[lq]);
		const int rs = (ls == MASK_LETTER || ls == STOP_LETTER || ls == DELIMITER) ? c->reduction_size + 1 : (c->reduction[ls] == MASK_LETTER ? 0 : c->reduction[ls]);
		if (rq == rs) m |= 1ull << i;
	}
	return m;
}

static uint64_t seed_mask_bits(const int8_t* q, int len)
{
	uint64_t m = 0;
	for (int i = 0; i < len && i < 64; ++i)
		if (q[i] & SEED_MASK_BIT) m |= 1ull << i;
	return m;
}

/* PatternMatcher::hit over the shape masks [0, n_patterns) */
static uint32_t pattern_hit(const oracle_seed_cfg* c, int n_patterns, uint32_t h, uint32_t len)
{
	if (n_patterns == 0) return 0;          /* min_len_ stays 32: nothing can match (pattern_matcher.h:25,47) */
	uint32_t min_len = 32, max_len = 0;
	for (int i = 0; i < n_patterns; ++i) {
		uint32_t l = 0, m = c->shape_mask[i];
		while (m) { ++l; m >>= 1; }
		if (l < min_len) min_len = l;
		if (l > max_len) max_len = l;
	}
	if (len < min_len) return 0;
	const uint32_t suffix_mask = max_len >= 32 ? 0xffffffffu : ((1u << max_len) - 1), end = len - min_len + 1;
	uint32_t r = 0;
	for (uint32_t i = 0; i < end && i < 32; ++i) {
		const uint32_t w = h & suffix_mask;
		for (int p = 0; p < n_patterns; ++p)
			if ((w & c->shape_mask[p]) == c->shape_mask[p]) { r |= 1u << i; break; }
		h >>= 1;
	}
	return r;
}

typedef struct { int lo, hi; } range_t;      /* current_range: partitions [lo, hi) of this index chunk */

static int verify_hit(const oracle_seed_cfg* c, const int8_t* q, const int8_t* s, int left, uint32_t match_mask, int sid, int chunked, range_t r)
{
	if (chunked && (c->shape_mask[sid] & match_mask) == c->shape_mask[sid]) {
		uint64_t seed;
		if (!seed_unreduced(c, sid, s, &seed)) return 0;
		const int part = (int)(seed & ((1ull << c->seedp_bits) - 1));
		if (left && !(part < r.hi)) return 0;          /* lower_or_equal */
		if (!left && !(part < r.lo)) return 0;         /* lower */
	}
	return fingerprint_id(q, s) >= c->hamming_filter_id;
}

static int verify_hits(const oracle_seed_cfg* c, uint32_t mask, const int8_t* q, const int8_t* s, int left, uint32_t match_mask, int sid, int chunked, range_t r)
{
	for (int pos = 0; mask != 0 && pos < 32; ++pos) {
		if (mask & 1u) {
			if (verify_hit(c, q + pos, s + pos, left, match_mask >> pos, sid, chunked, r)) return 1;
		}
		mask >>= 1;
	}
	return 0;
}

static int left_most_filter(const oracle_seed_cfg* c, const int8_t* qdata, int qlen, const int8_t* subject, int seed_offset, int sid, int chunked, range_t r)
{
	const int seed_len = c->shape_len[sid], first_shape = sid == 0;
	int d = seed_offset - 16 > 0 ? seed_offset - 16 : 0, window_left = seed_offset < 16 ? seed_offset : 16;
	const int8_t *q = qdata + d, *s = subject + d;
	int window = qlen - d;
	if (window > window_left + 1 + 32) window = window_left + 1 + 32;
	const int8_t *cb, *ce;
	clip(s, window, window_left, &cb, &ce);
	window -= (int)((s + window) - ce);
	d = (int)(cb - s);
	q += d; s += d; window_left -= d; window -= d;

	const uint64_t match_mask = reduced_match(c, q, s, window), query_seed_mask = ~seed_mask_bits(q, window);
	const uint32_t len_left = (uint32_t)(window_left + seed_len - 1);
	const uint32_t match_mask_left = (uint32_t)(((1ull << len_left) - 1) & match_mask),
		query_mask_left = (uint32_t)(((1ull << len_left) - 1) & query_seed_mask);
	const uint32_t left_hit = pattern_hit(c, sid + 1, match_mask_left, len_left) & query_mask_left;
	if (first_shape && !chunked)
		return left_hit == 0 || !verify_hits(c, left_hit, q, s, 1, match_mask_left, sid, chunked, r);
	const uint32_t len_right = (uint32_t)(window - window_left - 1),
		match_mask_right = (uint32_t)(match_mask >> (window_left + 1)), query_mask_right = (uint32_t)(query_seed_mask >> (window_left + 1));
	const uint32_t right_hit = pattern_hit(c, chunked ? sid + 1 : sid, match_mask_right, len_right) & query_mask_right;
	return (left_hit == 0 || !verify_hits(c, left_hit, q, s, 1, match_mask_left, sid, chunked, r))
		&& (right_hit == 0 || !verify_hits(c, right_hit, q + window_left + 1, s + window_left + 1, 0, match_mask_right, sid, chunked, r));
}

/* ungapped_window: best local ungapped score over `window` aligned letters */
static int ungapped_window(const oracle_seed_cfg* c, const int8_t* q, const int8_t* s, int window)
{
	int score = 0, st = 0;
	for (int n = 0; n < window; ++n) {
		st += c->matrix[(q[n] & LETTER_MASK) * 32 + (s[n] & LETTER_MASK)];
		if (st < 0) st = 0;
		if (st > score) score = st;
	}
	return score;
}

static int ungapped_cutoff(const oracle_seed_cfg* c, int query_len)
{
	if (!c->use_ungapped) return 0;
	if (query_len <= c->short_query_max_len) return c->short_query_cutoff;
	int b = 0;
	for (unsigned x = (unsigned)query_len; x; x >>= 1) ++b;
	return (c->query_translated && query_len <= 85) ? c->cutoff_table_short[b] : c->cutoff_table[b];
}

/* query != 0 (hashed encoding only): the query side of enum_seeds_hashed drops low-complexity seeds and sets SEED_MASK on them */
static int64_t enumerate(const oracle_seed_cfg* c, int sid, int8_t* data, const int64_t* limits, int64_t n, range_t r, entry_t* out, int query)
{
	int64_t m = 0;
	const uint64_t pmask = (1ull << c->seedp_bits) - 1;
	for (int64_t i = 0; i < n; ++i) {
		const int64_t len = limits[i + 1] - limits[i] - 1;
		if (c->seed_encoding == 1) {
			for (int64_t j = 0; j + c->shape_len[sid] <= len; ++j) {
				uint64_t seed;
				if (!seed_hashed(c, sid, data + limits[i], len, j, &seed)) continue;
				if (query && !seed_is_complex(c, sid, data + limits[i] + j)) { data[limits[i] + j] |= (int8_t)SEED_MASK_BIT; continue; }
				out[m].seed = seed; out[m].loc = limits[i] + j; ++m;
			}
			continue;
		}
		for (int64_t j = 0; j + c->shape_len[sid] <= len; ++j) {
			uint64_t seed;
			if (!seed_at(c, sid, data + limits[i] + j, &seed)) continue;
			const int part = (int)(seed & pmask);
			if (part < r.lo || part >= r.hi) continue;
			out[m].seed = seed; out[m].loc = limits[i] + j; ++m;
		}
	}
	return m;
}

/* qdata is modified (SEED_MASK bits) during the call and restored before returning (double_indexed.cpp:212).
 * Returns the number of hits, or -1 if more than cap. Hits are emitted in processing order (no particular order). */
int64_t oracle_seed_search(const oracle_seed_cfg* c, int8_t* qdata, const int64_t* qlimits, int64_t nq,
	const int8_t* tdata, const int64_t* tlimits, int64_t nt, oracle_hit* hits, int64_t cap)
{
	const int64_t qraw = qlimits[nq], traw = tlimits[nt];
	entry_t* qe = (entry_t*)malloc(sizeof(entry_t) * (size_t)(qraw + 1));
	entry_t* te = (entry_t*)malloc(sizeof(entry_t) * (size_t)(traw + 1));
	int64_t n_hits = 0;
	const int parts = 1 << c->seedp_bits, chunks = c->index_chunks < parts ? c->index_chunks : parts;
	const int chunked = c->index_chunks > 1;
	for (int sid = 0; sid < c->n_shapes && n_hits >= 0; ++sid)
		for (int chunk = 0; chunk < chunks && n_hits >= 0; ++chunk) {
			/* Partition<SeedPartition>(parts, chunks) */
			const int size = parts / chunks, rem = parts % chunks, b = chunk < rem ? chunk : rem;
			range_t r;
			r.lo = b * (size + 1) + (chunk - b) * size;
			r.hi = r.lo + (chunk < rem ? size + 1 : size);
			const int64_t mq = enumerate(c, sid, qdata, qlimits, nq, r, qe, 1), mt = enumerate(c, sid, (int8_t*)tdata, tlimits, nt, r, te, 0);
			qsort(qe, (size_t)mq, sizeof(entry_t), cmp_entry);
			qsort(te, (size_t)mt, sizeof(entry_t), cmp_entry);
			/* pass 1: mask_seeds over every joined group of the chunk (before any search of the chunk, stage0.cpp:173) */
			for (int pass = 0; pass < 2 && n_hits >= 0; ++pass) {
				int64_t i = 0, j = 0;
				while (i < mq && j < mt) {
					if (qe[i].seed < te[j].seed) { ++i; continue; }
					if (qe[i].seed > te[j].seed) { ++j; continue; }
					int64_t i1 = i, j1 = j;
					while (i1 < mq && qe[i1].seed == qe[i].seed) ++i1;
					while (j1 < mt && te[j1].seed == te[j].seed) ++j1;
					if (pass == 0) {
						if (c->seed_encoding == 0 && !seed_is_complex(c, sid, qdata + qe[i].loc)) {
							for (int64_t x = i; x < i1; ++x) qdata[qe[x].loc] |= (int8_t)SEED_MASK_BIT;
							qe[i].seed = qe[i].seed;          /* group is erased: mark by negative loc on the first ref entry */
							te[j].loc = -te[j].loc - 1;
						}
					}
					else if (te[j].loc >= 0) {
						for (int64_t x = i; x < i1 && n_hits >= 0; ++x) {
							const int64_t qloc = qe[x].loc;
							/* query_data(): local position */
							int64_t lo = 0, hi = nq;
							while (hi - lo > 1) { const int64_t mid = (lo + hi) / 2; if (qlimits[mid] <= qloc) lo = mid; else hi = mid; }
							const int64_t query_id = lo;
							const int seed_offset = (int)(qloc - qlimits[query_id]);
							const int query_len = (int)(qlimits[query_id + 1] - qlimits[query_id] - 1);
							/* ungapped_window(query_len), stage2.h:58-63 */
							const int window = (c->query_translated && query_len <= 85) ? query_len : c->ungapped_window;
							const int8_t *cb, *ce;
							clip(qdata + qloc - window, window * 2, window, &cb, &ce);
							const int window_left = (int)((qdata + qloc) - cb), window_clipped = (int)(ce - cb);
							const int interval_mod = c->left_most_interval > 0 ? seed_offset % c->left_most_interval : window_left;
							const int overhang = window_left - interval_mod > 0 ? window_left - interval_mod : 0;
							const int cutoff = ungapped_cutoff(c, query_len);
							/* search_tile / search_query_offset: per S tile, Hamming survivors in batches of simd_lanes */
							const int64_t tile = c->tile_size > 0 ? c->tile_size : (j1 - j);
							for (int64_t tj = j; tj < j1 && n_hits >= 0; tj += tile) {
								const int64_t tj1 = tj + tile < j1 ? tj + tile : j1;
								int64_t* surv = (int64_t*)malloc(sizeof(int64_t) * (size_t)(tj1 - tj));
								int64_t ns = 0;
								for (int64_t y = tj; y < tj1; ++y)
									if (fingerprint_id(qdata + qloc, tdata + te[y].loc) >= c->hamming_filter_id) surv[ns++] = te[y].loc;
								const int64_t lanes = c->simd_lanes > 0 ? c->simd_lanes : 32;
								for (int64_t b0 = 0; b0 < ns && n_hits >= 0; b0 += lanes) {
									const int64_t nb = ns - b0 < lanes ? ns - b0 : lanes;
									for (int64_t y = b0; y < b0 + nb; ++y) {
										const int64_t sloc = surv[y];
										const int8_t* subject = tdata + sloc - window_left;
										int score = 0xFFFF;
										if (cutoff) {
											score = ungapped_window(c, cb, subject, window_clipped);
											if (nb >= 4 && score > 255) score = 255;
											if (score <= cutoff) continue;
										}
										if (!left_most_filter(c, cb + overhang, window_clipped - overhang, subject + overhang, window_left - overhang, sid, chunked, r))
											continue;
										if (n_hits >= cap) { n_hits = -1; break; }
										hits[n_hits].query = (uint32_t)query_id; hits[n_hits].subject = sloc;
										hits[n_hits].seed_offset = seed_offset; hits[n_hits].score = score;
										++n_hits;
									}
								}
								free(surv);
							}
						}
					}
					i = i1; j = j1;
				}
			}
		}
	for (int64_t x = 0; x < qraw; ++x) qdata[x] &= (int8_t)~SEED_MASK_BIT;    /* clear_masking is not needed for parity but keeps the input intact */
	free(qe); free(te);
	return n_hits;
}
