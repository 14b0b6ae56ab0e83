/* TEST INFRASTRUCTURE ONLY (see oracle.h): CPU restatement of DIAMOND's gapped filter, the
 * per-target test the extension stage applies before chaining when gapped_filter_evalue > 0 (--sensitive and above).
 *
 *   gapped_filter (per hit / per target)     src/align/gapped_filter.cpp:33-63
 *   make_profile8 (AVX2 build semantics)     src/dp/score_profile.cpp:33-64   (Hauser bias only on rows l < 20, saturating add)
 *   scan_diags64 / scan_diags128             src/dp/scan_diags.cpp:128,30      (int8 biased saturating: floor 0, ceiling 255)
 *   diag_alignment                           src/dp/scan_diags.cpp:277-300
 *   CutoffTable2D                            src/util/scores/cutoff_table.h:50-83, ScoreMatrix::evalue_norm score_matrix.cpp:222
 *
 * Pinned against the genuine reference by tests/golden/gf_sens.tap (tap at Extension::gapped_filter, oracle/ref_tap.cpp):
 * surviving target sets and both cutoffs per (query, target). */
#include "oracle.h"

static int sat8(int x) { return x < -128 ? -128 : x > 127 ? 127 : x; }

/* Profile entry: score of target letter l against query position i (i outside the query = padding -1) */
static int profile_at(const int8_t* matrix8, const int8_t* query, int qlen, const int8_t* cbs, int l, int i)
{
	if (i < 0 || i >= qlen) return -1;
	int s = matrix8[(l << 5) + (query[i] & 31)];
	if (cbs && l < 20) s = sat8(s + cbs[i]);
	return s;
}

/* scan_diags<band>: out[k] = best local ungapped score (saturated at 255) on diagonal d_begin + k over target columns [j_begin, j_end) */
void oracle_scan_diags(const int8_t* matrix8, const int8_t* query, int qlen, const int8_t* cbs, const int8_t* target,
	int d_begin, int j_begin, int j_end, int band, int* out)
{
	int j0 = j_begin > -(d_begin + band - 1) ? j_begin : -(d_begin + band - 1);
	const int j1 = qlen - d_begin < j_end ? qlen - d_begin : j_end;
	int v[128];
	for (int k = 0; k < band; ++k) { v[k] = 0; out[k] = 0; }
	for (int j = j0; j < j1; ++j) {
		const int i = d_begin + j, l = target[j];
		for (int k = 0; k < band; ++k) {
			int x = v[k] + profile_at(matrix8, query, qlen, cbs, l, i + k);
			x = x < 0 ? 0 : x > 255 ? 255 : x;
			v[k] = x;
			if (x > out[k]) out[k] = x;
		}
	}
}

int oracle_diag_alignment(const int* s, int count, int diag_score, int gap_open, int gap_extend)
{
	int best = 0, best_gap = -gap_open, d = -1;
	for (int i = 0; i < count; ++i) {
		if (s[i] < diag_score) continue;
		const int gap_score = -gap_extend * (i - d) + best_gap;
		int n = s[i];
		if (gap_score + s[i] > best) best = n = gap_score + s[i];
		if (s[i] > best) best = n = s[i];
		const int open_score = -gap_open + n;
		if (open_score > gap_score) { best_gap = open_score; d = i; }
	}
	return best;
}

int oracle_gapped_filter_hit(const int8_t* matrix8, const int8_t* query, int qlen, const int8_t* cbs, const int8_t* target, int slen,
	int hit_i, int hit_j, int band, int window, int diag_score, int gap_open, int gap_extend)
{
	int d = hit_i - hit_j - band / 2;
	if (d < -(slen - 1)) d = -(slen - 1);
	const int j0 = hit_j - window > 0 ? hit_j - window : 0, j1 = hit_j + window < slen ? hit_j + window : slen;
	int scores[128];
	oracle_scan_diags(matrix8, query, qlen, cbs, target, d, j0, j1, band, scores);
	return oracle_diag_alignment(scores, band, diag_score, gap_open, gap_extend);
}

/* per target: any hit with f1 > cutoff1 and f2 > cutoff2 (blastp; gapped_filter.cpp:42-62) */
int oracle_gapped_filter_target(const int8_t* matrix8, const int8_t* query, int qlen, const int8_t* cbs, const int8_t* target, int slen,
	const int32_t* hit_i, const int32_t* hit_j, int n_hits, int cutoff1, int cutoff2, int window2, int diag_score, int gap_open, int gap_extend)
{
	for (int h = 0; h < n_hits; ++h) {
		const int f1 = oracle_gapped_filter_hit(matrix8, query, qlen, cbs, target, slen, hit_i[h], hit_j[h], 64, 100, diag_score, gap_open, gap_extend);
		if (f1 > cutoff1) {
			const int f2 = oracle_gapped_filter_hit(matrix8, query, qlen, cbs, target, slen, hit_i[h], hit_j[h], 128, window2, diag_score, gap_open, gap_extend);
			if (f2 > cutoff2) return 1;
		}
	}
	return 0;
}

/* CutoffTable2D(evalue): table[b1*32+b2] = smallest raw score in [10,1000) whose normalised e-value (1e9 db letters)
 * for lengths 2^(b1-1) x 2^(b2-1) is <= evalue; e must have been initialised with db_letters = 1e9. */
void oracle_cutoff_table2d(const oracle_evaluer* e, double evalue, int32_t* table)
{
	for (int i = 0; i < 32 * 32; ++i) table[i] = 0;
	for (int b1 = 1; b1 <= 31; ++b1)
		for (int b2 = 1; b2 <= 31; ++b2) {
			int r = 1000;
			for (int i = 10; i < 1000; ++i)
				if (oracle_evalue(e, i, 1u << (b1 - 1), 1u << (b2 - 1)) <= evalue) { r = i; break; }
			table[b1 * 32 + b2] = r;
		}
}
