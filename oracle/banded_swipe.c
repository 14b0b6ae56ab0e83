/* oracle/banded_swipe.c -- TEST INFRASTRUCTURE ONLY (parity checker; never linked, imported or
 * called by the product path -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use it).
 *
 * Plain-C restatement of the reference's banded Smith-Waterman (SWIPE) for ONE target, i.e. one
 * SIMD channel of
 *     DP::BandedSwipe::DISPATCH_ARCH::swipe<Sv,Cbs,Cfg>()      src/dp/swipe/banded_swipe.h:189-351
 * with the cell recurrence of
 *     swipe_cell_update()                                      src/dp/swipe/cell_update.h:103-140
 * the band-row/column bookkeeping of Matrix / TracebackVectorMatrix
 *                                                              src/dp/swipe/banded_matrix.h:30-134,314-440
 * the per-channel start column of TargetIterator               src/dp/swipe/target_iterator.h:60-90
 * the traceback walk                                           src/dp/swipe/banded_swipe.h:128-183
 * the transcript/statistics accounting of Hsp::push_match / push_gap   src/basic/hssp.cpp:260-290
 * and the forward / backward statistics cells                  src/dp/swipe/stat_cell.h:47-279.
 *
 * Semantics pinned here (and checked against the genuine reference through oracle/_ref/diamond_tap):
 *  - scores are the reference's saturating vectors seen as true integers clamped below at 0
 *    (ScoreVector<int8_t,SCHAR_MIN> / <int16_t,SHRT_MIN>: dp/score_vector_int8.h:212-330); the
 *    8->16->32 bit escalation (swipe_wrapper.cpp:446-470) only re-runs a target whose best hit the
 *    type maximum, so the reported numbers equal unbounded arithmetic -- which is what is computed.
 *  - STRICT_BAND (CMakeLists.txt:35): a target only ever sees cells of its own diagonal band
 *    [d_begin, d_end); the other channels of a SIMD batch never influence it, so a per-target
 *    restatement is exact (the -T_MIN lane masks of RangePartition, range_partition.h:29, make
 *    out-of-band cells zero).
 *  - letters are masked with LETTER_MASK=31 before the matrix lookup (basic/value.h:61,71;
 *    basic/sequence.h:80-87); the matrix is the reference's 32x32 int8 table (score_matrix.h:69).
 *
 * Pinned: tests/test_oracle_swipe.py compares every field against tests/golden/swipe_*.tap,
 * minted from the reference itself by tests/golden/make_swipe_golden.sh.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define LETTER_MASK 31
#define OP_MATCH 0
#define OP_INSERTION 1
#define OP_DELETION 2
#define OP_SUBSTITUTION 3
#define COUNT_BITS 6
#define MAX_COUNT 63

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* DpTarget::banded_cols, src/dp/dp.h:47-52 */
int oracle_banded_cols(int qlen, int tlen, int d_begin, int d_end)
{
	const int pos = imax(d_end - 1, 0) - (d_end - 1);
	const int j1 = imin(qlen - 1 - d_begin, tlen - 1) + 1;
	return j1 - pos;
}

/* Optional model of ONE CHANNEL of the reference's 8-bit vector pass whose own band is narrower than the vector's band
 * (banded_swipe.h:211-224,277-297, RangePartition: the rows of the vector band are cut into parts, and on a part that lies outside
 * a channel's own band the lane mask is ADDED, saturating, to vgap (once per part), hgap and the match scores (every cell)). For
 * the biased int8 vectors the mask is -128, not minus infinity, so a gap value of 128 or more survives as value - 128 in a cell
 * outside the band. oracle_set_channel_band(own_d_begin, own_d_end, 128) makes oracle_banded_swipe treat its d_begin / d_end as
 * the VECTOR band and the given band as the channel's own; penalty 0 switches the model off. Only tests/test_oracle_swipe.py uses
 * it, to show that the leak never reaches a result. */
static int g_own_b = 0, g_own_e = 0, g_pen = 0;
void oracle_set_channel_band(int own_d_begin, int own_d_end, int penalty) { g_own_b = own_d_begin; g_own_e = own_d_end; g_pen = penalty; }

/* One target through the banded sweep.
 * mode: ORACLE_SCORE_ONLY  -> score (DummyRowCounter: max_band_row stays 0)        swipe_wrapper.cpp:187-190
 *       ORACLE_COORDS      -> + end coordinates (VectorRowCounter)                 swipe_wrapper.cpp:196-200
 *       ORACLE_TRACEBACK   -> + start coordinates, transcript, alignment statistics (bins 0-2) :191-194
 *       ORACLE_STATS_FWD   -> end coordinates + identities/length (ForwardCell)    :201-204
 *       ORACLE_STATS_BWD   -> end coordinates + mismatches/gap openings (BackwardCell) :207-215
 */
int oracle_banded_swipe(const int8_t* query, int qlen, const int8_t* cbs,
	const int8_t* target, int tlen, int d_begin, int d_end,
	const int8_t* matrix8, int gap_open, int gap_extend, int mode,
	oracle_hsp* out, uint8_t* transcript, int transcript_cap)
{
	memset(out, 0, sizeof(*out));
	const int band = d_end - d_begin;
	if (band <= 0 || qlen <= 0 || tlen <= 0)
		return ORACLE_ERR_ARG;
	const int go = gap_open + gap_extend, ge = gap_extend;          /* banded_swipe.h:232-233 */
	int i1 = imax(d_end - 1, 0);                                    /* :212 */
	int i0 = i1 + 1 - band;                                         /* :214 */
	const int pos0 = i1 - (d_end - 1);                              /* target_iterator.h:71 */
	const int i0_init = i0;
	const int cols = oracle_banded_cols(qlen, tlen, d_begin, d_end);
	const int want_trace = mode == ORACLE_TRACEBACK;
	const int want_rows = mode != ORACLE_SCORE_ONLY;
	const int fwd = mode == ORACLE_STATS_FWD, bwd = mode == ORACLE_STATS_BWD;

	int32_t* hgap = (int32_t*)calloc((size_t)band + 1, sizeof(int32_t));
	int32_t* score = (int32_t*)calloc((size_t)band, sizeof(int32_t));
	/* statistics carried with H (score_) and E (hgap_): a = ident | mismatch, b = len | gapopen */
	int32_t *sa = NULL, *sb = NULL, *ha = NULL, *hb = NULL;
	if (fwd || bwd) {
		sa = (int32_t*)calloc((size_t)band, sizeof(int32_t));
		sb = (int32_t*)calloc((size_t)band, sizeof(int32_t));
		ha = (int32_t*)calloc((size_t)band + 1, sizeof(int32_t));
		hb = (int32_t*)calloc((size_t)band + 1, sizeof(int32_t));
	}
	uint8_t* trace = NULL;                                          /* bit0 gap-v, bit1 gap-h, bit2 open-v, bit3 open-h */
	if (want_trace)
		trace = (uint8_t*)calloc((size_t)(cols + 1) * (size_t)band, 1);  /* banded_matrix.h:419 */

	int best = 0, max_col = 0, max_band_row = 0, stat_a = 0, stat_b = 0;
	int j = 0;
	for (int pos = pos0; pos < tlen; ++pos, ++j, ++i0, ++i1) {      /* :246, :319-321 */
		const int i0_ = imax(i0, 0), i1_ = imin(i1, qlen - 1) + 1;   /* :247 */
		if (i0_ >= i1_)
			break;
		const int t = target[pos] & LETTER_MASK;
		const int t_raw = target[pos];                                /* VectorIdMask compares the raw target byte, stat_cell.h:41-44 */
		const int8_t* mrow = matrix8 + 32 * t;
		int vgap = 0, va = 0, vb = 0, col_best = 0, i_max = 0;
		int prev_own = g_pen ? 0 : 1;                               /* the part above the own band, if any, is entered first */
		if (g_pen && g_own_b > d_begin) vgap = imax(vgap - g_pen, 0);
		for (int i = i0_; i < i1_; ++i) {
			const int r = i - i0;
			const int q = query[i] & LETTER_MASK;
			int hg = hgap[r + 1];                                   /* banded_matrix.h:47 */
			int m = mrow[q];
			if (cbs)
				m += cbs[i];
			if (g_pen) {                                             /* channel model: lane mask on the parts outside the own band */
				const int d = i - pos, own = d >= g_own_b && d < g_own_e;
				if (!own && prev_own && d >= g_own_e) vgap = imax(vgap - g_pen, 0);      /* vgap += target_mask at the start of the part below */
				if (!own) { hg = imax(hg - g_pen, 0); m -= g_pen; }
				prev_own = own;
			}
			int cur = score[r] + m;                                 /* cell_update.h:116-117 */
			int ca = 0, cb = 0, hga = 0, hgb = 0;
			if (fwd) {                                              /* stat_cell.h:225-231 */
				ca = sa[r] + (q == t_raw ? 1 : 0);
				cb = sb[r] + 1;
				hga = ha[r + 1];
				hgb = hb[r + 1] + 1;
				vb += 1;
			}
			else if (bwd) {                                         /* stat_cell.h:234-236 */
				ca = sa[r] + (q == t_raw ? 0 : 1);
				cb = sb[r];
				hga = ha[r + 1];
				hgb = hb[r + 1];
			}
			/* set_max: ties take the argument's statistics, stat_cell.h:257-271 */
			if (hg >= cur) { cur = hg; ca = hga; cb = hgb; }
			if (vgap >= cur) { cur = vgap; ca = va; cb = vb; }
			if (cur < 0) cur = 0;                                    /* saturate() */
			if (want_trace) {
				uint8_t mk = 0;
				if (cur == vgap) mk |= 1;                           /* make_gap_mask, cell_update.h:80-82 */
				if (cur == hg) mk |= 2;
				trace[(size_t)(j + 1) * band + r] = mk;
			}
			if (cur > col_best) col_best = cur;                     /* :126 */
			if (want_rows && col_best == cur) i_max = r;            /* VectorRowCounter::inc, cell_update.h:44-47 */
			vgap = imax(vgap - ge, 0);                              /* :130-131, saturating */
			hg = imax(hg - ge, 0);
			int open = imax(cur - go, 0);                           /* :132-133 */
			int oa = ca, ob = cb;
			if (fwd || bwd) {
				if (bwd) ob += 1;                                   /* update_open, stat_cell.h:246-252 */
				if (cur == 0) { ca = 0; cb = 0; }                    /* zero_mask: resets current, not open */
			}
			if (open >= hg) { hg = open; hga = oa; hgb = ob; }       /* :135-136 */
			if (open >= vgap) { vgap = open; va = oa; vb = ob; }
			if (want_trace) {
				uint8_t mk = 0;
				if (vgap == open) mk |= 4;                          /* make_open_mask, cell_update.h:89-91 */
				if (hg == open) mk |= 8;
				trace[(size_t)(j + 1) * band + r] |= mk;
			}
			hgap[r] = hg;                                           /* :297-298 */
			score[r] = cur;
			if (fwd || bwd) { ha[r] = hga; hb[r] = hgb; sa[r] = ca; sb[r] = cb; }
		}
		if (col_best > best) {                                      /* :312-318 */
			best = col_best;
			max_col = j;
			max_band_row = i_max;
			if (fwd || bwd) { stat_a = sa[i_max]; stat_b = sb[i_max]; }
		}
	}

	out->score = best;
	out->max_col = max_col;
	out->max_band_row = max_band_row;
	out->cols = cols;
	/* traceback() for Matrix<Cell>, banded_swipe.h:101-105 */
	out->q_end = i0_init + max_col + max_band_row + 1;
	out->s_end = pos0 + max_col + 1;
	if (fwd) { out->identities = stat_a; out->length = stat_b; }
	if (bwd) { out->mismatches = stat_a; out->gap_openings = stat_b; }

	int rc = ORACLE_OK;
	if (want_trace && best > 0) {
		/* banded_swipe.h:128-183 with TracebackVectorMatrix::TracebackIterator, banded_matrix.h:359-408 */
		const uint8_t* mask = trace + (size_t)(max_col + 1) * band + max_band_row;
		int i = i0_init + max_col + max_band_row, jj = pos0 + max_col;
		int sc = 0, n = 0;
		out->q_end = i + 1;
		out->s_end = jj + 1;
		while (i >= 0 && jj >= 0 && sc < best) {
			if ((*mask & 3) == 0) {
				const int q = query[i] & LETTER_MASK, s = target[jj] & LETTER_MASK;
				int m = matrix8[32 * s + q];
				const int positive = m > 0;
				if (cbs) m += cbs[i];
				sc += m;
				if (n < transcript_cap)
					transcript[n] = (uint8_t)(q == s ? (OP_MATCH << COUNT_BITS) | 1 : (OP_SUBSTITUTION << COUNT_BITS) | s);
				++n;
				if (q == s) { ++out->identities; ++out->positives; }
				else { ++out->mismatches; if (positive) ++out->positives; }
				++out->length;
				mask -= band; --i; --jj;                            /* walk_diagonal */
			}
			else {
				int l = 0;
				if (*mask & 1) {                                    /* vertical gap first, banded_matrix.h:382 */
					do { ++l; --i; --mask; } while ((*mask & 4) == 0 && i > 0);
					int c = l;
					while (c > 0) {                                 /* PackedTranscript::push_back(op,count) */
						const int k = imin(c, MAX_COUNT);
						if (n < transcript_cap) transcript[n] = (uint8_t)((OP_INSERTION << COUNT_BITS) | k);
						++n; c -= k;
					}
				}
				else {
					const int j_before = jj;
					do { ++l; --jj; mask -= band - 1; } while ((*mask & 8) == 0 && jj > 0);
					for (int k = 0; k < l; ++k) {                   /* Hsp::push_gap, hssp.cpp:284-289 */
						if (n < transcript_cap)
							transcript[n] = (uint8_t)((OP_DELETION << COUNT_BITS) | (target[j_before - k] & LETTER_MASK));
						++n;
					}
				}
				++out->gap_openings;
				out->length += l;
				out->gaps += l;
				sc -= gap_open + l * gap_extend;
			}
		}
		if (sc != best)
			rc = ORACLE_ERR_TRACEBACK;                              /* "Traceback error." :168 */
		out->q_begin = i + 1;
		out->s_begin = jj + 1;
		if (n > transcript_cap)
			rc = ORACLE_ERR_CAP;
		else {
			for (int a = 0, b = n - 1; a < b; ++a, --b) {           /* transcript.reverse() */
				const uint8_t x = transcript[a]; transcript[a] = transcript[b]; transcript[b] = x;
			}
		}
		out->transcript_len = n;                                    /* terminator not counted */
	}

	free(hgap); free(score); free(sa); free(sb); free(ha); free(hb); free(trace);
	return rc;
}

/* The "stats without traceback" path: forward pass with ForwardCell, then
 * recompute_reversed() -- src/dp/swipe/swipe_wrapper.cpp:364-444 -- a second banded sweep over the
 * reversed query and the reversed target prefix [0, s_end) with BackwardCell, whose end point gives
 * the start coordinates (carry-over branch of traceback(), banded_swipe.h:107-117).
 * hsp_values is the reference's HspValues bit set (basic/match.h); it selects the cell types as
 * dispatch_swipe() does (swipe_wrapper.cpp:196-215). */
int oracle_swipe_stats(const int8_t* query, int qlen, const int8_t* cbs,
	const int8_t* target, int tlen, int d_begin, int d_end,
	const int8_t* matrix8, int gap_open, int gap_extend, unsigned hsp_values, oracle_hsp* out)
{
	enum { IDENT = 1 << 5, LENGTH = 1 << 6, MISMATCHES = 1 << 7, GAP_OPENINGS = 1 << 8, QUERY_START = 1 << 1, TARGET_START = 1 << 3 };
	oracle_hsp f, b;
	const int mode_f = (hsp_values & (IDENT | LENGTH)) ? ORACLE_STATS_FWD : ORACLE_COORDS;
	int rc = oracle_banded_swipe(query, qlen, cbs, target, tlen, d_begin, d_end, matrix8, gap_open, gap_extend, mode_f, &f, NULL, 0);
	if (rc != ORACLE_OK)
		return rc;
	*out = f;
	if (f.score <= 0 || !(hsp_values & (QUERY_START | TARGET_START | MISMATCHES | GAP_OPENINGS)))   /* reversed(), :115-118 */
		return ORACLE_OK;
	const int tl = f.s_end;                                            /* :378,382 */
	int8_t* rq = (int8_t*)malloc((size_t)qlen);
	int8_t* rt = (int8_t*)malloc((size_t)tl);
	int8_t* rc_ = cbs ? (int8_t*)malloc((size_t)qlen) : NULL;
	for (int i = 0; i < qlen; ++i) {
		rq[i] = query[qlen - 1 - i];
		if (cbs) rc_[i] = cbs[qlen - 1 - i];
	}
	for (int i = 0; i < tl; ++i)
		rt[i] = target[tl - 1 - i];
	const int rd0 = -(d_end - 1) + qlen - tl;                          /* Geo::rev_diag, util/geo/geo.h:37 */
	const int rd1 = -d_begin + qlen - tl + 1;
	const int mode_b = (hsp_values & (MISMATCHES | GAP_OPENINGS)) ? ORACLE_STATS_BWD : ORACLE_COORDS;
	rc = oracle_banded_swipe(rq, qlen, rc_, rt, tl, rd0, rd1, matrix8, gap_open, gap_extend, mode_b, &b, NULL, 0);
	free(rq); free(rt); free(rc_);
	if (rc != ORACLE_OK)
		return rc;
	out->score = b.score;
	out->q_begin = qlen - b.q_end;                                     /* banded_swipe.h:114-115 */
	out->s_begin = tl - b.s_end;
	out->mismatches = b.mismatches;
	out->gap_openings = b.gap_openings;
	out->gaps = out->length - out->identities - out->mismatches;       /* assign_stats, stat_cell.h:216-220 */
	return ORACLE_OK;
}
