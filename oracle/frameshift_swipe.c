/* oracle/frameshift_swipe.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's three-frame banded sweep of frameshift alignment (blastx -F):
 *   banded_3frame_swipe<Sv, Traceback>   /root/reference/src/dp/swipe/banded_3frame_swipe.cpp:416-531
 *   cell_update (three frames)           src/dp/swipe/swipe.h:56-82
 *   the two matrices and their iterators banded_3frame_swipe.cpp:44-318
 *   traceback (transcript) / (score only) :345-414
 * One target = one channel of the reference's vector; the geometry the channel inherits from its vector batch (band = the
 * widest band of the batch, i0 / i1 = first and last query row of column 0, the target's position at column 0) is an input,
 * because the reference's score-only pass sweeps 16 targets (AVX2 int16 vectors) on ONE band geometry while its traceback pass
 * (int32_t "vector", one channel) gives every target its own. Arithmetic is plain int: the reference's biased saturating int16
 * vectors floor every sum at 0, which never changes a cell (a cell is floored at 0 anyway and gaps only lose score), and saturate
 * at 65535 -- reported here through *overflow so that the caller can repeat the target alone in 32 bits as the reference does
 * (banded_3frame_swipe.cpp:610-640).
 * Row r of a column = 3 * i + f (query position i in frame f of the strand); the band moves down one query position (3 rows) per
 * column. The arrays are the reference's, index for index, so that what the band edges read is what the reference reads. */
#include <limits.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define F3_MASK_LETTER 25      /* SUPER_HARD_MASK: what a channel reads before its target begins (target_iterator.h:107-113) */

static int mscore(const int8_t* m, int q, int s) { return m[(q & 31) * 32 + (s & 31)]; }
static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }

typedef struct { int sm4, sm3, sm2; } win3;

/* cell_update<Sv>(diag, shift0, shift1, scores, ge, go, fs, hgap, vgap, best), swipe.h:56-82 */
static int cell3(int diag, int shift0, int shift1, int m, int ge, int go, int fs, int* hgap, int* vgap, int* best)
{
	int cur = diag + m;
	const int f = m - fs;
	cur = imax(cur, shift0 + f);
	cur = imax(cur, shift1 + f);
	cur = imax(imax(cur, *vgap), *hgap);
	if (cur < 0) cur = 0;
	*best = imax(*best, cur);
	*vgap -= ge;
	*hgap -= ge;
	const int open = cur - go;
	*vgap = imax(*vgap, open);
	*hgap = imax(*hgap, open);
	return cur;
}

/* Score-only sweep of one channel. band / i0 / i1: the batch geometry (query positions), pos0: the target position of column 0
 * (negative: the channel starts later). Returns the best score, *max_col = its first column. */
int oracle_3frame_score(const int8_t* const frames[3], const int32_t lens[3], const int8_t* target, int tlen,
	int band, int i0, int i1, int pos0, const int8_t* matrix8, int gap_open, int gap_extend, int frame_shift, int* max_col, int* overflow)
{
	const int qlen = lens[0], qlen2 = lens[1], qlen3 = lens[2], B = band * 3;
	int* hgap = (int*)calloc((size_t)B + 3, sizeof(int));
	int* score = (int*)calloc((size_t)B + 2, sizeof(int));      /* (the reference's look-ahead reads one entry past its band + 1) */
	const int go = gap_open + gap_extend, ge = gap_extend;
	int best = 0, j = 0, pos = pos0;
	*max_col = 0;
	for (;; ++j, ++i0, ++i1, ++pos) {
		if (pos >= tlen) break;                                   /* TargetIterator::inc removed the channel */
		const int i0_ = imax(i0, 0), i1_ = imin(i1, qlen - 1);
		if (i0_ > i1_) break;
		const int off = (i0_ - i0) * 3;
		int* hp = hgap + off;
		int* sp = score + off;
		win3 w;                                                   /* Banded3FrameSwipeMatrix::ColumnIterator, :49-85 */
		w.sm4 = 0; w.sm3 = sp[0]; w.sm2 = sp[1];
		if (i0_ - i0 > 0) { sp[-1] = 0; sp[-2] = 0; sp[-3] = 0; }
		int vgap[3] = { 0, 0, 0 }, col_best = 0;
		const int s = pos >= 0 ? target[pos] : F3_MASK_LETTER;
		for (int i = i0_; i <= i1_; ++i) {
			for (int f = 0; f < 3; ++f) {
				if ((f == 1 && i >= qlen2) || (f == 2 && i >= qlen3)) goto column_done;
				int hg = hp[3];
				const int next = cell3(w.sm3, w.sm4, w.sm2, mscore(matrix8, frames[f][i], s), ge, go, frame_shift, &hg, &vgap[f], &col_best);
				*hp = hg;
				*sp = next;
				++hp; ++sp;
				w.sm4 = w.sm3; w.sm3 = w.sm2; w.sm2 = sp[1];
			}
		}
	column_done:
		if (col_best > best) { best = col_best; *max_col = j; }
	}
	free(hgap); free(score);
	*overflow = best >= 65535;
	return best;
}

/* the Hsp a score-only sweep reports (traceback(...) for Banded3FrameSwipeMatrix, :398-414) */
void oracle_3frame_score_range(int strand, int dna_len, int qlen, int band, int i0, int pos0, int max_col, oracle_hsp3* out)
{
	memset(out, 0, sizeof *out);
	out->q_end = imin(i0 + max_col + (band * 3) / 3 / 2, qlen);
	out->q_begin = imax(out->q_end - (pos0 + max_col), 0);
	out->frame = strand == 0 ? 0 : 3;
	if (strand == 0) { out->qs_begin = 3 * out->q_begin; out->qs_end = 3 * out->q_end; }
	else { out->qs_begin = dna_len - 3 * out->q_end; out->qs_end = dna_len - 3 * out->q_begin; }
}

/* TranslatedPosition(i, Frame(strand, f)).absolute(dna_len), basic/translated_position.h:121-128,160-163 */
static int absolute_pos(int i, int f, int strand, int dna_len)
{
	const int in_strand = f + 3 * i;
	return strand == 0 ? in_strand : dna_len - in_strand - 1;
}

/* Traceback sweep of ONE target on its own band [d_begin, d_end) (the reference runs it as a one-channel vector) and the walk back
 * over the stored scores. transcript: PackedOperation codes in alignment order, terminator not included. */
int oracle_3frame_traceback(const int8_t* const frames[3], const int32_t lens[3], int strand, int dna_len, const int8_t* target, int tlen,
	int d_begin, int d_end, const int8_t* matrix8, int gap_open, int gap_extend, int frame_shift,
	oracle_hsp3* out, uint8_t* transcript, int transcript_cap)
{
	const int qlen = lens[0], qlen2 = lens[1], qlen3 = lens[2];
	const int band = d_end - d_begin, B = band * 3;
	const int i2 = imax(d_end - 1, 0);
	int i1 = i2, i0 = i2 + 1 - band;
	const int pos0 = i1 - (d_end - 1);
	/* TargetIterator: cols = j1 - pos (target_iterator.h:71-76) */
	const int j1 = imin(qlen - 1 - d_begin, tlen - 1) + 1;
	const int cols = imax(j1 - pos0, 0);
	memset(out, 0, sizeof *out);
	if (band <= 0) return ORACLE_ERR_ARG;
	const size_t stride = (size_t)B + 1;
	int* hgap = (int*)calloc((size_t)B + 3, sizeof(int));
	int* sc = (int*)calloc(stride * ((size_t)cols + 2), sizeof(int));        /* Banded3FrameSwipeTracebackMatrix::score_, zero-filled */
	const int go = gap_open + gap_extend, ge = gap_extend;
	int best = 0, max_col = 0, j = 0, pos = pos0;
	const int i0_first = i0, i1_first = i1;
	for (;; ++j, ++i0, ++i1, ++pos) {
		if (pos >= tlen || j > cols) break;
		const int i0_ = imax(i0, 0), i1_ = imin(i1, qlen - 1);
		if (i0_ > i1_) break;
		const int off = (i0_ - i0) * 3;
		int* hp = hgap + off;
		const int* sp = sc + (size_t)j * stride + off;           /* previous column */
		int* sp1 = sc + ((size_t)j + 1) * stride + off;          /* this column */
		win3 w;                                                   /* ColumnIterator of the traceback matrix, :137-171 */
		w.sm4 = 0; w.sm3 = *(sp++); w.sm2 = *sp;
		if (i0_ - i0 > 0) { sp1[-1] = 0; sp1[-2] = 0; sp1[-3] = 0; }
		int vgap[3] = { 0, 0, 0 }, col_best = 0;
		const int s = pos >= 0 ? target[pos] : F3_MASK_LETTER;
		for (int i = i0_; i <= i1_; ++i) {
			for (int f = 0; f < 3; ++f) {
				if ((f == 1 && i >= qlen2) || (f == 2 && i >= qlen3)) goto column_done;
				int hg = hp[3];
				const int next = cell3(w.sm3, w.sm4, w.sm2, mscore(matrix8, frames[f][i], s), ge, go, frame_shift, &hg, &vgap[f], &col_best);
				*hp = hg;
				*sp1 = next;
				++hp; ++sp; ++sp1;
				w.sm4 = w.sm3; w.sm3 = w.sm2; w.sm2 = *sp;
			}
		}
	column_done:
		if (col_best > best) { best = col_best; max_col = j; }
	}
	free(hgap);
	out->score = best;
	if (best <= 0) { free(sc); return ORACLE_OK; }

	/* traceback<_sv>(...) :345-396 with i0 = i0_first, i1 = i1_first */
	const int j0 = i1_first - (d_end - 1);
	int rc = ORACLE_OK;
	{
		/* dp.traceback(max_col + 1, i0 + max_col, j0 + max_col, dna_len, channel, score), :282-292 */
		const int ci0 = i0_first + max_col;
		const int i_ = imax(-ci0, 0) * 3, i1b = imin(B, dna_len - 2 - ci0 * 3);
		const int* s = sc + ((size_t)max_col + 1) * stride + i_;
		const int* cur = NULL;
		int frame = 0, ti = 0, tj = j0 + max_col;
		for (int i = i_; i < i1b; ++i, ++s)
			if (*s == best) { cur = s; frame = i % 3; ti = ci0 + i / 3; break; }
		if (!cur) { free(sc); return ORACLE_ERR_TRACEBACK; }
		const int end_i = ti + 1, end_j = tj + 1, end_frame = frame;
		int n = 0;
#define PUSH(code) do { if (n >= transcript_cap) { rc = ORACLE_ERR_CAP; goto done; } transcript[n++] = (uint8_t)(code); } while (0)
		while (*cur > 0) {
			const int q = frames[frame][ti] & 31, t = target[tj] & 31;
			const int m = mscore(matrix8, q, t), score = *cur;
			const int sm3 = *(cur - (B + 1)), sm4 = *(cur - (B + 2)), sm2 = *(cur - B);
			int kind;
			if (score == sm3 + m) kind = 0;
			else if (score == sm4 + m - frame_shift) kind = 1;
			else if (score == sm2 + m - frame_shift) kind = 2;
			else kind = 3;
			if (kind < 3) {
				/* Hsp::push_match, basic/hssp.cpp:260-274 */
				if (q == t) { PUSH((0 << 6) | 1); ++out->identities; ++out->positives; }
				else { PUSH((3 << 6) | t); ++out->mismatches; if (m > 0) ++out->positives; }
				++out->length;
				if (kind == 0) { cur -= B + 1; --ti; --tj; }
				else if (kind == 1) {                             /* walk_forward_shift */
					PUSH((3 << 6) | 27);
					cur -= B + 2; --ti; --tj; --frame;
					if (frame == -1) { frame = 2; --ti; }
				}
				else {                                            /* walk_reverse_shift */
					PUSH((3 << 6) | 26);
					cur -= B; --ti; --tj; ++frame;
					if (frame == 3) { frame = 0; ++ti; }
				}
				continue;
			}
			/* walk_gap(d_begin, d_end), :219-263 */
			{
				const int gi0 = imax(d_begin + tj, 0), gj0 = imax(ti - d_end, -1);
				const int* h = cur - (B - 2);
				const int* h0 = cur - (ptrdiff_t)(tj - gj0) * (B - 2);
				const int* v = cur - 3;
				const int* v0 = cur - (ptrdiff_t)(ti - gi0 + 1) * 3;
				int g = gap_open + gap_extend, l = 1, found = 0;
				while (v > v0 && h > h0) {
					if (score + g == *h) { found = 2; break; }
					else if (score + g == *v) { found = 1; break; }
					h -= B - 2; v -= 3; ++l; g += gap_extend;
				}
				if (!found) while (v > v0) { if (score + g == *v) { found = 1; break; } v -= 3; ++l; g += gap_extend; }
				if (!found) while (h > h0) { if (score + g == *h) { found = 2; break; } h -= B - 2; ++l; g += gap_extend; }
				if (!found) { rc = ORACLE_ERR_TRACEBACK; goto done; }
				/* Hsp::push_gap(op, l, target.seq.data() + it.j + l) after the walk moved it.i / it.j */
				++out->gap_openings; out->length += l; out->gaps += l;
				if (found == 1) {                                 /* op_insertion, counts packed 63 at a time (PackedTranscript::push_back) */
					cur = v; ti -= l;
					int c = l;
					while (c > 0) { const int k = c > 63 ? 63 : c; PUSH((1 << 6) | k); c -= k; }
				}
				else {
					cur = h; tj -= l;
					for (int x = 0; x < l; ++x) PUSH((2 << 6) | (target[tj + l - x] & 31));
				}
			}
		}
		/* out.set_end(it.i + 1, it.j + 1, Frame(strand, it.frame)) was taken before the walk; set_begin after it (hssp.cpp:197-216) */
		out->q_end = end_i; out->s_end = end_j;
		out->q_begin = ti + 1; out->s_begin = tj + 1; out->frame = strand * 3 + frame;
		{
			/* set_end: TranslatedPosition(i, frame).absolute; set_begin likewise, strand decides which side */
			const int e = absolute_pos(end_i, end_frame, strand, dna_len), b = absolute_pos(ti + 1, frame, strand, dna_len);
			if (strand == 0) { out->qs_begin = b; out->qs_end = e; }
			else { out->qs_end = b + 1; out->qs_begin = e + 1; }
		}
		/* transcript.reverse() */
		for (int a = 0, z = n - 1; a < z; ++a, --z) { const uint8_t x = transcript[a]; transcript[a] = transcript[z]; transcript[z] = x; }
		out->transcript_len = n;
	}
done:
	free(sc);
	return rc;
}
