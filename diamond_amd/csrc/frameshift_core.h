// frameshift_core.h -- per-item arithmetic of the three-frame banded sweep (frameshift alignment, blastx -F).
//
// What it computes: the reference's banded_3frame_swipe (/root/reference/src/dp/swipe/banded_3frame_swipe.cpp:416-531, cell update
// src/dp/swipe/swipe.h:56-82) -- a local alignment of ONE strand of a DNA read, in all three reading frames at once, against a
// protein: row R = 3 i + f is query position i of frame f, a cell continues its own frame's diagonal (R - 3 in the previous
// column) for free, or the neighbouring frames' (R - 4, R - 2) for the frameshift penalty; gaps are affine inside a frame; the
// band [d_begin, d_end) is in query positions, so a column holds 3 * band rows and the window moves down three rows per column.
//
// How the device runs it: ONE LANE PER WORK ITEM, 64 items per wavefront (the items of a wavefront are neighbours in an order by
// band and column count). The reference's own parallelism is the same idea 16 wide (one target per int16 SIMD channel); a
// three-frame cell depends on three rows of the previous column AND on its own frame's cell three rows up in the SAME column, which
// leaves no room for the anti-diagonal lane = diagonal mapping of the protein sweeps without tripling the cross-lane traffic,
// while a read against its 10 - 50 targets on both strands gives thousands of independent items per query block. A lane keeps its
// window of the previous column (three scores) in registers and its two state columns -- scores and horizontal gaps, 3 * band
// entries each -- in HBM, interleaved over the 64 lanes of the wavefront (entry k of lane l at k * 64 + l), so that every load or
// store of the wavefront is one 256-byte line.
//
// Values outside the band or the matrix read as 0, as the reference's zero-initialised band arrays make them; scores are exact
// ints (the reference's biased int16 vectors floor at 0 exactly where a cell is floored anyway; its saturation at 65535 is reported
// so that the host can repeat the item alone, as the reference does).
// Shared by frameshift_kernels.hip and the CPU emulator of the test-suite (tests/emu/frameshift_emu.cpp).
#pragma once
#include <stdint.h>
#include "swipe_core.h"

namespace dmnd {

enum { F3_MASKED_COLUMN = 25, OP_FRAMESHIFT_REVERSE = (OP_SUBSTITUTION << OP_COUNT_BITS) | 26, OP_FRAMESHIFT_FORWARD = (OP_SUBSTITUTION << OP_COUNT_BITS) | 27 };

// One work item: a strand's three frames against one target on a band.
struct F3Item {
	const int8_t* frame[3];      // letters of the strand's frames 0, 1, 2
	int len[3];                  // their lengths (len[0] >= len[1] >= len[2], at most one apart)
	const int8_t* target;
	int tlen;
	// geometry of the sweep (query positions): rows i0 + j .. i1 + j in column j, band = i1 - i0 + 1; the target's letter of column j
	// is pos0 + j (a column before the target's first letter scores as the hard-mask letter). For an item swept alone:
	// i1 = max(d_end - 1, 0), i0 = i1 + 1 - (d_end - d_begin), pos0 = i1 - (d_end - 1).
	int i0, i1, pos0;
};

DMND_HD void f3_own_geometry(F3Item& it, int d_begin, int d_end)
{
	it.i1 = imax(d_end - 1, 0);
	it.i0 = it.i1 + 1 - (d_end - d_begin);
	it.pos0 = it.i1 - (d_end - 1);
}

// strided array of the lane's state (stride 64 on the device: the wavefront's lanes side by side; 1 in the emulator)
struct F3Column {
	int32_t* p;
	int stride;
	DMND_HD int32_t get(int k) const { return p[(int64_t)k * stride]; }
	DMND_HD void set(int k, int32_t v) const { p[(int64_t)k * stride] = v; }
};

struct F3Penalties { int open, extend, shift; };      // open = gap_open + gap_extend

// The cell: continue the frame's own diagonal, change frame (two ways), or end a gap; then what the gaps that start or go on here
// are worth to the cells below (vgap, same frame) and to the right (hgap).
DMND_HD int f3_cell(int own, int from_prev_frame, int from_next_frame, int m, const F3Penalties& p, int& hgap, int& vgap)
{
	const int shifted = m - p.shift;
	int c = imax(imax(own + m, from_prev_frame + shifted), imax(from_next_frame + shifted, imax(vgap, hgap)));
	c = imax(c, 0);
	const int opened = c - p.open;
	vgap = imax(vgap - p.extend, opened);
	hgap = imax(hgap - p.extend, opened);
	return c;
}

// ---- score-only sweep ------------------------------------------------------------------------------------------------------
// S: the score column, entry k = row 3 (i0 + j) + k of the column at hand, updated in place (B + 2 entries, zero on entry);
// G: horizontal gaps, entry k + 3 read for band row k, entry k written (B + 3 entries, zero on entry).
// Returns the best score; max_col = the first column that reaches it (the reference keeps `col_best > best`).
DMND_HD int f3_sweep_score(const F3Item& it, const F3Column& S, const F3Column& G, const int8_t* M, const F3Penalties& pen, int& max_col)
{
	const int qlen = it.len[0];
	int best = 0;
	max_col = 0;
	for (int j = 0;; ++j) {
		const int pos = it.pos0 + j;
		if (pos >= it.tlen) break;
		const int top = imax(it.i0 + j, 0), bottom = imin(it.i1 + j, qlen - 1);
		if (top > bottom) break;
		int k = (top - (it.i0 + j)) * 3;                      // band row of the first cell
		// the window of the previous column around band row k: rows k - 1, k, k + 1 (a column that starts below the band's top
		// has nothing above it)
		int above = 0, here = S.get(k), below = S.get(k + 1);
		if (k > 0) { S.set(k - 1, 0); S.set(k - 2, 0); S.set(k - 3, 0); }
		const int8_t* column = M + (pos >= 0 ? (it.target[pos] & LETTER_MASK) : (int)F3_MASKED_COLUMN);      // score_matrix(query letter, target letter)
		int vgap[3] = { 0, 0, 0 }, col_best = 0;
		for (int i = top; i <= bottom; ++i) {
#pragma unroll
			for (int f = 0; f < 3; ++f) {
				if (f > 0 && i >= it.len[f]) { i = bottom; break; }      // the shorter frames end one position early: so does the column
				int hg = G.get(k + 3);
				const int c = f3_cell(here, above, below, column[(it.frame[f][i] & LETTER_MASK) * 32], pen, hg, vgap[f]);
				G.set(k, hg);
				S.set(k, c);
				col_best = imax(col_best, c);
				++k;
				above = here; here = below; below = S.get(k + 1);
			}
		}
		if (col_best > best) { best = col_best; max_col = j; }
	}
	return best;
}

// ---- traceback sweep -------------------------------------------------------------------------------------------------------
// All columns are kept: column c (1-based; column 0 is the empty one before the first) at T + c * (B + 1), entry k = band row k,
// entry B = the row below the band (0). The caller zero-fills T ((cols + 2) * (B + 1) entries).
DMND_HD int f3_trace_cols(const F3Item& it)
{
	const int d_begin = it.i0 - it.pos0;                      // own geometry: i0 = d_begin + pos0
	const int j1 = imin(it.len[0] - 1 - d_begin, it.tlen - 1) + 1;
	return imax(j1 - it.pos0, 0);
}

DMND_HD int f3_sweep_trace(const F3Item& it, int32_t* T, const F3Column& G, const int8_t* M, const F3Penalties& pen, int& max_col)
{
	const int qlen = it.len[0], B = (it.i1 - it.i0 + 1) * 3, cols = f3_trace_cols(it);
	int best = 0;
	max_col = 0;
	for (int j = 0; j <= cols; ++j) {
		const int pos = it.pos0 + j;
		if (pos >= it.tlen) break;
		const int top = imax(it.i0 + j, 0), bottom = imin(it.i1 + j, qlen - 1);
		if (top > bottom) break;
		int k = (top - (it.i0 + j)) * 3;
		const int32_t* prev = T + (int64_t)j * (B + 1);
		int32_t* cur = T + (int64_t)(j + 1) * (B + 1);
		int above = 0, here = prev[k], below = prev[k + 1];
		const int8_t* column = M + (pos >= 0 ? (it.target[pos] & LETTER_MASK) : (int)F3_MASKED_COLUMN);      // score_matrix(query letter, target letter)
		int vgap[3] = { 0, 0, 0 }, col_best = 0;
		for (int i = top; i <= bottom; ++i) {
#pragma unroll
			for (int f = 0; f < 3; ++f) {
				if (f > 0 && i >= it.len[f]) { i = bottom; break; }
				int hg = G.get(k + 3);
				const int c = f3_cell(here, above, below, column[(it.frame[f][i] & LETTER_MASK) * 32], pen, hg, vgap[f]);
				G.set(k, hg);
				cur[k] = c;
				col_best = imax(col_best, c);
				++k;
				above = here; here = below; below = prev[k + 1];
			}
		}
		if (col_best > best) { best = col_best; max_col = j; }
	}
	return best;
}

// What the walk back over the kept columns reports (Hsp fields of the reference's traceback, banded_3frame_swipe.cpp:345-396).
struct F3Walk {
	int status;                  // 0, or DMND_E_TRACEBACK (-6) / DMND_E_CAP (-5)
	int frame;                   // frame (0 - 2) of the first aligned query position
	int q_begin, q_end, s_begin, s_end, end_frame;
	int length, identities, mismatches, positives, gap_openings, gaps;
	int transcript_len;
};

// Walk from the first cell of column max_col that holds the best score (rows top to bottom) back to a cell with score 0. At every
// cell: the frame's own diagonal if it explains the score, else the previous frame's, else the next frame's (both with the
// frameshift penalty), else the shortest gap that explains it -- horizontal before vertical at equal length. transcript receives
// the PackedOperation codes in alignment order (no terminator); cap = its size.
DMND_HD F3Walk f3_walk(const F3Item& it, const int32_t* T, const int8_t* M, int gap_open, int gap_extend, int shift, int best, int max_col,
	uint8_t* transcript, int cap, int dna_len)
{
	F3Walk r;
	r.status = 0; r.frame = 0; r.q_begin = r.q_end = r.s_begin = r.s_end = r.end_frame = 0;
	r.length = r.identities = r.mismatches = r.positives = r.gap_openings = r.gaps = 0; r.transcript_len = 0;
	const int B = (it.i1 - it.i0 + 1) * 3, W = B + 1;
	const int d_begin = it.i0 - it.pos0, d_end = d_begin + (it.i1 - it.i0 + 1);
	const int first_row = it.i0 + max_col;                    // query position of band row 0 in column max_col
	const int k0 = imax(-first_row, 0) * 3, k1 = imin(B, dna_len - 2 - first_row * 3);
	const int32_t* cell = nullptr;
	int frame = 0, i = 0, j = it.pos0 + max_col;
	for (int k = k0; k < k1; ++k)
		if (T[(int64_t)(max_col + 1) * W + k] == best) { cell = T + (int64_t)(max_col + 1) * W + k; frame = k % 3; i = first_row + k / 3; break; }
	if (!cell) { r.status = -6; return r; }
	r.q_end = i + 1; r.s_end = j + 1; r.end_frame = frame;
	int n = 0;
	auto push = [&](int code) { if (n < cap) transcript[cap - 1 - n] = (uint8_t)code; ++n; };      // built back to front
	while (*cell > 0) {
		const int q = it.frame[frame][i] & LETTER_MASK, t = it.target[j] & LETTER_MASK;
		const int m = M[q * 32 + t], score = *cell;
		int step = 0;                                         // 1: own diagonal, 2: from the previous frame, 3: from the next frame
		if (score == cell[-W] + m) step = 1;
		else if (score == cell[-W - 1] + m - shift) step = 2;
		else if (score == cell[-W + 1] + m - shift) step = 3;
		if (step) {
			if (q == t) { push((OP_MATCH << OP_COUNT_BITS) | 1); ++r.identities; ++r.positives; }
			else { push((OP_SUBSTITUTION << OP_COUNT_BITS) | t); ++r.mismatches; if (m > 0) ++r.positives; }
			++r.length;
			--i; --j;
			if (step == 1) cell -= W;
			else if (step == 2) { push(OP_FRAMESHIFT_FORWARD); cell -= W + 1; if (--frame < 0) { frame = 2; --i; } }
			else { push(OP_FRAMESHIFT_REVERSE); cell -= W - 1; if (++frame > 2) { frame = 0; ++i; } }
			continue;
		}
		// a gap: same row in earlier columns (the window moves three rows per column: W - 3 entries back per column), or the
		// frame's rows above in this column (3 entries up per position)
		const int max_h = j - imax(i - d_end, -1), max_v = i - imax(d_begin + j, 0) + 1;      // exclusive bounds of both searches
		int l = 1, g = gap_open + gap_extend, kind = 0;
		for (; l < max_h || l < max_v; ++l, g += gap_extend) {
			if (l < max_h && score + g == cell[-(int64_t)l * (W - 3)]) { kind = 2; break; }
			if (l < max_v && score + g == cell[-3 * l]) { kind = 1; break; }
		}
		if (!kind) { r.status = -6; return r; }
		++r.gap_openings; r.length += l; r.gaps += l;
		if (kind == 1) {
			cell -= 3 * l; i -= l;
			for (int c = l; c > 0;) { const int x = imin(c, (int)OP_MAX_COUNT); push((OP_INSERTION << OP_COUNT_BITS) | x); c -= x; }
		}
		else {
			cell -= (int64_t)l * (W - 3);
			for (int x = 0; x < l; ++x) push((OP_DELETION << OP_COUNT_BITS) | (it.target[j - x] & LETTER_MASK));
			j -= l;
		}
	}
	r.q_begin = i + 1; r.s_begin = j + 1; r.frame = frame;
	r.transcript_len = n;
	if (n > cap) r.status = -5;
	return r;
}

// position of query position i of frame f in the read (TranslatedPosition::absolute, basic/translated_position.h:121-128)
DMND_HD int f3_read_position(int i, int f, int strand, int dna_len)
{
	const int in_strand = f + 3 * i;
	return strand == 0 ? in_strand : dna_len - in_strand - 1;
}

// Hsp::query_source_range of a walked alignment (Hsp::set_begin / set_end, basic/hssp.cpp:197-216)
DMND_HD void f3_read_range(const F3Walk& w, int strand, int dna_len, int& begin, int& end)
{
	const int b = f3_read_position(w.q_begin, w.frame, strand, dna_len), e = f3_read_position(w.q_end, w.end_frame, strand, dna_len);
	if (strand == 0) { begin = b; end = e; }
	else { end = b + 1; begin = e + 1; }
}

// the ranges a score-only sweep reports (banded_3frame_swipe.cpp:398-414): no walk, an estimate from the end column
DMND_HD void f3_score_range(int strand, int dna_len, int qlen, int band, int i0, int pos0, int max_col, int& q_begin, int& q_end, int& rs_begin, int& rs_end)
{
	q_end = imin(i0 + max_col + band / 2, qlen);
	q_begin = imax(q_end - (pos0 + max_col), 0);
	if (strand == 0) { rs_begin = 3 * q_begin; rs_end = 3 * q_end; }
	else { rs_begin = dna_len - 3 * q_end; rs_end = dna_len - 3 * q_begin; }
}

}  // namespace dmnd
