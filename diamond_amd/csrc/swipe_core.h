// swipe_core.h -- per-lane arithmetic of the MI355X banded Smith-Waterman kernels.
//
// What it computes: the reference's banded SWIPE (DP::BandedSwipe::swipe,
// /root/reference/src/dp/swipe/banded_swipe.h:189-351; cell update cell_update.h:103-140) for one
// target per 64-lane wavefront -- NOT as the reference's column sweep with one target per SIMD
// channel, but as an anti-diagonal wavefront sweep: step a handles all cells with i + j = a.
// Inside the diagonal band [d_begin, d_end) only every other diagonal has a cell on a given
// anti-diagonal, so lane l owns 2*P consecutive diagonals (k = 2*P*l .. 2*P*l + 2*P - 1,
// k = i - j - d_begin) and computes P cells per step, alternating between its even and odd
// diagonals. A cell on diagonal k needs
//     H(i-1,j-1)   same diagonal, two steps ago         -> register
//     E(i,j-1)     diagonal k+1, previous step          -> register, or lane l+1 (one DPP/shuffle)
//     F(i-1,j)     diagonal k-1, previous step          -> register, or lane l-1 (one DPP/shuffle)
// so the whole DP state lives in VGPRs; no LDS or HBM traffic for H/E/F.
//
// Semantics (bit-exact with the reference, see oracle/banded_swipe.c): values are clamped below at 0
// like the reference's saturating score vectors; cells outside the band or the matrix read as 0;
// the end cell is, among all cells with the best score, the one with the smallest column j and
// then the largest row i (column-major scan with `col_best > best` and VectorRowCounter keeping the
// last row equal to the column maximum: banded_swipe.h:312-318, cell_update.h:44-47).
//
// The functions here are plain inline C++ shared by the HIP kernels (swipe_kernels.hip) and by the
// CPU lane-emulator used in the CPU test-suite (tests/emu) so the index arithmetic, tie-breaking and
// traceback walk are tested without a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DMND_HD __host__ __device__ __forceinline__
#else
#define DMND_HD inline
#endif

namespace dmnd {

enum { LETTER_MASK = 31 };
enum { TB_GAP_V = 1, TB_GAP_H = 2, TB_OPEN_V = 4, TB_OPEN_H = 8 };
enum { OP_MATCH = 0, OP_INSERTION = 1, OP_DELETION = 2, OP_SUBSTITUTION = 3, OP_COUNT_BITS = 6, OP_MAX_COUNT = 63 };

DMND_HD int imax(int a, int b) { return a > b ? a : b; }
DMND_HD int imin(int a, int b) { return a < b ? a : b; }

// Geometry of one work item shared by all lanes of the wave.
struct Geom {
	int qlen, tlen, d_begin, band;   // band = d_end - d_begin
	int a_first, a_last;             // first / last anti-diagonal that holds a cell (a_first has the parity of d_begin)
};

DMND_HD Geom make_geom(int qlen, int tlen, int d_begin, int d_end)
{
	Geom g;
	g.qlen = qlen; g.tlen = tlen; g.d_begin = d_begin; g.band = d_end - d_begin;
	// first anti-diagonal with a cell inside band x matrix
	int a0;
	if (d_begin > 0) a0 = d_begin;              // cell (d_begin, 0)
	else if (d_end <= 0) a0 = -(d_end - 1);     // cell (0, -(d_end-1))
	else a0 = 0;                                // cell (0, 0)
	// last: a = 2*min(qlen-1, tlen-1+d) - d is maximal at d* = clamp(qlen - tlen, d_begin, d_end-1)
	const int ds = imin(imax(qlen - tlen, d_begin), d_end - 1);
	g.a_last = 2 * imin(qlen - 1, tlen - 1 + ds) - ds;
	if ((a0 + d_begin) & 1) --a0;                // loop is unrolled by parity: start on an even (a + d_begin)
	g.a_first = a0;
	return g;
}

// Number of anti-diagonal steps and trace bytes of one item (trace: one byte per (step, diagonal pair)).
DMND_HD int64_t n_steps(const Geom& g) { return g.a_last >= g.a_first ? (int64_t)(g.a_last - g.a_first + 1) : 0; }
// The kernel runs steps in (even, odd) pairs, so the trace holds an even number of rows.
DMND_HD int64_t trace_rows(const Geom& g) { return (n_steps(g) + 1) / 2 * 2; }
// lanes own 2*P diagonals each: smallest power of two with 128*P >= band
DMND_HD int band_class(int band) { int P = 1; while (128 * P < band) P *= 2; return P; }
// ROW classes (packed 16-bit sweeps only, swipe16_kernels.hip): P = 3 / 5 -- the item is swept by ONE 16-lane DPP row of a wavefront
// instead of all 64 lanes, 2 P = 6 / 10 diagonals per lane: 96 / 160 diagonals. Four rows x two register halves = eight items per
// wavefront. They sit between the power-of-two classes: most bands of an alignment run are 61-81 or 131-161 diagonals wide, which
// leaves 37-52 % of the lanes of the classes 1 / 2 (128 / 256 diagonals) on diagonals outside the band.
DMND_HD bool row_class(int P) { return P == 3 || P == 5; }
DMND_HD int class_lanes(int P) { return row_class(P) ? 16 : 64; }
DMND_HD int band_class_rows(int band) { return band <= 96 ? 3 : band <= 128 ? 1 : band <= 160 ? 5 : band_class(band); }
// launch classes as small numbers (counters per class, sort buckets): P = 1 << c for c = 0 .. 9 (512 = 16 wavefronts of 32), then the rows
enum { CLASS_INDEX_ROW3 = 10, CLASS_INDEX_ROW5 = 11, CLASS_INDICES = 12 };
DMND_HD int class_index(int P) { if (P == 3) return CLASS_INDEX_ROW3; if (P == 5) return CLASS_INDEX_ROW5; int c = 0; while ((1 << c) < P) ++c; return c; }
DMND_HD int class_of_index(int c) { return c == CLASS_INDEX_ROW3 ? 3 : c == CLASS_INDEX_ROW5 ? 5 : 1 << c; }
// work items that share a wavefront of the packed 16-bit sweep
DMND_HD int class_items_per_wave16(int P) { return row_class(P) ? 8 : 2; }

// ---- trace layout (traceback-mode sweeps -> traceback walk) --------------------------------------------------------------
// A PAIR-STEP t is the two anti-diagonal steps a_first + 2t (even: the cells of the even band diagonals) and a_first + 2t + 1
// (odd diagonals). One trace BYTE holds the two cells of band diagonals 2x and 2x + 1 of one pair-step: low nibble = the even
// diagonal's cell, high nibble = the odd one's (4 trace bits each, TB_*). x = 0 .. L P - 1 for an item of band class P swept by
// L = class_lanes(P) lanes (the lane-local pair p of lane l is x = l P + p). The bytes of one lane over G = trace_group(P)
// consecutive pair-steps are contiguous -- a 16-byte record for P <= 16 (G P = 16 bytes; 15 + one unused for the row classes
// P = 3 / 5) -- and the L records of a group follow each other lane by lane:
//     index(t, x) = (t / G) * (16 L) + (x / P) * 16 + (t % G) * P + x % P          (P >= 16: G = 1, index = t * 64 P + x)
// Why: the walk follows an alignment along a diagonal, i.e. through the SAME x over consecutive pair-steps. With one row per
// anti-diagonal step (the first layout: 64 P bytes per step, 4 bits used per byte) every column of the alignment was a
// different 128-byte line; here 16 / P consecutive columns of a diagonal are one record and a gap's neighbour diagonal is the
// neighbouring record or the same one. The sweep writes a record with ONE 16-byte store per lane and group (64 lanes = 1 KiB
// contiguous), half the bytes of before.
DMND_HD int trace_group(int P) { return P >= 16 ? 1 : 16 / P; }
DMND_HD int64_t trace_pairs(const Geom& g) { return (n_steps(g) + 1) / 2; }
DMND_HD int64_t trace_bytes(const Geom& g, int P)
{
	if (P >= 16) return trace_pairs(g) * 64 * P;
	const int G = 16 / P;
	return (trace_pairs(g) + G - 1) / G * (16 * class_lanes(P));
}
DMND_HD int64_t trace_byte_index(int P, int t, int x)
{
	if (P >= 16) return (int64_t)t * (64 * P) + x;
	const int G = 16 / P, lane = x / P;
	return (int64_t)(t / G) * (16 * class_lanes(P)) + lane * 16 + (t % G) * P + (x - lane * P);
}

// validity window of diagonal k (global index) on anti-diagonal a:  a_lo <= a <= a_hi
//   i >= 0 <=> a >= -d ; j >= 0 <=> a >= d ; i < qlen <=> a <= 2*qlen-2-d ; j < tlen <=> a <= 2*tlen-2+d
DMND_HD void diag_window(const Geom& g, int k, int& a_lo, int& a_span)
{
	const int d = g.d_begin + k;
	a_lo = d < 0 ? -d : d;
	const int a_hi = imin(2 * g.qlen - 2 - d, 2 * g.tlen - 2 + d);
	a_span = a_hi - a_lo;
	if (k >= g.band || a_span < 0) {              // never valid: (unsigned)(a - a_lo) is huge for every a
		a_lo = 0x3fffffff;
		a_span = 0;
	}
}

// "x is a better end cell than the current best" -- see header comment
DMND_HD bool better_end(int s, int j, int i, int bs, int bj, int bi)
{
	return s > bs || (s == bs && (j < bj || (j == bj && i > bi)));
}

// STAT: 0 = none, 1 = ForwardCell (a = identities, b = length), 2 = BackwardCell (a = mismatches,
// b = gap openings) -- the statistics carried along the arg-max path, stat_cell.h:47-279.
enum { STAT_NONE = 0, STAT_FWD = 1, STAT_BWD = 2 };

template<int N, bool ON> struct StatRegs { int Ha[N], Hb[N], Ea[N], Eb[N], Fa[N], Fb[N]; };
template<int N> struct StatRegs<N, false> {};

template<int P, bool COORDS, int STAT = STAT_NONE>
struct Lane {
	int H[2 * P], E[2 * P], F[2 * P];
	int a_lo[2 * P], a_span[2 * P];
	int best, best_i, best_j, best_a, best_b;
	StatRegs<2 * P, STAT != STAT_NONE> st;

	DMND_HD void init(const Geom& g, int lane)
	{
#pragma unroll
		for (int k = 0; k < 2 * P; ++k) {
			H[k] = E[k] = F[k] = 0;
			diag_window(g, 2 * P * lane + k, a_lo[k], a_span[k]);
			if constexpr (STAT != STAT_NONE) { st.Ha[k] = st.Hb[k] = st.Ea[k] = st.Eb[k] = st.Fa[k] = st.Fb[k] = 0; }
		}
		best = 0; best_i = 0; best_j = 0x7fffffff; best_a = 0; best_b = 0;
	}
};

// One cell with statistics (oracle/banded_swipe.c is the line-by-line restatement of stat_cell.h):
// set_max ties take the argument's statistics (E before F), the reset on a zero cell applies to the
// stored H statistics but not to the copy that opens a gap.
template<int STAT>
DMND_HD void cell_update_stats(int Hd, int s, bool ident, int E_in, int F_in, int go, int ge,
	int Hda, int Hdb, int Ea, int Eb, int Fa, int Fb,
	int& cur, int& E_out, int& F_out, int& ca, int& cb, int& ea, int& eb, int& fa, int& fb)
{
	int c = Hd + s, xa, xb;
	if (STAT == STAT_FWD) { xa = Hda + (ident ? 1 : 0); xb = Hdb + 1; Eb += 1; Fb += 1; }
	else { xa = Hda + (ident ? 0 : 1); xb = Hdb; }
	if (E_in >= c) { c = E_in; xa = Ea; xb = Eb; }
	if (F_in >= c) { c = F_in; xa = Fa; xb = Fb; }
	if (c < 0) c = 0;
	const int open = imax(c - go, 0);
	int oa = xa, ob = xb;
	if (STAT == STAT_BWD) ob += 1;
	if (c == 0) { xa = 0; xb = 0; }
	int e = imax(E_in - ge, 0), f = imax(F_in - ge, 0);
	if (open >= e) { e = open; Ea = oa; Eb = ob; }
	if (open >= f) { f = open; Fa = oa; Fb = ob; }
	cur = c; E_out = e; F_out = f; ca = xa; cb = xb; ea = Ea; eb = Eb; fa = Fa; fb = Fb;
}

// One cell. Returns the 4 trace bits when TRACE.  (cell_update.h:103-140 with values clamped at 0.)
template<bool TRACE>
DMND_HD int cell_update(int Hd, int s, int E_in, int F_in, int go, int ge, int& cur, int& E_out, int& F_out)
{
	int c = Hd + s;
	c = imax(c, E_in);
	c = imax(c, F_in);
	c = imax(c, 0);
	const int open = imax(c - go, 0);
	E_out = imax(imax(E_in - ge, 0), open);
	F_out = imax(imax(F_in - ge, 0), open);
	cur = c;
	if (TRACE)
		return (c == F_in ? TB_GAP_V : 0) | (c == E_in ? TB_GAP_H : 0) | (F_out == open ? TB_OPEN_V : 0) | (E_out == open ? TB_OPEN_H : 0);
	return 0;
}

// Sequence/matrix access policy used by the step: q/t are letter pointers, cbs may be null,
// M is the 32x32 int8 matrix (in LDS on the device).
// rev_q / rev_t >= 0 make the view read the sequences back to front (index x -> rev - x): the reversed
// pass of recompute_reversed() (swipe_wrapper.cpp:364-444) without materialising reversed copies.
struct SeqView {
	const int8_t* q;
	const int8_t* t;
	const int8_t* cbs;
	const int8_t* M;
	int rev_q = -1, rev_t = -1;      // qlen-1 / tlen-1 when reversed
};

DMND_HD int match_score(const SeqView& v, int i, int j)
{
	const int ql = v.q[i] & LETTER_MASK, tl = v.t[j] & LETTER_MASK;
	int s = v.M[tl * 32 + ql];
	if (v.cbs) s += v.cbs[i];
	return s;
}

DMND_HD int match_score_stats(const SeqView& v, int i, int j, bool& ident)
{
	const int ii = v.rev_q >= 0 ? v.rev_q - i : i, jj = v.rev_t >= 0 ? v.rev_t - j : j;
	const int ql = v.q[ii] & LETTER_MASK, traw = v.t[jj], tl = traw & LETTER_MASK;
	int s = v.M[tl * 32 + ql];
	if (v.cbs) s += v.cbs[ii];
	ident = ql == traw;                 // VectorIdMask compares the masked query letter with the RAW target byte (stat_cell.h:41-44)
	return s;
}

// One anti-diagonal step of one lane. PAR = parity of (a + d_begin): 0 -> the lane's even local
// diagonals are active and nb is F_out of lane-1's top diagonal; 1 -> odd diagonals, nb is E_out of
// lane+1's bottom diagonal. trace (if TRACE) points at this lane's P bytes of the step's trace row.
template<int P, bool COORDS, bool TRACE, int PAR, int STAT = STAT_NONE>
DMND_HD void lane_step(Lane<P, COORDS, STAT>& st, const Geom& g, const SeqView& v, int lane, int a, int nb, int go, int ge, uint8_t* trace,
	int nb_a = 0, int nb_b = 0)
{
#pragma unroll
	for (int p = 0; p < P; ++p) {
		const int k = 2 * p + PAR;
		int E_in, F_in, Ea = 0, Eb = 0, Fa = 0, Fb = 0;
		if (PAR == 0) {
			E_in = st.E[k + 1];
			F_in = p == 0 ? nb : st.F[k - 1];
			if constexpr (STAT != STAT_NONE) {
				Ea = st.st.Ea[k + 1]; Eb = st.st.Eb[k + 1];
				Fa = p == 0 ? nb_a : st.st.Fa[k - 1]; Fb = p == 0 ? nb_b : st.st.Fb[k - 1];
			}
		}
		else {
			E_in = p == P - 1 ? nb : st.E[k + 1];
			F_in = st.F[k - 1];
			if constexpr (STAT != STAT_NONE) {
				Ea = p == P - 1 ? nb_a : st.st.Ea[k + 1]; Eb = p == P - 1 ? nb_b : st.st.Eb[k + 1];
				Fa = st.st.Fa[k - 1]; Fb = st.st.Fb[k - 1];
			}
		}
		const bool valid = (unsigned)(a - st.a_lo[k]) <= (unsigned)st.a_span[k];
		int cur = 0, E_out = 0, F_out = 0, tb = 0;
		int ca = 0, cb = 0, ea = 0, eb = 0, fa = 0, fb = 0;
		if (valid) {
			const int d = g.d_begin + 2 * P * lane + k;
			const int i = (a + d) >> 1, j = (a - d) >> 1;
			if constexpr (STAT != STAT_NONE) {
				bool ident;
				const int s = match_score_stats(v, i, j, ident);
				cell_update_stats<STAT>(st.H[k], s, ident, E_in, F_in, go, ge, st.st.Ha[k], st.st.Hb[k], Ea, Eb, Fa, Fb,
					cur, E_out, F_out, ca, cb, ea, eb, fa, fb);
			}
			else {
				const int s = match_score(v, i, j);
				tb = cell_update<TRACE>(st.H[k], s, E_in, F_in, go, ge, cur, E_out, F_out);
			}
			if (COORDS) {
				if (better_end(cur, j, i, st.best, st.best_j, st.best_i)) {
					st.best = cur; st.best_j = j; st.best_i = i;
					if constexpr (STAT != STAT_NONE) { st.best_a = ca; st.best_b = cb; }
				}
			}
			else
				st.best = imax(st.best, cur);
		}
		st.H[k] = cur; st.E[k] = E_out; st.F[k] = F_out;
		if constexpr (STAT != STAT_NONE) {
			st.st.Ha[k] = ca; st.st.Hb[k] = cb; st.st.Ea[k] = ea; st.st.Eb[k] = eb; st.st.Fa[k] = fa; st.st.Fb[k] = fb;
		}
		if (TRACE)
			trace[p] = (uint8_t)tb;
	}
}

// ---- register-window lane (score-only / coordinates / traceback kernels) -----------------------------------------
// The lane's cells of one (even, odd) step pair sit on diagonals d0 .. d0 + 2P - 1 (d0 = d_begin + 2P * lane). With
// I0 = (a + d0) / 2 and J0 = (a - d0) / 2 for the even anti-diagonal a of the pair,
//     even cell p : (I0 + p,     J0 - p)         odd cell p : (I0 + p + 1, J0 - p)
// and the next pair is the same picture one row and one column further (I0 + 1, J0 + 1). So the lane keeps a window of
// P + 1 query letters (with their composition bias) and P target letters in registers and fetches exactly ONE new query
// letter, ONE bias byte and ONE target letter per step pair, a whole pair ahead of their first use -- instead of loading
// both letters and the bias of every cell at the cell (2P of each per pair, each load on the cell's critical path).
// Cells are computed branch-free: an invalid cell (outside the matrix or the band) computes on whatever letters the
// clamped window holds and is then forced to H = E = F = 0, which is what lane_step's skipped cells read as.
template<int P, bool COORDS>
struct WinLane {
	int H[2 * P], E[2 * P], F[2 * P];
	int rel[2 * P];                 // (anti-diagonal of diagonal k's next cell) - a_lo[k]; valid iff (unsigned)rel <= span
	int span[2 * P];
	int Q[P + 1], C[P + 1];         // query letters (masked) and bias at rows I0 .. I0 + P
	int T[P];                       // (target letter << 5) at columns J0 - p: the matrix row offset
	int iq, jt;                     // row / column of the next letters to fetch (I0 + P + 1, J0 + 1)
	int best, best_i, best_j;
	// COORDS, P <= WIN_DIAG_BEST_MAX_P: best score of every diagonal and the anti-diagonal of its FIRST occurrence. Along one
	// diagonal the column grows with the anti-diagonal, so "first" is the reference's tie-break (smallest column) and no
	// coordinates are needed per cell; win_finish picks the lane's end cell from these. Wider classes keep one end cell per lane.
	int bk[COORDS && P <= 4 ? 2 * P : 1], ba[COORDS && P <= 4 ? 2 * P : 1];
};
enum { WIN_DIAG_BEST_MAX_P = 4 };

// index into a sequence of hi + 1 letters, any x: rows / columns outside the sequence belong to invalid cells only, which
// may read any letter (a negative x wraps to a huge unsigned value and lands on hi)
DMND_HD int clampi(int x, int hi) { return (int)((unsigned)x < (unsigned)hi ? (unsigned)x : (unsigned)hi); }

template<int P, bool COORDS>
DMND_HD void win_init(WinLane<P, COORDS>& st, const Geom& g, const SeqView& v, int lane)
{
#pragma unroll
	for (int k = 0; k < 2 * P; ++k) {
		st.H[k] = st.E[k] = st.F[k] = 0;
		int a_lo, a_span;
		diag_window(g, 2 * P * lane + k, a_lo, a_span);
		st.rel[k] = g.a_first + (k & 1) - a_lo;       // never-valid diagonals: a_lo = 0x3fffffff keeps this negative for the whole sweep
		st.span[k] = a_span;
	}
	const int d0 = g.d_begin + 2 * P * lane;
	const int I0 = (g.a_first + d0) >> 1, J0 = (g.a_first - d0) >> 1;
#pragma unroll
	for (int x = 0; x <= P; ++x) {
		const int i = clampi(I0 + x, g.qlen - 1);
		st.Q[x] = v.q[i] & LETTER_MASK;
		st.C[x] = v.cbs ? v.cbs[i] : 0;
	}
#pragma unroll
	for (int p = 0; p < P; ++p)
		st.T[p] = (v.t[clampi(J0 - p, g.tlen - 1)] & LETTER_MASK) << 5;
	st.iq = I0 + P + 1; st.jt = J0 + 1;
	st.best = 0; st.best_i = 0; st.best_j = 0x7fffffff;
	if (COORDS && P <= WIN_DIAG_BEST_MAX_P) {
#pragma unroll
		for (int k = 0; k < 2 * P; ++k) { st.bk[k] = 0; st.ba[k] = 0; }
	}
}

// after the sweep: the lane's end cell (best, best_i, best_j) from the per-diagonal records
template<int P, bool COORDS>
DMND_HD void win_finish(WinLane<P, COORDS>& st, int d0)
{
	if (COORDS && P <= WIN_DIAG_BEST_MAX_P) {
#pragma unroll
		for (int k = 0; k < 2 * P; ++k) {
			const int d = d0 + k, i = (st.ba[k] + d) >> 1, j = (st.ba[k] - d) >> 1;
			if (st.bk[k] > 0 && better_end(st.bk[k], j, i, st.best, st.best_j, st.best_i)) { st.best = st.bk[k]; st.best_j = j; st.best_i = i; }
		}
	}
}

// One anti-diagonal step (PAR as in lane_step; `a` = its anti-diagonal, d0 = the lane's first diagonal: only read when a new
// end cell is recorded). tb4 (TRACE): receives the step's P trace bytes (one cell each, 4 bits used), four to a dword.
template<int P, bool COORDS, bool TRACE, int PAR>
DMND_HD void win_step(WinLane<P, COORDS>& st, const int8_t* M, int nb, int go, int ge, int a, int d0, uint32_t* tb4)
{
	uint32_t packed = 0;
#pragma unroll
	for (int p = 0; p < P; ++p) {
		const int k = 2 * p + PAR;
		int E_in, F_in;
		if (PAR == 0) { E_in = st.E[k + 1]; F_in = p == 0 ? nb : st.F[k - 1]; }
		else { E_in = p == P - 1 ? nb : st.E[k + 1]; F_in = st.F[k - 1]; }
		const bool valid = (unsigned)st.rel[k] <= (unsigned)st.span[k];
		st.rel[k] += 2;
		const int s = M[st.T[p] | st.Q[p + PAR]] + st.C[p + PAR];
		int c = imax(imax(st.H[k] + s, E_in), imax(F_in, 0));
		const int open = imax(c - go, 0);
		const int e = imax(E_in - ge, open), f = imax(F_in - ge, open);        // open >= 0 already clamps them
		if (TRACE) {
			const uint32_t tb = (c == F_in ? TB_GAP_V : 0) | (c == E_in ? TB_GAP_H : 0) | (f == open ? TB_OPEN_V : 0) | (e == open ? TB_OPEN_H : 0);
			packed |= tb << (8 * (p & 3));
			if (P >= 4 && (p & 3) == 3) { tb4[p >> 2] = packed; packed = 0; }
		}
		c = valid ? c : 0;
		st.H[k] = c; st.E[k] = valid ? e : 0; st.F[k] = valid ? f : 0;
		if (COORDS && P <= WIN_DIAG_BEST_MAX_P) {
			const bool up = c > st.bk[k];                // invalid cells carry c = 0 and never improve a record
			st.ba[k] = up ? a : st.ba[k];
			st.bk[k] = imax(st.bk[k], c);
		}
		else if (COORDS) {
			if (valid && c >= st.best) {
				const int d = d0 + k, i = (a + d) >> 1, j = (a - d) >> 1;
				if (better_end(c, j, i, st.best, st.best_j, st.best_i)) { st.best = c; st.best_j = j; st.best_i = i; }
			}
		}
		else
			st.best = imax(st.best, c);
	}
	if (TRACE && P < 4) tb4[0] = packed;
}

// The lane's P bytes of one pair-step from the two steps' cells (tb_even / tb_odd as win_step returns them), stored at `rec` =
// the lane's record position of that pair-step: trace + trace_byte_index(P_class, t, lane * P)
template<int P>
DMND_HD void win_store_trace(uint8_t* rec, const uint32_t* tb_even, const uint32_t* tb_odd)
{
	if (P == 1) rec[0] = (uint8_t)(tb_even[0] | (tb_odd[0] << 4));
	else if (P == 2) *reinterpret_cast<uint16_t*>(rec) = (uint16_t)(tb_even[0] | (tb_odd[0] << 4));
	else {
#pragma unroll
		for (int x = 0; x < P / 4; ++x) reinterpret_cast<uint32_t*>(rec)[x] = tb_even[x] | (tb_odd[x] << 4);
	}
}

// end of a step pair: the window moves one row down and one column right; nq / nc / nt are the letters of row st.iq and
// column st.jt (clamped into the sequences by the caller, which fetched them before the pair's two steps)
template<int P, bool COORDS>
DMND_HD void win_advance(WinLane<P, COORDS>& st, int nq, int nc, int nt)
{
#pragma unroll
	for (int x = 0; x < P; ++x) { st.Q[x] = st.Q[x + 1]; st.C[x] = st.C[x + 1]; }
	st.Q[P] = nq & LETTER_MASK; st.C[P] = nc;
#pragma unroll
	for (int p = P - 1; p > 0; --p) st.T[p] = st.T[p - 1];
	st.T[0] = (nt & LETTER_MASK) << 5;
	++st.iq; ++st.jt;
}

// Geometry of the reversed pass of recompute_reversed() (swipe_wrapper.cpp:378-391): the target is the
// prefix [0, s_end) of the forward target, both sequences are read back to front, the band is mirrored
// with Geo::rev_diag (util/geo/geo.h:37).
DMND_HD void reversed_band(int qlen, int s_end, int d_begin, int d_end, int& r_tlen, int& r_d_begin, int& r_d_end)
{
	r_tlen = s_end;
	r_d_begin = -(d_end - 1) + qlen - s_end;
	r_d_end = -d_begin + qlen - s_end + 1;
}

// ---- traceback walk over the anti-diagonal trace (one thread per item) -------------------------
// Follows banded_swipe.h:128-183 + TracebackVectorMatrix::TracebackIterator (banded_matrix.h:359-408)
// and the accounting of Hsp::push_match / push_gap (basic/hssp.cpp:260-290).
// trace layout: trace_byte_index above; P = the item's band class (for an item swept by several wavefronts: P per lane x wavefronts).
struct WalkResult {
	int q_begin, s_begin, length, identities, mismatches, positives, gap_openings, gaps, transcript_len, status;
};

DMND_HD uint8_t trace_at(const uint8_t* trace, const Geom& g, int P, int i, int j)
{
	const int a = i + j, k = i - j - g.d_begin;
	if (k < 0 || k >= g.band || a < g.a_first)       // cannot happen on a valid path; keeps the walk memory-safe
		return TB_OPEN_V | TB_OPEN_H;
	// a_first + d_begin is even (make_geom), so the parity of the step inside its pair is the parity of the diagonal
	const uint8_t b = trace[trace_byte_index(P, (a - g.a_first) >> 1, k >> 1)];
	return (k & 1) ? (uint8_t)(b >> 4) : (uint8_t)(b & 15);
}

// transcript receives the packed operations in forward order followed by a 0 terminator;
// cap is the number of bytes available (including the terminator).
DMND_HD WalkResult traceback_walk(const uint8_t* trace, const Geom& g, int P, const SeqView& v, int gap_open, int gap_extend,
	int best, int end_i, int end_j, uint8_t* transcript, int cap)
{
	WalkResult r;
	r.length = r.identities = r.mismatches = r.positives = r.gap_openings = r.gaps = 0;
	r.status = 0;
	int i = end_i, j = end_j, sc = 0, n = 0;
	// written backwards from the end of the slot, then moved to the front
	while (i >= 0 && j >= 0 && sc < best) {
		const uint8_t m = trace_at(trace, g, P, i, j);
		if ((m & (TB_GAP_V | TB_GAP_H)) == 0) {
			const int ql = v.q[i] & LETTER_MASK, tl = v.t[j] & LETTER_MASK;
			int s = v.M[tl * 32 + ql];
			const bool positive = s > 0;
			if (v.cbs) s += v.cbs[i];
			sc += s;
			if (n < cap - 1) transcript[cap - 2 - n] = (uint8_t)(ql == tl ? (OP_MATCH << OP_COUNT_BITS) | 1 : (OP_SUBSTITUTION << OP_COUNT_BITS) | tl);
			++n;
			if (ql == tl) { ++r.identities; ++r.positives; }
			else { ++r.mismatches; if (positive) ++r.positives; }
			++r.length;
			--i; --j;
		}
		else {
			int l = 0;
			if (m & TB_GAP_V) {
				do { ++l; --i; } while (i > 0 && (trace_at(trace, g, P, i, j) & TB_OPEN_V) == 0);
				// reference loop: do { ++l; --i; --mask; } while (!(mask->open & v) && i > 0)
				int c = l;
				while (c > 0) {
					const int kk = imin(c, (int)OP_MAX_COUNT);
					if (n < cap - 1) transcript[cap - 2 - n] = (uint8_t)((OP_INSERTION << OP_COUNT_BITS) | kk);
					++n; c -= kk;
				}
			}
			else {
				const int j_before = j;
				do { ++l; --j; } while (j > 0 && (trace_at(trace, g, P, i, j) & TB_OPEN_H) == 0);
				for (int x = 0; x < l; ++x) {
					if (n < cap - 1) transcript[cap - 2 - n] = (uint8_t)((OP_DELETION << OP_COUNT_BITS) | (v.t[j_before - x] & LETTER_MASK));
					++n;
				}
			}
			++r.gap_openings;
			r.length += l;
			r.gaps += l;
			sc -= gap_open + l * gap_extend;
		}
	}
	if (sc != best) r.status = -6;           // DMND_E_TRACEBACK
	if (n > cap - 1) { r.status = -5; n = 0; }    // DMND_E_CAP
	// move to the front of the slot (already in forward order because it was written from the back)
	for (int x = 0; x < n; ++x) transcript[x] = transcript[cap - 1 - n + x];
	if (cap > 0) transcript[n] = 0;
	r.q_begin = i + 1; r.s_begin = j + 1; r.transcript_len = n;
	return r;
}

}  // namespace dmnd
