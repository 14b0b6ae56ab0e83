// swipe16_core.h -- per-lane arithmetic of the packed-int16, two-items-per-wavefront banded Smith-Waterman sweep.
//
// Same recurrence, band semantics and tie rules as swipe_core.h (reference: DP::BandedSwipe::swipe,
// /root/reference/src/dp/swipe/banded_swipe.h:189-351, cell_update.h:103-140; the reference's own 16-bit score vectors:
// src/dp/score_vector_int16.h, with escalation to 32 bits on overflow: swipe_wrapper.cpp:317-360), re-formulated for the
// CDNA4 packed 16-bit VALU (v_pk_add_i16 / v_pk_max_i16 / v_pk_sub_u16 clamp):
//   * a wavefront sweeps TWO work items at once: item A lives in the low 16 bits of every DP register, item B in the high
//     16 bits. Both items use the same lane -> diagonal map (lane l owns band diagonals 2P*l .. 2P*l + 2P-1 of its item) but
//     their own coordinates: pair-step t of an item covers its anti-diagonals a_first + 2t and a_first + 2t + 1.
//   * no validity predicate per cell. A cell outside its matrix is computed on the sentinel letter 31, whose row and column
//     of the LDS score table hold -128: in the corner before the matrix every value stays 0 by induction, and in the corner
//     behind it every value is strictly below a score already seen inside the matrix (it descends from one through at least
//     one sentinel score or gap penalty), so neither region can reach a matrix cell or the best-score record. Diagonals above
//     the band are zeroed with one static AND per cell (m[]), which is all the band's upper edge needs: E only flows to lower
//     diagonals, F only to higher ones. The lower edge is lane 0's DPP bound (0).
//   * letters are not fetched per lane: a lane's window is its neighbour's window one pair-step later (rows come down from
//     lane l+1, columns from lane l-1), so the windows -- packed like the DP registers, both items in one register -- move
//     through the wave with three DPP shifts per pair-step and only the two edge lanes take new letters, from a wave-uniform
//     edge record (sw16_edge) that the kernel prepares in LDS a chunk of pair-steps at a time.
//   * the end cell is recovered from one 32-bit max per diagonal and item over keys (H << 16 | 0xffff - t): equal scores keep
//     the earliest pair-step, i.e. the smallest column of that diagonal -- the reference's tie rule.
//   * scores saturate at 32767 (clamped adds): an item whose best score reaches it is re-run in the 32-bit kernel.
// Shared by swipe16_kernels.hip and the CPU emulator (tests/emu/swipe16_emu.cpp).
#pragma once
#include <stdint.h>
#include "swipe_core.h"

namespace dmnd {

typedef uint32_t pk16;                       // two int16 halves: low = item A, high = item B
enum { SW16_MAX_SCORE = 32767, SW16_MAX_PAIRS = 65535, SW16_SENTINEL = 31, SW16_MAX_P = 4, SW16_Q_SHIFT = 1, SW16_T_SHIFT = 6 };
// the classes of these kernels: P = 1, 2, 4 on the 64 lanes of a wavefront (two items per wavefront), the row classes P = 3, 5 on the
// 16 lanes of a DPP row (eight items per wavefront: swipe_core.h row_class)
DMND_HD bool sw16_class(int P) { return P <= SW16_MAX_P || P == 5; }
// pair-steps of one trace record = trace_group(P) of swipe_core.h for the classes of this kernel: the sweep runs in groups
template<int P> struct Sw16Group { enum { G = 16 / P }; };

#if defined(__HIP_DEVICE_COMPILE__)
typedef short sw16_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short sw16_u2 __attribute__((ext_vector_type(2)));
DMND_HD sw16_s2 sw16_s(pk16 x) { return __builtin_bit_cast(sw16_s2, x); }
DMND_HD sw16_u2 sw16_u(pk16 x) { return __builtin_bit_cast(sw16_u2, x); }
DMND_HD pk16 pk_adds(pk16 a, pk16 b) { return __builtin_bit_cast(pk16, __builtin_elementwise_add_sat(sw16_s(a), sw16_s(b))); }      // v_pk_add_i16 clamp
DMND_HD pk16 pk_add(pk16 a, pk16 b) { return __builtin_bit_cast(pk16, (sw16_u2)(sw16_u(a) + sw16_u(b))); }                          // v_pk_add_u16
DMND_HD pk16 pk_sub(pk16 a, pk16 b) { return __builtin_bit_cast(pk16, (sw16_u2)(sw16_u(a) - sw16_u(b))); }                          // v_pk_sub_u16
DMND_HD pk16 pk_subus(pk16 a, pk16 b) { return __builtin_bit_cast(pk16, __builtin_elementwise_sub_sat(sw16_u(a), sw16_u(b))); }     // v_pk_sub_u16 clamp
DMND_HD pk16 pk_max(pk16 a, pk16 b) { return __builtin_bit_cast(pk16, __builtin_elementwise_max(sw16_s(a), sw16_s(b))); }           // v_pk_max_i16
DMND_HD pk16 pk_twice_plus(pk16 a, pk16 b)                    // 2a + b per half: one v_pk_mad_u16 (the compiler would split it into shift + or)
{
	pk16 r;
	asm("v_pk_mad_u16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
#else
inline int16_t sw16_lo(pk16 x) { return (int16_t)(x & 0xffffu); }
inline int16_t sw16_hi(pk16 x) { return (int16_t)(x >> 16); }
inline pk16 sw16_mk(int lo, int hi) { return ((uint32_t)(uint16_t)lo) | ((uint32_t)(uint16_t)hi << 16); }
inline int sw16_sat(int x) { return x > 32767 ? 32767 : x < -32768 ? -32768 : x; }
inline pk16 pk_adds(pk16 a, pk16 b) { return sw16_mk(sw16_sat(sw16_lo(a) + sw16_lo(b)), sw16_sat(sw16_hi(a) + sw16_hi(b))); }
inline pk16 pk_add(pk16 a, pk16 b) { return sw16_mk(sw16_lo(a) + sw16_lo(b), sw16_hi(a) + sw16_hi(b)); }
inline pk16 pk_sub(pk16 a, pk16 b) { return sw16_mk(sw16_lo(a) - sw16_lo(b), sw16_hi(a) - sw16_hi(b)); }
inline pk16 pk_subus(pk16 a, pk16 b)
{
	const int al = a & 0xffff, bl = b & 0xffff, ah = a >> 16, bh = b >> 16;
	return sw16_mk(al > bl ? al - bl : 0, ah > bh ? ah - bh : 0);
}
inline pk16 pk_max(pk16 a, pk16 b) { return sw16_mk(imax(sw16_lo(a), sw16_lo(b)), imax(sw16_hi(a), sw16_hi(b))); }
inline pk16 pk_twice_plus(pk16 a, pk16 b) { return sw16_mk(2 * (int)(a & 0xffff) + (int)(b & 0xffff), 2 * (int)(a >> 16) + (int)(b >> 16)); }
#endif

// keeps a value where it is computed (device). The traceback sweep ends a group of pair-steps with conditional stores, and the
// compiler sinks whatever is only needed after them -- the packing of the trace nibbles, the updates of the end-cell keys -- past
// the branch, keeping every step's cell values alive until then (+60 registers at P = 1)
DMND_HD void pin_here(uint32_t& x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("" : "+v"(x));
#else
	(void)x;
#endif
}

DMND_HD pk16 pk_both(int x) { return ((uint32_t)(uint16_t)x) | ((uint32_t)(uint16_t)x << 16); }
DMND_HD pk16 pk_make(int lo, int hi) { return ((uint32_t)(uint16_t)lo) | ((uint32_t)(uint16_t)hi << 16); }

// LDS score table of the sweep: the substitution matrix with the sentinel's row and column at -128, one 16-bit entry per
// letter pair (two lookups then pack into a register with one v_lshl_or_b32; gfx950's d16 loads clear the other half)
DMND_HD uint16_t sw16_table_entry(const int8_t* matrix, int x) { return (uint16_t)(int16_t)(((x >> 5) == SW16_SENTINEL || (x & 31) == SW16_SENTINEL) ? -128 : (int)matrix[x]); }

// letter of row / column idx of a sequence of len letters: the sentinel outside [0, len)
DMND_HD int sw16_letter(const int8_t* s, int idx, int len)
{
	const int l = s[clampi(idx, len - 1)] & LETTER_MASK;
	return (unsigned)idx < (unsigned)len ? l : (int)SW16_SENTINEL;
}
DMND_HD int sw16_bias(const int8_t* cbs, int idx, int len)
{
	if (!cbs) return 0;
	const int c = cbs[clampi(idx, len - 1)];
	return (unsigned)idx < (unsigned)len ? c : 0;
}

template<int P>
struct Lane16 {
	pk16 H[2 * P], E[2 * P], F[2 * P];
	pk16 m[2 * P];                          // 0xffff in an item's half iff the diagonal lies inside that item's band
	uint32_t keyA[2 * P], keyB[2 * P];      // max over the sweep of (H << 16 | 0xffff - pair-step), per diagonal and item
	pk16 best;                              // score-only sweeps (COORDS = false in lane16_step) keep this instead: the lane's best score of either item
	// letter windows, packed like the DP registers (A low, B high):
	pk16 QQ[P + 1];                         // (query letter << 1) of rows I0 + t .. I0 + t + P
	pk16 CC[P + 1];                         // their composition bias
	pk16 TT[P];                             // (target letter << 6) of columns J0 + t - p: QQ | TT = byte offsets into the score table
};

// first row / column of a lane's windows at pair-step 0 (swipe_core.h: I0, J0 of WinLane)
DMND_HD int sw16_row0(const Geom& g, int P, int lane) { return (g.a_first + g.d_begin + 2 * P * lane) >> 1; }
DMND_HD int sw16_col0(const Geom& g, int P, int lane) { return (g.a_first - g.d_begin - 2 * P * lane) >> 1; }
// pair-steps of an item = trace rows / 2
DMND_HD int sw16_pairs(const Geom& g) { return (int)(trace_rows(g) / 2); }

// window entries of row i / column j of both items
DMND_HD pk16 sw16_qq(const Geom& gA, const SeqView& vA, const Geom& gB, const SeqView& vB, int iA, int iB)
{
	return pk_make(sw16_letter(vA.q, iA, gA.qlen) << SW16_Q_SHIFT, sw16_letter(vB.q, iB, gB.qlen) << SW16_Q_SHIFT);
}
DMND_HD pk16 sw16_cc(const Geom& gA, const SeqView& vA, const Geom& gB, const SeqView& vB, int iA, int iB)
{
	return pk_make(sw16_bias(vA.cbs, iA, gA.qlen), sw16_bias(vB.cbs, iB, gB.qlen));
}
DMND_HD pk16 sw16_tt(const Geom& gA, const SeqView& vA, const Geom& gB, const SeqView& vB, int jA, int jB)
{
	return pk_make(sw16_letter(vA.t, jA, gA.tlen) << SW16_T_SHIFT, sw16_letter(vB.t, jB, gB.tlen) << SW16_T_SHIFT);
}

// What enters the edge lanes of an item pair at the end of pair-step t: the row that becomes the last lane's (63; 15 for a row
// class) top row and the column that becomes lane 0's newest column (the same for all lanes of the pair; the kernel keeps a chunk
// of these records in LDS)
struct Edge16 { pk16 qq, tt, cc; };
DMND_HD Edge16 sw16_edge(const Geom& gA, const SeqView& vA, const Geom& gB, const SeqView& vB, int P, int t, int last_lane = 63)
{
	const int iA = sw16_row0(gA, P, last_lane) + P + 1 + t, iB = sw16_row0(gB, P, last_lane) + P + 1 + t;
	const int jA = sw16_col0(gA, P, 0) + 1 + t, jB = sw16_col0(gB, P, 0) + 1 + t;
	Edge16 e;
	e.qq = sw16_qq(gA, vA, gB, vB, iA, iB);
	e.cc = sw16_cc(gA, vA, gB, vB, iA, iB);
	e.tt = sw16_tt(gA, vA, gB, vB, jA, jB);
	return e;
}

template<int P>
DMND_HD void lane16_init(Lane16<P>& st, const Geom& gA, const SeqView& vA, const Geom& gB, const SeqView& vB, int lane)
{
#pragma unroll
	for (int k = 0; k < 2 * P; ++k) {
		st.H[k] = st.E[k] = st.F[k] = 0;
		st.keyA[k] = st.keyB[k] = 0;
		st.best = 0;
		const int kk = 2 * P * lane + k;
		st.m[k] = (kk < gA.band ? 0xffffu : 0u) | (kk < gB.band ? 0xffff0000u : 0u);
	}
	const int iA = sw16_row0(gA, P, lane), iB = sw16_row0(gB, P, lane), jA = sw16_col0(gA, P, lane), jB = sw16_col0(gB, P, lane);
#pragma unroll
	for (int x = 0; x <= P; ++x) {
		st.QQ[x] = sw16_qq(gA, vA, gB, vB, iA + x, iB + x);
		st.CC[x] = sw16_cc(gA, vA, gB, vB, iA + x, iB + x);
	}
#pragma unroll
	for (int p = 0; p < P; ++p)
		st.TT[p] = sw16_tt(gA, vA, gB, vB, jA - p, jB - p);
}

// table entry at byte offset oa in the low half, at byte offset ob in the high half
DMND_HD pk16 sw16_lookup(const uint16_t* table, uint32_t oa, uint32_t ob)
{
	const char* b = reinterpret_cast<const char*>(table);
	return (uint32_t)*reinterpret_cast<const uint16_t*>(b + oa) | ((uint32_t)*reinterpret_cast<const uint16_t*>(b + ob) << 16);
}

// substitution score + bias of the lane's cells of one pair-step from its current windows: S0[p] even step, S1[p] odd step
// (the two table offsets of a cell pair are the halves of TT | QQ: one v_or_b32 with sub-dword operand selects each)
template<int P>
DMND_HD void lane16_scores(const Lane16<P>& st, const uint16_t* table, pk16* S0, pk16* S1)
{
#pragma unroll
	for (int p = 0; p < P; ++p) {
		const pk16 o0 = st.TT[p] | st.QQ[p], o1 = st.TT[p] | st.QQ[p + 1];
		S0[p] = pk_add(sw16_lookup(table, o0 & 0xffffu, o0 >> 16), st.CC[p]);
		S1[p] = pk_add(sw16_lookup(table, o1 & 0xffffu, o1 >> 16), st.CC[p + 1]);
	}
}

// The windows move one row down and one column right. nqq / ncc: the new top row (lane l+1's QQ[1] / CC[1] before ITS
// advance; the edge record in lane 63). ntt: the new column (lane l-1's TT[P-1] before its advance; the edge record in lane 0).
template<int P>
DMND_HD void lane16_advance(Lane16<P>& st, pk16 nqq, pk16 ncc, pk16 ntt)
{
#pragma unroll
	for (int x = 0; x < P; ++x) { st.QQ[x] = st.QQ[x + 1]; st.CC[x] = st.CC[x + 1]; }
	st.QQ[P] = nqq; st.CC[P] = ncc;
#pragma unroll
	for (int p = P - 1; p > 0; --p) st.TT[p] = st.TT[p - 1];
	st.TT[0] = ntt;
}

// One anti-diagonal step of both items. PAR 0: the lane's even diagonals, nb = F of lane-1's top diagonal; PAR 1: odd
// diagonals, nb = E of lane+1's bottom diagonal (as lane_step / win_step in swipe_core.h). S = the P packed scores of this
// step, go = gap_open + gap_extend and ge = gap_extend in both halves, revt = 0xffff - pair-step.
// tb[p] (TRACE) receives the cell's four trace bits of item A in bits 0-3 and of item B in bits 16-19.
// COORDS = false: only the best score is wanted (the first pass of an extension whose survivors are swept again, DMND_SWIPE_SCORE):
// one packed max per cell instead of the two 32-bit keys (4 of a cell's 18 instructions).
template<int P, bool TRACE, int PAR, bool COORDS = true>
DMND_HD void lane16_step(Lane16<P>& st, const pk16* S, pk16 nb, pk16 go, pk16 ge, uint32_t revt, pk16* tb)
{
	const pk16 one = pk_both(1);
#pragma unroll
	for (int p = 0; p < P; ++p) {
		const int k = 2 * p + PAR;
		pk16 E_in, F_in;
		if (PAR == 0) { E_in = st.E[k + 1]; F_in = p == 0 ? nb : st.F[k - 1]; }
		else { E_in = p == P - 1 ? nb : st.E[k + 1]; F_in = st.F[k - 1]; }
		// E_in, F_in >= 0 always, so the recurrence's clamp at 0 is implied
		const pk16 c = pk_max(pk_max(pk_adds(st.H[k], S[p]), E_in), F_in) & st.m[k];
		const pk16 open = pk_subus(c, go);
		const pk16 e = pk_max(pk_subus(E_in, ge), open), f = pk_max(pk_subus(F_in, ge), open);
		if (TRACE) {
			// flag = 1 - min(difference, 1); every difference is >= 0 inside the band
			const pk16 gv = pk_subus(one, pk_sub(c, F_in)), gh = pk_subus(one, pk_sub(c, E_in));
			const pk16 ov = pk_subus(one, pk_sub(f, open)), oh = pk_subus(one, pk_sub(e, open));
			tb[p] = pk_twice_plus(pk_twice_plus(pk_twice_plus(oh, ov), gh), gv);      // TB_OPEN_H 8 | TB_OPEN_V 4 | TB_GAP_H 2 | TB_GAP_V 1
		}
		st.H[k] = c; st.E[k] = e; st.F[k] = f;
		if (!COORDS) { st.best = pk_max(st.best, c); continue; }
		const uint32_t ka = (c << 16) | revt, kb = (c & 0xffff0000u) | revt;
		st.keyA[k] = st.keyA[k] > ka ? st.keyA[k] : ka;
		st.keyB[k] = st.keyB[k] > kb ? st.keyB[k] : kb;
		if (TRACE) { pin_here(st.keyA[k]); pin_here(st.keyB[k]); }      // see pin_here: else every cell value of a group stays alive to its end
	}
}

// ---- trace records (layout: swipe_core.h, trace_byte_index) --------------------------------------------------------------
// v_perm_b32: byte x of the result = byte sel[x] (0-7) of the 8 bytes { hi : lo }
DMND_HD uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_perm(hi, lo, sel);
#else
	const uint64_t v = ((uint64_t)hi << 32) | lo;
	uint32_t r = 0;
	for (int x = 0; x < 4; ++x) r |= (uint32_t)((v >> (8 * ((sel >> (8 * x)) & 7))) & 0xffu) << (8 * x);
	return r;
#endif
}

// A lane's 16-byte trace records of both items over one group of G = 16 / P pair-steps, collected in registers and written
// with one 16-byte store per item and group. put<R>() takes pair-step R of the group (a compile-time position: the kernel
// unrolls a group): tb0 / tb1 = the P trace nibbles of the even / odd step as lane16_step returns them (item A in bits 0-3,
// item B in bits 16-19). Byte (R * P + p) of an item's record = tb0[p] | tb1[p] << 4.
template<int P>
struct Trace16Group {
	uint32_t a[4], b[4];                    // the records of item A / item B
	uint32_t h0, h1;                        // partly filled words: [A A B B] byte pairs waiting for their other half
	template<int R>
	DMND_HD void put(const pk16* tb0, const pk16* tb1)
	{
		constexpr uint32_t ZIP = 0x06020400u;            // { hi, lo } -> [lo.b0, hi.b0, lo.b2, hi.b2]
		constexpr uint32_t LOW = 0x05040100u, HIGH = 0x07060302u;      // { hi, lo } -> [lo.h0, hi.h0] / [lo.h1, hi.h1]
		uint32_t w[P];
#pragma unroll
		for (int p = 0; p < P; ++p) w[p] = tb0[p] | (tb1[p] << 4);      // A's byte in bits 0-7, B's in bits 16-23
		if (P == 4) {
			const uint32_t x01 = byte_perm(w[1], w[0], ZIP), x23 = byte_perm(w[P - 1], w[P == 4 ? 2 : 0], ZIP);
			a[R] = byte_perm(x23, x01, LOW); b[R] = byte_perm(x23, x01, HIGH);
			pin_here(a[R]); pin_here(b[R]);
		}
		else if (P == 2) {
			const uint32_t x = byte_perm(w[P - 1], w[0], ZIP);         // [A0 A1 B0 B1] of this pair-step
			if (R % 2 == 0) { h0 = x; pin_here(h0); }
			else { a[R / 2] = byte_perm(x, h0, LOW); b[R / 2] = byte_perm(x, h0, HIGH); pin_here(a[R / 2]); pin_here(b[R / 2]); }
		}
		else if (P == 1) {
			if (R % 2 == 0) { h0 = w[0]; pin_here(h0); }
			else {
				uint32_t x = byte_perm(w[0], h0, ZIP);                   // [A(R-1) A(R) B(R-1) B(R)]
				if (R % 4 == 1) { h1 = x; pin_here(h1); }
				else { a[R / 4] = byte_perm(x, h1, LOW); b[R / 4] = byte_perm(x, h1, HIGH); pin_here(a[R / 4]); pin_here(b[R / 4]); }
			}
		}
		else {
			// the row classes (P = 3, 5: G P = 15 bytes, the 16th is not used): byte n = R P + p of the record, taken as they come --
			// two bytes zip into [A A B B], two such halves make a word of either record
#pragma unroll
			for (int p = 0; p < P; ++p) {
				const int n = R * P + p;
				if (n % 2 == 0 && n != 14) { h0 = w[p]; pin_here(h0); }
				else {
					const uint32_t x = n == 14 ? byte_perm(0u, w[p], ZIP) : byte_perm(w[p], h0, ZIP);      // [A(n-1) A(n) B(n-1) B(n)]; the last: [A14 0 B14 0]
					if (n % 4 == 1) { h1 = x; pin_here(h1); }
					else { a[n / 4] = byte_perm(x, h1, LOW); b[n / 4] = byte_perm(x, h1, HIGH); pin_here(a[n / 4]); pin_here(b[n / 4]); }
				}
			}
		}
	}
};

// after the sweep: the lane's end cell of one item from its per-diagonal keys (COORDS = false: its best score, no cell)
template<int P, bool COORDS = true>
DMND_HD void lane16_finish(const Lane16<P>& st, const Geom& g, bool second, int lane, int& bs, int& bi, int& bj)
{
	bs = 0; bi = 0; bj = 0x7fffffff;
	if (!COORDS) { bs = (int)(int16_t)(second ? st.best >> 16 : st.best & 0xffffu); return; }
#pragma unroll
	for (int k = 0; k < 2 * P; ++k) {
		const uint32_t key = second ? st.keyB[k] : st.keyA[k];
		const int s = (int)(key >> 16), t = 0xffff - (int)(key & 0xffffu);
		const int a = g.a_first + 2 * t + (k & 1), d = g.d_begin + 2 * P * lane + k;
		const int i = (a + d) >> 1, j = (a - d) >> 1;
		if (s > 0 && better_end(s, j, i, bs, bj, bi)) { bs = s; bi = i; bj = j; }
	}
}

}  // namespace dmnd
