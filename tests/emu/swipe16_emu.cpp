// tests/emu/swipe16_emu.cpp -- TEST INFRASTRUCTURE ONLY.
// CPU lane-emulator of the packed-int16 two-items-per-wavefront sweep (diamond_amd/csrc/swipe16_kernels.hip): runs the SAME
// per-lane code (diamond_amd/csrc/swipe16_core.h) for 64 emulated lanes (16 for the row classes P = 3, 5: one DPP row) in lock-step, DPP shifts replaced by array indexing,
// including the systolic letter flow (rows come down from lane l+1, columns from lane l-1, edge lanes read memory). The trace
// is collected and stored as the kernel does it (Trace16Group, one 16-byte record per lane, item and group of pair-steps) in
// the layout of swipe_core.h, so the reference walk (traceback_walk) decodes it.
#include <vector>
#include <cstring>
#include "../../diamond_amd/csrc/swipe16_core.h"

using namespace dmnd;

struct Emu16Out {
	int32_t score, q_begin, q_end, s_begin, s_end, length, identities, mismatches, positives, gap_openings, gaps, transcript_len, status;
};

struct Emu16Item {
	const int8_t* q; int32_t qlen; const int8_t* cbs; const int8_t* t; int32_t tlen; int32_t d_begin, d_end;
};

// Trace16Group::put<R> with a run-time R (the kernel unrolls the group and knows R at compile time)
template<int P, int R = 0>
static void put_r(Trace16Group<P>& acc, int r, const pk16* tb0, const pk16* tb1)
{
	if constexpr (R < Sw16Group<P>::G) {
		if (r == R) acc.template put<R>(tb0, tb1);
		else put_r<P, R + 1>(acc, r, tb0, tb1);
	}
}

template<int P, bool TRACE, bool COORDS = true>
static void run16(const Emu16Item& A, const Emu16Item& B, const int8_t* M, int gap_open, int gap_extend, Emu16Out* outA, Emu16Out* outB,
	uint8_t* trA, uint8_t* trB, int cap)
{
	uint16_t table[1024];
	for (int x = 0; x < 1024; ++x) table[x] = sw16_table_entry(M, x);
	const Geom gA = make_geom(A.qlen, A.tlen, A.d_begin, A.d_end), gB = make_geom(B.qlen, B.tlen, B.d_begin, B.d_end);
	const SeqView vA{ A.q, A.t, A.cbs, M }, vB{ B.q, B.t, B.cbs, M };
	constexpr int G = Sw16Group<P>::G, L = (P == 3 || P == 5) ? 16 : 64;      // lanes of an item pair
	const int nA = sw16_pairs(gA), nB = sw16_pairs(gB), T = ((nA > nB ? nA : nB) + G - 1) / G * G;      // whole groups, as the kernel
	const pk16 go = pk_both(gap_open + gap_extend), ge = pk_both(gap_extend);
	std::vector<Lane16<P>> st(L);
	for (int l = 0; l < L; ++l) lane16_init(st[l], gA, vA, gB, vB, l);
	std::vector<uint8_t> traceA, traceB;
	if (TRACE) { traceA.assign((size_t)trace_bytes(gA, P) + 8, 0xee); traceB.assign((size_t)trace_bytes(gB, P) + 8, 0xee); }
	pk16 S0[L][P], S1[L][P], tb[L][P], tbe[L][P], nb[L];
	std::vector<Trace16Group<P>> acc(L);
	for (int t = 0; t < T; ++t) {
		const uint32_t revt = 0xffffu - (uint32_t)t;
		for (int l = 0; l < L; ++l) lane16_scores(st[l], table, S0[l], S1[l]);
		for (int par = 0; par < 2; ++par) {
			if (par == 0) for (int l = 0; l < L; ++l) nb[l] = l == 0 ? 0 : st[l - 1].F[2 * P - 1];
			else for (int l = 0; l < L; ++l) nb[l] = l == L - 1 ? 0 : st[l + 1].E[0];
			for (int l = 0; l < L; ++l) {
				if (par == 0) lane16_step<P, TRACE, 0, COORDS>(st[l], S0[l], nb[l], go, ge, revt, tb[l]);
				else lane16_step<P, TRACE, 1, COORDS>(st[l], S1[l], nb[l], go, ge, revt, tb[l]);
			}
			if (TRACE && par == 0)
				for (int l = 0; l < L; ++l) for (int p = 0; p < P; ++p) tbe[l][p] = tb[l][p];
		}
		if (TRACE) {
			// the kernel's register accumulation (Trace16Group::put<R>, R = position inside the group) and its one store per group
			for (int l = 0; l < L; ++l) put_r<P>(acc[l], t % G, tbe[l], tb[l]);
			if (t % G == G - 1) {
				const int g0 = t - (G - 1);
				for (int l = 0; l < L; ++l) {
					if (g0 < nA) memcpy(traceA.data() + trace_byte_index(P, g0, l * P), acc[l].a, 16);
					if (g0 < nB) memcpy(traceB.data() + trace_byte_index(P, g0, l * P), acc[l].b, 16);
				}
			}
		}
		// systolic letter flow
		const Edge16 e = sw16_edge(gA, vA, gB, vB, P, t, L - 1);
		pk16 nqq[L], ncc[L], ntt[L];
		for (int l = 0; l < L; ++l) {
			nqq[l] = l < L - 1 ? st[l + 1].QQ[1] : e.qq;
			ncc[l] = l < L - 1 ? st[l + 1].CC[1] : e.cc;
			ntt[l] = l > 0 ? st[l - 1].TT[P - 1] : e.tt;
		}
		for (int l = 0; l < L; ++l) lane16_advance(st[l], nqq[l], ncc[l], ntt[l]);
	}
	for (int item = 0; item < 2; ++item) {
		const Geom& g = item ? gB : gA;
		const SeqView& v = item ? vB : vA;
		Emu16Out* out = item ? outB : outA;
		int bs = 0, bi = 0, bj = 0x7fffffff;
		for (int l = 0; l < L; ++l) {
			int s, i, j;
			lane16_finish<P, COORDS>(st[l], g, item == 1, l, s, i, j);
			if (better_end(s, j, i, bs, bj, bi)) { bs = s; bi = i; bj = j; }
		}
		memset(out, 0, sizeof(*out));
		out->score = bs;
		if (bs > 0 && COORDS) { out->q_end = bi + 1; out->s_end = bj + 1; }
		if (TRACE && bs > 0 && bs < SW16_MAX_SCORE) {
			const WalkResult r = traceback_walk((item ? traceB : traceA).data(), g, P, v, gap_open, gap_extend, bs, bi, bj, item ? trB : trA, cap);
			out->q_begin = r.q_begin; out->s_begin = r.s_begin; out->length = r.length; out->identities = r.identities;
			out->mismatches = r.mismatches; out->positives = r.positives; out->gap_openings = r.gap_openings; out->gaps = r.gaps;
			out->transcript_len = r.transcript_len; out->status = r.status;
		}
	}
}

// trace = 1: traceback mode (coordinates, statistics and transcripts of both items), 0: end cells, 2: scores only; force_p: 0 = the pair's class; 3 / 5: the row
// class, if both bands fit its 96 / 160 diagonals
extern "C" int emu_banded_swipe16(const Emu16Item* A, const Emu16Item* B, const int8_t* M, int gap_open, int gap_extend, int trace, int force_p,
	Emu16Out* outA, Emu16Out* outB, uint8_t* trA, uint8_t* trB, int cap)
{
	int P = 1;
	while (128 * P < A->d_end - A->d_begin || 128 * P < B->d_end - B->d_begin) P *= 2;
	if (force_p == 3 || force_p == 5) {
		if (32 * force_p < A->d_end - A->d_begin || 32 * force_p < B->d_end - B->d_begin) return -4;
		P = force_p;
	}
	else if (force_p > P) P = force_p;
	switch (P) {
	case 3: if (trace == 1) run16<3, true>(*A, *B, M, gap_open, gap_extend, outA, outB, trA, trB, cap); else if (trace == 2) run16<3, false, false>(*A, *B, M, gap_open, gap_extend, outA, outB, trA, trB, cap); else run16<3, false>(*A, *B, M, gap_open, gap_extend, outA, outB, trA, trB, cap); break;
	case 5: if (trace == 1) run16<5, true>(*A, *B, M, gap_open, gap_extend, outA, outB, trA, trB, cap); else if (trace == 2) run16<5, false, false>(*A, *B, M, gap_open, gap_extend, outA, outB, trA, trB, cap); else run16<5, false>(*A, *B, M, gap_open, gap_extend, outA, outB, trA, trB, cap); break;
	case 1: if (trace == 1) run16<1, true>(*A, *B, M, gap_open, gap_extend, outA, outB, trA, trB, cap); else if (trace == 2) run16<1, false, false>(*A, *B, M, gap_open, gap_extend, outA, outB, trA, trB, cap); else run16<1, false>(*A, *B, M, gap_open, gap_extend, outA, outB, trA, trB, cap); break;
	case 2: if (trace == 1) run16<2, true>(*A, *B, M, gap_open, gap_extend, outA, outB, trA, trB, cap); else if (trace == 2) run16<2, false, false>(*A, *B, M, gap_open, gap_extend, outA, outB, trA, trB, cap); else run16<2, false>(*A, *B, M, gap_open, gap_extend, outA, outB, trA, trB, cap); break;
	case 4: if (trace == 1) run16<4, true>(*A, *B, M, gap_open, gap_extend, outA, outB, trA, trB, cap); else if (trace == 2) run16<4, false, false>(*A, *B, M, gap_open, gap_extend, outA, outB, trA, trB, cap); else run16<4, false>(*A, *B, M, gap_open, gap_extend, outA, outB, trA, trB, cap); break;
	default: return -4;
	}
	return 0;
}

// ---- the launch classes and the trace layout as plain numbers (tests/test_wavefront16_emu.py checks them without a sweep) ----
extern "C" int emu_band_class(int band, int rows) { return rows ? band_class_rows(band) : band_class(band); }
extern "C" int emu_class_index(int P) { return class_index(P); }
extern "C" int emu_class_of_index(int c) { return class_of_index(c); }
extern "C" int emu_class_lanes(int P) { return class_lanes(P); }
extern "C" int emu_items_per_wave16(int P) { return class_items_per_wave16(P); }
extern "C" long long emu_trace_bytes(int qlen, int tlen, int d_begin, int d_end, int P) { return (long long)trace_bytes(make_geom(qlen, tlen, d_begin, d_end), P); }
extern "C" long long emu_trace_byte_index(int P, int t, int x) { return (long long)trace_byte_index(P, t, x); }
extern "C" int emu_trace_pairs(int qlen, int tlen, int d_begin, int d_end) { return (int)trace_pairs(make_geom(qlen, tlen, d_begin, d_end)); }
