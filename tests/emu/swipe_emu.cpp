// tests/emu/swipe_emu.cpp -- TEST INFRASTRUCTURE ONLY.
// CPU lane-emulator of the wavefront schedule in diamond_amd/csrc/swipe_kernels.hip: runs the SAME
// per-lane code (diamond_amd/csrc/swipe_core.h) for 64 emulated lanes in lock-step, with the DPP
// wave shifts replaced by array indexing. Lets the CPU test-suite check the anti-diagonal index
// arithmetic, tie-breaking and the traceback walk against oracle/ without a GPU. Never used by the
// product (the product path is the HIP library and fails loudly without a GPU).
#include <vector>
#include <cstring>
#include "../../diamond_amd/csrc/swipe_core.h"

using namespace dmnd;

struct EmuOut {
	int32_t score, q_begin, q_end, s_begin, s_end, length, identities, mismatches, positives, gap_openings, gaps, transcript_len, status;
};

template<int P, bool COORDS, bool TRACE>
static void run(const SeqView& v, int qlen, int tlen, int d_begin, int d_end, int gap_open, int gap_extend, EmuOut* out, uint8_t* transcript, int cap)
{
	const Geom g = make_geom(qlen, tlen, d_begin, d_end);
	const int go = gap_open + gap_extend, ge = gap_extend;
	// the register-window lanes of the score / coordinates / traceback kernels (WinLane, swipe_core.h), driven exactly as
	// banded_swipe_kernel drives them: fetch the pair's new letters, even step, odd step, advance the windows
	std::vector<WinLane<P, COORDS>> st(64);
	for (int l = 0; l < 64; ++l) win_init(st[l], g, v, l);
	std::vector<uint8_t> trace;
	if (TRACE) trace.assign((size_t)trace_bytes(g, P) + 8, 0xee);
	int nb[64], nq[64], nc[64], nt[64];
	std::vector<uint32_t> tbe(64 * ((P + 3) / 4)), tbo(64 * ((P + 3) / 4));
	constexpr int D = (P + 3) / 4;
	int t = 0;
	for (int a = g.a_first; a <= g.a_last; a += 2, ++t) {
		for (int l = 0; l < 64; ++l) {
			const int xi = clampi(st[l].iq, g.qlen - 1), xj = clampi(st[l].jt, g.tlen - 1);
			nq[l] = v.q[xi]; nt[l] = v.t[xj]; nc[l] = v.cbs ? v.cbs[xi] : 0;
		}
		for (int l = 0; l < 64; ++l) nb[l] = l == 0 ? 0 : st[l - 1].F[2 * P - 1];       // wave_shr:1, lane 0 reads 0
		for (int l = 0; l < 64; ++l) win_step<P, COORDS, TRACE, 0>(st[l], v.M, nb[l], go, ge, a, g.d_begin + 2 * P * l, &tbe[l * D]);
		for (int l = 0; l < 64; ++l) nb[l] = l == 63 ? 0 : st[l + 1].E[0];               // wave_shl:1, lane 63 reads 0
		for (int l = 0; l < 64; ++l) win_step<P, COORDS, TRACE, 1>(st[l], v.M, nb[l], go, ge, a + 1, g.d_begin + 2 * P * l, &tbo[l * D]);
		if (TRACE) for (int l = 0; l < 64; ++l) win_store_trace<P>(trace.data() + trace_byte_index(P, t, l * P), &tbe[l * D], &tbo[l * D]);
		for (int l = 0; l < 64; ++l) win_advance(st[l], nq[l], nc[l], nt[l]);
	}
	for (int l = 0; l < 64; ++l) win_finish(st[l], g.d_begin + 2 * P * l);
	int bs = 0, bi = 0, bj = 0x7fffffff;
	for (int l = 0; l < 64; ++l) {
		if (COORDS ? better_end(st[l].best, st[l].best_j, st[l].best_i, bs, bj, bi) : st[l].best > bs) {
			bs = st[l].best; bi = st[l].best_i; bj = st[l].best_j;
		}
	}
	memset(out, 0, sizeof(*out));
	out->score = bs;
	if (COORDS && bs > 0) { out->q_end = bi + 1; out->s_end = bj + 1; }
	if (TRACE && bs > 0) {
		const WalkResult r = traceback_walk(trace.data(), g, P, v, gap_open, gap_extend, bs, bi, bj, transcript, cap);
		out->q_begin = r.q_begin; out->s_begin = r.s_begin; out->length = r.length; out->identities = r.identities;
		out->mismatches = r.mismatches; out->positives = r.positives; out->gap_openings = r.gap_openings; out->gaps = r.gaps;
		out->transcript_len = r.transcript_len; out->status = r.status;
	}
}

// one statistics pass (STAT_FWD forward, STAT_BWD on the reversed views); returns score, end cell, (a, b)
template<int P, int STAT>
static void run_stats(const SeqView& v, int qlen, int tlen, int d_begin, int d_end, int gap_open, int gap_extend, int* res)
{
	const Geom g = make_geom(qlen, tlen, d_begin, d_end);
	const int go = gap_open + gap_extend, ge = gap_extend;
	std::vector<Lane<P, true, STAT>> st(64);
	for (int l = 0; l < 64; ++l) st[l].init(g, l);
	int nb[64], na[64], nbb[64];
	for (int a = g.a_first; a <= g.a_last; a += 2) {
		for (int l = 0; l < 64; ++l) {
			nb[l] = l == 0 ? 0 : st[l - 1].F[2 * P - 1];
			na[l] = l == 0 ? 0 : st[l - 1].st.Fa[2 * P - 1];
			nbb[l] = l == 0 ? 0 : st[l - 1].st.Fb[2 * P - 1];
		}
		for (int l = 0; l < 64; ++l) lane_step<P, true, false, 0, STAT>(st[l], g, v, l, a, nb[l], go, ge, nullptr, na[l], nbb[l]);
		for (int l = 0; l < 64; ++l) {
			nb[l] = l == 63 ? 0 : st[l + 1].E[0];
			na[l] = l == 63 ? 0 : st[l + 1].st.Ea[0];
			nbb[l] = l == 63 ? 0 : st[l + 1].st.Eb[0];
		}
		for (int l = 0; l < 64; ++l) lane_step<P, true, false, 1, STAT>(st[l], g, v, l, a + 1, nb[l], go, ge, nullptr, na[l], nbb[l]);
	}
	int bs = 0, bi = 0, bj = 0x7fffffff, ba = 0, bb = 0;
	for (int l = 0; l < 64; ++l)
		if (better_end(st[l].best, st[l].best_j, st[l].best_i, bs, bj, bi)) {
			bs = st[l].best; bi = st[l].best_i; bj = st[l].best_j; ba = st[l].best_a; bb = st[l].best_b;
		}
	res[0] = bs; res[1] = bi; res[2] = bj; res[3] = ba; res[4] = bb;
}

template<int P>
static void stats_p(const int8_t* q, int qlen, const int8_t* cbs, const int8_t* t, int tlen, int d_begin, int d_end,
	const int8_t* M, int gap_open, int gap_extend, EmuOut* out)
{
	int f[5], b[5];
	SeqView v{ q, t, cbs, M };
	run_stats<P, STAT_FWD>(v, qlen, tlen, d_begin, d_end, gap_open, gap_extend, f);
	memset(out, 0, sizeof(*out));
	out->score = f[0];
	if (f[0] <= 0) return;
	out->q_end = f[1] + 1; out->s_end = f[2] + 1; out->identities = f[3]; out->length = f[4];
	int rt, rd0, rd1;
	reversed_band(qlen, out->s_end, d_begin, d_end, rt, rd0, rd1);
	SeqView r{ q, t, cbs, M };
	r.rev_q = qlen - 1; r.rev_t = rt - 1;
	run_stats<P, STAT_BWD>(r, qlen, rt, rd0, rd1, gap_open, gap_extend, b);
	out->score = b[0];
	out->q_begin = qlen - (b[1] + 1); out->s_begin = rt - (b[2] + 1);
	out->mismatches = b[3]; out->gap_openings = b[4];
	out->gaps = out->length - out->identities - out->mismatches;
}

extern "C" int emu_swipe_stats(const int8_t* q, int qlen, const int8_t* cbs, const int8_t* t, int tlen, int d_begin, int d_end,
	const int8_t* M, int gap_open, int gap_extend, EmuOut* out)
{
	int P = 1;
	while (128 * P < d_end - d_begin) P *= 2;
	switch (P) {
	case 1: stats_p<1>(q, qlen, cbs, t, tlen, d_begin, d_end, M, gap_open, gap_extend, out); break;
	case 2: stats_p<2>(q, qlen, cbs, t, tlen, d_begin, d_end, M, gap_open, gap_extend, out); break;
	case 4: stats_p<4>(q, qlen, cbs, t, tlen, d_begin, d_end, M, gap_open, gap_extend, out); break;
	default: return -4;
	}
	return 0;
}

template<int P>
static void run_mode(int mode, const SeqView& v, int qlen, int tlen, int d_begin, int d_end, int go, int ge, EmuOut* out, uint8_t* tr, int cap)
{
	if (mode == 0) run<P, false, false>(v, qlen, tlen, d_begin, d_end, go, ge, out, tr, cap);
	else if (mode == 1) run<P, true, false>(v, qlen, tlen, d_begin, d_end, go, ge, out, tr, cap);
	else run<P, true, true>(v, qlen, tlen, d_begin, d_end, go, ge, out, tr, cap);
}

extern "C" int emu_banded_swipe(const int8_t* q, int qlen, const int8_t* cbs, const int8_t* t, int tlen, int d_begin, int d_end,
	const int8_t* M, int gap_open, int gap_extend, int mode, int force_p, EmuOut* out, uint8_t* transcript, int cap)
{
	const int band = d_end - d_begin;
	int P = 1;
	while (128 * P < band) P *= 2;
	if (force_p > P) P = force_p;
	const SeqView v{ q, t, cbs, M };
	switch (P) {
	case 1: run_mode<1>(mode, v, qlen, tlen, d_begin, d_end, gap_open, gap_extend, out, transcript, cap); break;
	case 2: run_mode<2>(mode, v, qlen, tlen, d_begin, d_end, gap_open, gap_extend, out, transcript, cap); break;
	case 4: run_mode<4>(mode, v, qlen, tlen, d_begin, d_end, gap_open, gap_extend, out, transcript, cap); break;
	case 8: run_mode<8>(mode, v, qlen, tlen, d_begin, d_end, gap_open, gap_extend, out, transcript, cap); break;
	default: return -4;
	}
	return 0;
}
