// tests/emu/chain_emu.cpp -- TEST INFRASTRUCTURE ONLY.
// Runs the product's chaining (diamond_amd/csrc/chain_graph.h) and the retired round-1 restatement of the reference
// (oracle/chain_ref.h) on the same seed hits of one (query, target) pair and hands both results to the test.
#include <cstring>
#include <vector>
#include "../../diamond_amd/csrc/chain_graph.h"
#include "../../oracle/chain_ref.h"

namespace {

template<typename WsT, typename TableT, typename SeqT, typename SegT, typename ChainT>
bool run_chains(WsT& ws, const TableT& S, const SeqT& qs, const SeqT& ts, const std::vector<SegT>& segs, std::vector<ChainT>& chains)
{
	ws.run(S, qs, ts, segs, chains);
	return true;
}

// the device planner's instance of the chaining (fixed-capacity arrays, insertion sort), run on the host: false = a target that does
// not fit its arrays (the device hands those to the host)
typedef dmnd::ChainWorkspaceT<dmnd::FixedChainPolicy> FixedWs;
bool run_chains(FixedWs& ws, const dmnd::ScoreTable& S, const dmnd::SeqRef& qs, const dmnd::SeqRef& ts, const std::vector<dmnd::Seg>& segs, std::vector<dmnd::Chain>& chains)
{
	if (segs.size() > 16) return false;
	dmnd::FixedVec<dmnd::Chain, 16> out;
	ws.run_segs(S, qs, ts, segs.data(), segs.size(), out);
	if (ws.overflowed() || out.overflow) return false;
	chains.assign(out.begin(), out.end());
	return true;
}

template<typename SegT, typename SeqT, typename TableT, typename WsT, typename ChainT, typename XdropF>
int run_one(const int8_t* q, int qlen, const int8_t* cbs, const int8_t* t, int tlen, const int8_t* matrix8, int gap_open, int gap_extend,
	const int* hi, const int* hj, int n_hits, int* seg_out, int seg_cap, int* n_segs, int* chain_out, int chain_cap, XdropF xdrop)
{
	TableT S;
	for (int i = 0; i < 1024; ++i) S.m[i] = matrix8[i];
	S.gap_open = gap_open; S.gap_extend = gap_extend;
	const SeqT qs{ q, qlen }, ts{ t, tlen };
	WsT ws;
	std::vector<SegT> segs;
	// ungapped_stage: hits sorted by (diagonal, j); a hit inside the previous segment of its diagonal is skipped
	for (int x = 0; x < n_hits; ++x) {
		if (!segs.empty() && segs.back().diag() == hi[x] - hj[x] && segs.back().j_end() >= hj[x]) continue;
		const SegT d = xdrop(S, qs, cbs, ts, hi[x], hj[x], 20);      // config.raw_ungapped_xdrop = rawscore(12.3 bits) for BLOSUM62 11/1, config.cpp:428,853
		if (d.score > 0) segs.push_back(d);
	}
	*n_segs = (int)segs.size();
	for (int x = 0; x < (int)segs.size() && x < seg_cap; ++x) { seg_out[4 * x] = segs[x].i; seg_out[4 * x + 1] = segs[x].j; seg_out[4 * x + 2] = segs[x].len; seg_out[4 * x + 3] = segs[x].score; }
	if (segs.empty()) return 0;
	std::vector<ChainT> chains;
	if (!run_chains(ws, S, qs, ts, segs, chains)) return -1;
	for (int x = 0; x < (int)chains.size() && x < chain_cap; ++x) {
		const ChainT& c = chains[x];
		const int v[7] = { c.d_min, c.d_max, c.score, c.q0, c.q1, c.s0, c.s1 };
		std::memcpy(chain_out + 7 * x, v, sizeof v);
	}
	return (int)chains.size();
}

}

// q / t point INTO padded buffers (delimiter 31 before and after, as in a sequence block). Returns the number of chains.
extern "C" int emu_chain(int which, const int8_t* q, int qlen, const int8_t* cbs, const int8_t* t, int tlen, const int8_t* matrix8, int gap_open, int gap_extend,
	const int* hi, const int* hj, int n_hits, int* seg_out, int seg_cap, int* n_segs, int* chain_out, int chain_cap)
{
	if (which == 0)
		return run_one<dmnd::Seg, dmnd::SeqRef, dmnd::ScoreTable, dmnd::ChainWorkspace, dmnd::Chain>(q, qlen, cbs, t, tlen, matrix8, gap_open, gap_extend, hi, hj, n_hits,
			seg_out, seg_cap, n_segs, chain_out, chain_cap,
			[](const dmnd::ScoreTable& S, const dmnd::SeqRef& a, const int8_t* c, const dmnd::SeqRef& b, int i, int j, int x) { return dmnd::xdrop_ungapped(S, a, c, b, i, j, x); });
	if (which == 2)
		return run_one<dmnd::Seg, dmnd::SeqRef, dmnd::ScoreTable, FixedWs, dmnd::Chain>(q, qlen, cbs, t, tlen, matrix8, gap_open, gap_extend, hi, hj, n_hits,
			seg_out, seg_cap, n_segs, chain_out, chain_cap,
			[](const dmnd::ScoreTable& S, const dmnd::SeqRef& a, const int8_t* c, const dmnd::SeqRef& b, int i, int j, int x) { return dmnd::xdrop_ungapped(S, a, c, b, i, j, x); });
	return run_one<dmnd_ref::Seg, dmnd_ref::SeqRef, dmnd_ref::ScoreTable, dmnd_ref::ChainWorkspace, dmnd_ref::Chain>(q, qlen, cbs, t, tlen, matrix8, gap_open, gap_extend, hi, hj, n_hits,
		seg_out, seg_cap, n_segs, chain_out, chain_cap,
		[](const dmnd_ref::ScoreTable& S, const dmnd_ref::SeqRef& a, const int8_t* c, const dmnd_ref::SeqRef& b, int i, int j, int x) { return dmnd_ref::xdrop_ungapped(S, a, c, b, i, j, x); });
}
