// chain_graph.h -- host side of the extension stage that fixes the band geometry of the GPU Smith-Waterman (SURVEY.md 8 rows
// a12-a14): x-drop ungapped extension of the seed hits and chaining of the resulting diagonal segments into approximate HSPs.
//
// The RESULTS are pinned to the reference (they decide [d_begin, d_end) of every DpTarget, hence scores and CIGARs):
//   xdrop_ungapped            /root/reference/src/dp/ungapped_align.cpp:151-199
//   DiagGraph / Aligner       src/chaining/diag_graph.h, greedy_align.cpp:49-413, backtrace.cpp:36-357
//   merge_hsps, Chaining::run greedy_align.cpp:417-497
// The FORMULATION is this project's own:
//   * the segment graph keeps, per segment, a singly linked list of its incoming links in a pool (newest first -- the order in
//     which the best link is searched), instead of one edge array with positional inserts that renumber later segments;
//   * the sweep keeps its "latest live segment per diagonal" frontier as a flat vector ordered by diagonal (a few entries;
//     binary search + in-place erase) instead of a node-based map;
//   * the junction between two segments on different diagonals is found from the re-cut identities derived in junction():
//     with T(y) = score of the upstream segment re-cut to end before column y and R(y) = score of the downstream segment
//     re-cut to begin at column y, the junction after column c scores T(c + 1) + R(c + g + 1) up to a constant, and both
//     the search and the reported partial scores are expressed through T and R (the vertical case is the same code on
//     swapped coordinates);
//   * the chain walk is iterative: one descent that records (segment, link, running minimum) frames, then an unwind that
//     finds the frame at which the chain really begins; the reference recurses.
// tests/test_chain_graph.py checks it against the previous restatement (kept under oracle/ as a checker) on random segment
// sets, and tests/test_extend_plan.py against the reference's own DpTargets.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <vector>

#include "xdrop_core.h"

namespace dmnd {

struct ScoreTable {                // ScoreMatrix::operator()(a,b) = matrix32[a*32+b] on masked letters
	int m[32 * 32];
	int gap_open, gap_extend;
	int at(int a, int b) const { return m[(a << 5) + b]; }
};

struct SeqRef {                    // Sequence: letters are read through & 31 (basic/sequence.h:80-87)
	const int8_t* p;
	int len;
	int operator[](int i) const { return p[i] & 31; }
};

struct Seg {                       // one ungapped diagonal segment: cells (i + k, j + k), k < len
	int i, j, len, score;
	int diag() const { return i - j; }
	int j_end() const { return j + len; }
	int j_last() const { return j + len - 1; }
	int i_end() const { return i + len; }
	int i_last() const { return i + len - 1; }
};

struct Chain {                     // the fields of an approximate HSP the extension reads
	int d_min, d_max, score;
	int q0, q1, s0, s1;            // query / subject range, end exclusive
};

struct HostSeedHit { int i, j, score, frame; int src = -1; };      // src: index of the hit in the block pair's hit list (device x-drop results)

// chaining constants = the reference's config defaults (basic/config.cpp:549-603)
struct ChainCfg {
	int max_shift = 2000;                 // chaining_maxgap
	size_t range_cover = 8;               // chaining_range_cover
	size_t maxnodes = 0;                  // chaining_maxnodes
	double len_cap = 2.0;                 // chaining_len_cap
	size_t min_nodes = 200;               // chaining_min_nodes
	double stacked_hsp_ratio = 0.5;       // chaining_stacked_hsp_ratio
	int cutoff = 19;                      // minimum chain score
	double space_penalty = 0.1;
	int link_padding = 10, reverse_overhang = 10;
};

// ---- x-drop ungapped extension ------------------------------------------------------------------------------------
// One direction of the extension: from (qi, tj) in steps of `dir`, adding letter scores (+ bias) to a running sum that starts
// at `best`; stops at a delimiter or when the sum has dropped xdrop below the best. Returns the best sum and, in `reach`,
// how many letters it took to get there. Reads one position beyond either end: the blocks carry delimiters there.
inline int xdrop_walk(const ScoreTable& S, const SeqRef& q, const int8_t* cbs, const SeqRef& t, int qi, int tj, int dir, int best, int xdrop, int& reach)
{
	return xdrop_walk_core(S.m, q.p + qi, cbs ? cbs + qi : nullptr, t.p + tj, dir, best, xdrop, reach);
}

inline Seg xdrop_ungapped(const ScoreTable& S, const SeqRef& q, const int8_t* cbs, const SeqRef& t, int qa, int sa, int xdrop)
{
	int left, right;
	const int s_left = xdrop_walk(S, q, cbs, t, qa - 1, sa - 1, -1, 0, xdrop, left);
	const int s_both = xdrop_walk(S, q, cbs, t, qa, sa, +1, s_left, xdrop, right);
	return Seg{ qa - left, sa - left, left + right, s_both };
}

// ---- chaining -----------------------------------------------------------------------------------------------------
struct ChainWorkspace {
	struct Node : Seg {
		int best, peak, dip;          // best chain score ending here; the highest and lowest running score on that chain's way
		int newest;                   // most recent incoming link (index into links), -1 = none
		int rel() const { return best == peak ? best : best - dip; }
	};
	struct InLink {                   // an admissible way into `to` from `from`
		int best, peak, dip, begin;   // chain score through it (and its peak / dip); score at which the downstream part begins
		int j;                        // first subject column of the downstream part
		unsigned from;
		int older;                    // next link of the same segment, older first
	};
	struct Junction { int total, s1, q1, s2, q2, up, down; };      // up / down: re-cut scores of the two parts
	struct Front { int diag; unsigned node; };
	struct Frame { unsigned node; int link, dip; };

	std::vector<Node> nodes;
	std::vector<InLink> links;
	std::vector<Front> front;
	std::vector<Frame> frames;
	std::vector<unsigned> tops;
	const ScoreTable* S = nullptr;
	SeqRef query, subject;
	ChainCfg cfg;

	static Node node_of(const Seg& s) { Node n; (Seg&)n = s; n.best = n.peak = n.dip = s.score; n.newest = -1; return n; }

	// Segments arrive sorted by (diagonal, j): a segment that begins inside the stretch already covered on its diagonal is dropped
	void load(const std::vector<Seg>& segs)
	{
		nodes.clear(); links.clear();
		bool have = false;
		int d = 0, covered = 0;
		for (const Seg& s : segs) {
			if (have && s.diag() == d && s.j <= covered) continue;
			covered = have && s.diag() == d ? std::max(covered, s.j_end()) : s.j_end();
			d = s.diag(); have = true;
			nodes.push_back(node_of(s));
		}
	}

	// Drops a segment when more than range_cover still-open segments (in subject order) score at least as much and cover its
	// whole subject range. The survivors come out in the order in which they stop being open.
	void thin_out()
	{
		std::vector<Node> closed, open;
		closed.reserve(nodes.size());
		for (const Node& d : nodes) {
			size_t covering = 0, keep = 0;
			for (size_t x = 0; x < open.size(); ++x) {
				const Node& w = open[x];
				if (w.j_end() <= d.j) { closed.push_back(w); continue; }
				covering += w.score >= d.score && w.j <= d.j && w.j_end() >= d.j_end();
				if (keep != x) open[keep] = w;
				++keep;
			}
			open.resize(keep);
			if (covering <= cfg.range_cover) open.push_back(d);
		}
		closed.insert(closed.end(), open.begin(), open.end());
		nodes.swap(closed);
	}

	// Best incoming link of `node` among those whose downstream part begins before column j and that beat the bare
	// segment; of equally good ones the newest. -1 = none.
	int best_in(unsigned node, int j) const
	{
		int pick = -1, top = nodes[node].score;
		for (int l = nodes[node].newest; l >= 0; l = links[(size_t)l].older)
			if (links[(size_t)l].j < j && links[(size_t)l].best > top) { pick = l; top = links[(size_t)l].best; }
		return pick;
	}

	void add_link(unsigned to, const InLink& l)
	{
		Node& d = nodes[to];
		if (l.best > d.best) { d.best = l.best; d.peak = l.peak; d.dip = l.dip; }
		links.push_back(l);
		links.back().older = d.newest;
		d.newest = (int)links.size() - 1;
	}

	// Where to leave segment `up` (the one further left / higher) for segment `down` on a diagonal g >= 0 lower, in
	// coordinates (x = position along the sequence that both segments advance in, off = the other coordinate minus x):
	//   W(off, x)  score of the cell at x on diagonal `off`
	//   T(y)       score of `up` re-cut to end before x = y       = up.score + sum W(up.off, [up.end, y)) - sum W(up.off, [y, up.end))
	//   R(y)       score of `down` re-cut to begin at x = y       = down.score + sum W(down.off, [y, down.x)) - sum W(down.off, [down.x, y))
	// Leaving `up` after x = c means entering `down` at x = c + g + 1. Candidates c run from c0 (the last cell of `up` if the
	// segments do not reach each other, else `pad` cells before the first possible junction) while the entry point stays within
	// `pad` cells behind the first cell either segment offers, and inside `down`. The first candidate with the largest
	// T(c + 1) + R(c + g + 1) wins; since T(c + 1) = T(c0 + 1) + sum W(up.off, (c0, c]), the search runs on that partial sum.
	struct Axis { int x, off, len, score; int end() const { return x + len; } int last() const { return x + len - 1; } };

	template<typename Cell>
	static bool junction_on(const Axis& up, const Axis& down, int pad, Cell W, int& c_best, int& t_best, int& r_best, int& total)
	{
		const int g = up.off - down.off;
		const bool apart = up.last() < down.x - g - 1;
		const int c0 = apart ? up.last() : std::max(down.x - g - 1 - pad, up.x);
		const int entry_max = std::min(std::max(down.x, up.last() + g + 1 + pad), down.last());
		if (c0 + g + 1 > down.last()) return false;
		auto span = [&](int off, int a, int b) { int s = 0; for (int x = a; x < b; ++x) s += W(off, x); return s; };
		int run = 0;                                                                              // sum W(up.off, (c0, c])
		int r = down.score + span(down.off, c0 + g + 1, down.x) - span(down.off, down.x, c0 + g + 1);      // R(c + g + 1)
		int run_best = 0;
		total = INT_MIN;
		for (int c = c0;; ++c) {
			if (run + r > total) { total = run + r; c_best = c; r_best = r; run_best = run; }
			if (c + g + 1 >= entry_max) break;
			r -= W(down.off, c + g + 1);
			run += W(up.off, c + 1);
		}
		// T(c_best + 1)
		t_best = up.score + span(up.off, up.end(), c_best + 1) - span(up.off, c_best + 1, up.end());
		(void)run_best;
		return true;
	}

	// Junction from segment e into segment d (e is the upstream one). Lower or equal diagonal of d: the segments advance
	// along the subject; higher diagonal: along the query (same computation with the roles of the sequences swapped).
	bool junction(const Seg& e, const Seg& d, Junction& out) const
	{
		int c = 0, t = 0, r = 0, total = 0;
		if (e.diag() < d.diag()) {
			const Axis up{ e.i, -e.diag(), e.len, e.score }, down{ d.i, -d.diag(), d.len, d.score };
			auto cell = [&](int off, int x) { return S->at(subject[x + off], query[x]); };
			if (!junction_on(up, down, cfg.link_padding, cell, c, t, r, total)) return false;
			const int g = up.off - down.off;
			out = Junction{ total, c + up.off, c, c + g + 1 + down.off, c + g + 1, t, r };
		}
		else {
			const Axis up{ e.j, e.diag(), e.len, e.score }, down{ d.j, d.diag(), d.len, d.score };
			auto cell = [&](int off, int x) { return S->at(query[x + off], subject[x]); };
			if (!junction_on(up, down, cfg.link_padding, cell, c, t, r, total)) return false;
			const int g = up.off - down.off;
			out = Junction{ total, c, c + up.off, c + g + 1, c + g + 1 + down.off, t, r };
		}
		return true;
	}

	// Considers continuing the best chain of `from` with segment `to`, and records the link if it improves on the bare segment.
	void try_link(unsigned to, unsigned from)
	{
		const Node& d = nodes[to];
		const Node& e = nodes[from];
		const int shift = d.diag() - e.diag();
		const int gap = shift ? -(S->gap_open + std::abs(shift) * S->gap_extend) : 0;
		const int apart = shift > 0 ? d.j - e.j_last() : d.i - e.i_last();
		InLink l;
		l.from = from; l.older = -1;
		if (apart <= 0 || cfg.space_penalty == 0.0) {
			// the segments overlap in the advancing direction: cut both at the best junction
			const int have = best_in(to, d.j);
			if (have >= 0 && links[(size_t)have].best > e.best + gap + d.score) return;
			Junction jn;
			if (!junction(e, d, jn) || jn.total <= 0) return;
			const int cut = e.score - jn.up;                       // what `from` loses by ending at the junction
			const int in = best_in(from, jn.s1);
			const int e_score = e.score;
			const int e_best = in < 0 ? e_score : std::max(e_score, links[(size_t)in].best);
			int peak = in < 0 ? e_score : std::max(e_score, links[(size_t)in].peak);
			int dip = in < 0 ? e_score : links[(size_t)in].dip;
			l.best = e_best - cut + gap + jn.down;
			const int have2 = best_in(to, jn.s2);
			if (have2 >= 0 && links[(size_t)have2].best > l.best) return;
			l.begin = l.best - jn.down;
			dip = std::min(dip, l.begin);
			if (e_best == peak) peak -= cut;
			l.peak = peak; l.dip = dip; l.j = jn.s2;
		}
		else {
			// free space between the segments: a flat penalty per skipped position
			l.best = e.best + gap - int(cfg.space_penalty * std::max(apart - 1, 0)) + d.score;
			const int have = best_in(to, d.j);
			if (have >= 0 && links[(size_t)have].best > l.best) return;
			l.begin = l.best - d.score;
			l.peak = e.peak;
			l.dip = std::min(e.dip, l.begin);
			l.j = d.j;
		}
		if (l.best <= d.score) return;
		l.peak = std::max(l.peak, l.best);
		if (l.best == l.peak) l.dip = l.best;
		add_link(to, l);
	}

	bool faded(const Node& e, const Node& d) const { return e.best - int(cfg.space_penalty * std::max(d.j - e.j_end(), 0)) <= 0; }
	bool overhangs(const Node& e, const Node& d) const { return e.j_end() - (d.j_end() - std::min(e.diag() - d.diag(), 0)) >= cfg.reverse_overhang; }

	// Segments in (j, i) order; `front` holds for every diagonal the latest segment seen on it while that segment can still
	// contribute (its chain score has not faded over the distance). A new segment is linked to the frontier segments below
	// it (nearest diagonal first), then to those at and above it.
	void sweep()
	{
		front.clear();
		for (unsigned n = 0; n < nodes.size(); ++n) {
			const int dd = nodes[n].diag();
			size_t pos = (size_t)(std::lower_bound(front.begin(), front.end(), dd, [](const Front& f, int v) { return f.diag < v; }) - front.begin());
			const bool fresh = pos == front.size() || front[pos].diag != dd;
			if (fresh) front.insert(front.begin() + (ptrdiff_t)pos, Front{ dd, n });
			int reach_j = 0;
			for (size_t k = pos; k-- > 0;) {
				const unsigned en = front[k].node;
				if (faded(nodes[en], nodes[n])) { front.erase(front.begin() + (ptrdiff_t)k); --pos; continue; }
				if (nodes[en].j_end() < reach_j) continue;
				try_link(n, en);
				reach_j = std::max(reach_j, std::min(nodes[n].j, nodes[en].j_end()));
				if (overhangs(nodes[en], nodes[n])) try_link(en, n);
			}
			int reach_i = 0;
			for (size_t k = fresh ? pos + 1 : pos; k < front.size();) {
				const unsigned en = front[k].node;
				if (k != pos && faded(nodes[en], nodes[n])) { front.erase(front.begin() + (ptrdiff_t)k); continue; }
				if (nodes[en].i_end() >= reach_i) {
					try_link(n, en);
					if (nodes[en].i < nodes[n].i) reach_i = std::max(reach_i, std::min(nodes[en].i_end(), nodes[n].i));
					if (overhangs(nodes[en], nodes[n])) try_link(en, n);
				}
				++k;
			}
			front[pos].node = n;
		}
	}

	static double share(int a0, int a1, int b0, int b1)            // part of [a0, a1) that lies in [b0, b1)
	{
		const int lo = std::max(a0, b0), hi = std::min(a1, b1);
		return (double)(unsigned)(hi > lo ? hi - lo : 0) / (double)(a1 > a0 ? a1 - a0 : 0);
	}

	// May a candidate (ranges + score) coexist with the chains ts[first..)? A chain that the candidate mostly stacks on (in
	// either sequence) without being dwarfed by it does not count; otherwise what remains of the candidate outside that chain
	// must still be worth the cutoff.
	bool compatible(const std::vector<Chain>& ts, size_t first, int q0, int q1, int s0, int s1, int score) const
	{
		for (size_t x = first; x < ts.size(); ++x) {
			const double in_s = share(s0, s1, ts[x].s0, ts[x].s1), in_q = share(q0, q1, ts[x].q0, ts[x].q1);
			if ((1.0 - std::min(in_s, in_q)) * score / ts[x].score >= cfg.stacked_hsp_ratio) continue;
			if ((1.0 - std::max(in_s, in_q)) * score < cfg.cutoff) return false;
		}
		return true;
	}

	// Follows the best links from `top` back to where its chain begins. The chain may not pass through a point whose
	// running score exceeds the chain's final score; a part reached through such a link is cut off if the chain is still worth
	// more from there on. `jump` receives the segment behind a link that crosses more than max_shift diagonals (the walk stops
	// in front of it and the caller continues from there), else UINT_MAX. Returns false if nothing could be followed.
	bool follow(unsigned top, int j_bound, Chain& t, unsigned& jump)
	{
		const int final_score = nodes[top].best;
		frames.clear();
		jump = UINT_MAX;
		unsigned cur = top;
		int dip = final_score;
		bool rejected = false;
		for (;;) {
			const int l = best_in(cur, j_bound);
			if ((l < 0 ? nodes[cur].score : links[(size_t)l].best) > final_score) { rejected = true; break; }
			dip = std::min(dip, l < 0 ? 0 : links[(size_t)l].begin);
			frames.push_back(Frame{ cur, l, dip });
			if (l < 0) break;
			const InLink& in = links[(size_t)l];
			const int shift = nodes[cur].diag() - nodes[in.from].diag();
			if (std::abs(shift) > cfg.max_shift) { jump = in.from; break; }
			j_bound = shift > 0 ? in.j : in.j + shift;
			cur = in.from;
		}
		if (rejected)       // drop the frames that cannot do without what lies behind them
			while (!frames.empty() && links[(size_t)frames.back().link].begin > frames.back().dip) frames.pop_back();
		if (frames.empty()) return false;
		const Node& first = nodes[frames.back().node];
		t.q0 = first.i; t.s0 = first.j; t.score = final_score - frames.back().dip;
		for (const Frame& f : frames) {
			const int dd = nodes[f.node].diag();
			t.d_min = std::min(t.d_min, dd);
			t.d_max = std::max(t.d_max, dd);
		}
		return true;
	}

	void collect(std::vector<Chain>& ts)
	{
		tops.clear();
		for (unsigned n = 0; n < nodes.size(); ++n)
			if (nodes[n].rel() >= cfg.cutoff) tops.push_back(n);
		// same comparison sequence as the reference's sort of its candidate list, so equal scores end up in the same order
		std::sort(tops.begin(), tops.end(), [this](unsigned x, unsigned y) { return nodes[x].rel() > nodes[y].rel(); });
		const size_t none = (size_t)-1;
		size_t first = none;                              // first chain of this call in ts
		for (unsigned n : tops) {
			const Node& d = nodes[n];
			if (!compatible(ts, first == none ? ts.size() : first, d.i, d.i_end(), d.j, d.j_end(), d.score)) continue;
			unsigned from = n;
			int j_bound = subject.len;
			while (from != UINT_MAX) {
				const Node& top = nodes[from];
				Chain t{ INT_MAX, INT_MIN, 0, 0, top.i_end(), 0, top.j_end() };
				unsigned jump;
				follow(from, std::min(top.j_end(), j_bound), t, jump);
				if (t.score > 0) j_bound = t.s0;
				if (t.score >= cfg.cutoff && compatible(ts, first == none ? ts.size() : first, t.q0, t.q1, t.s0, t.s1, t.score)) {
					if (first == none) first = ts.size();
					ts.push_back(t);
				}
				from = jump;
			}
		}
	}

	// Two chains in sequence (a before b in both sequences) join when the sum minus gap costs beats both
	static int joined_score(const Chain& a, const Chain& b)
	{
		const int dq = b.q0 - a.q1, ds = b.s0 - a.s1;
		if (dq < 0 || ds < 0) return 0;
		const int longer = std::max(dq, ds), shorter = std::min(dq, ds);
		// the reference subtracts the two costs in this order (double arithmetic, truncated)
		return int((a.score + b.score) - longer * 0.5 - shorter * 0.1);
	}

	static void join_chains(std::vector<Chain>& h)
	{
		for (size_t a = 0; a < h.size(); ++a)
			for (size_t b = a + 1; b < h.size();) {
				const int limit = std::max(h[a].score, h[b].score);
				const bool ab = joined_score(h[a], h[b]) > limit, ba = !ab && joined_score(h[b], h[a]) > limit;
				if (!ab && !ba) { ++b; continue; }
				const Chain x = ab ? h[a] : h[b], y = ab ? h[b] : h[a];
				h[a] = Chain{ std::min(x.d_min, y.d_min), std::max(x.d_max, y.d_max), joined_score(x, y), x.q0, y.q1, x.s0, y.s1 };
				h.erase(h.begin() + (ptrdiff_t)b);
			}
	}

	// segments sorted by (diagonal, j) in, chains out (unsorted)
	void run(const ScoreTable& st, const SeqRef& q, const SeqRef& s, const std::vector<Seg>& segs, std::vector<Chain>& out)
	{
		out.clear();
		if (segs.size() == 1) {
			const Seg& g = segs[0];
			out.push_back(Chain{ g.diag(), g.diag(), g.score, g.i, g.i_end(), g.j, g.j_end() });
			return;
		}
		S = &st; query = q; subject = s;
		load(segs);
		auto by_score = [](const Seg& x, const Seg& y) { return x.score > y.score; };
		if (cfg.maxnodes > 0) {
			std::sort(nodes.begin(), nodes.end(), by_score);
			if (nodes.size() > cfg.maxnodes) nodes.resize(cfg.maxnodes);
		}
		if (cfg.len_cap > 0.0 && nodes.size() > cfg.min_nodes) {
			// many segments: the best-scoring ones until their lengths add up to len_cap query lengths, at least min_nodes
			std::sort(nodes.begin(), nodes.end(), by_score);
			const double cap = q.len * cfg.len_cap;
			double total = 0.0;
			size_t n = 0;
			while (n < nodes.size() && total < cap) total += nodes[n++].len;
			nodes.resize(std::max(cfg.min_nodes, n));
		}
		std::sort(nodes.begin(), nodes.end(), [](const Seg& x, const Seg& y) { return x.j < y.j || (x.j == y.j && x.i < y.i); });
		thin_out();
		sweep();
		collect(out);
		join_chains(out);
	}
};

}  // namespace dmnd
