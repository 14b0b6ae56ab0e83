// oracle/chain_ref.h -- TEST INFRASTRUCTURE ONLY (the checker of diamond_amd/csrc/chain_graph.h; never compiled into the product).
// Round 1's host-side chaining: a statement-by-statement restatement of the reference's functions listed below, which is why
// it was retired from the product path. It stays here as the known-good answer for tests/test_chain_graph.py: it reproduced
// the reference's DpTargets on every golden of round 1 (tests/test_extend_plan.py).
// Original header: host-side (CPU, per target) part of the extension stage that fixes the band geometry of the
// GPU Smith-Waterman: x-drop ungapped extension of the seed hits and greedy chaining of the resulting diagonal
// segments into approximate HSPs. Branchy, tiny per target, stays on the host.
//
// It must reproduce the reference's heuristics exactly, because they decide [d_begin, d_end) of every DpTarget and
// therefore scores and CIGARs downstream:
//   xdrop_ungapped                         /root/reference/src/dp/ungapped_align.cpp:151-199
//   ungapped_stage                         src/align/ungapped.cpp:62-126
//   DiagGraph load/sort/prune/edges        src/chaining/diag_graph.h:27-190, greedy_align.cpp:49-125
//   links between segments                 src/chaining/greedy_align.cpp:127-236
//   Aligner::get_approximate_link, forward_pass, run     greedy_align.cpp:238-413
//   Aligner::backtrace*, disjoint          src/chaining/backtrace.cpp:36-357
//   merge_hsps, Chaining::run              greedy_align.cpp:417-497
//   Extension::band, add_dp_targets        src/align/gapped_score.cpp:41-180
// Only what the default blastp path reads is carried (d_min, d_max, score, query/subject ranges of a chain); the
// reference's logging/transcript side outputs of the chaining are not produced.
// Written from scratch around flat vectors (one ChainWorkspace per host thread) rather than the reference's
// thread-local graph object + std::map window + std::list outputs.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <map>
#include <vector>

namespace dmnd_ref {

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

struct Seg {                       // DiagonalSegment (util/geo/diagonal_segment.h)
	int i, j, len, score;
	int diag() const { return i - j; }
	int j_end() const { return j + len; }
	int j_last() const { return j + len - 1; }
	int i_end() const { return i + len; }
	int i_last() const { return i + len - 1; }
};

struct Chain {                     // the fields of ApproxHsp the extension reads (util/hsp/approx_hsp.h:58-133)
	int d_min, d_max, score;
	int q0, q1, s0, s1;            // query_range / subject_range
};

struct HostSeedHit { int i, j, score, frame; };

// chaining constants = the reference's config defaults (basic/config.cpp:549-603)
struct ChainCfg {
	int xdrop = 20;                       // config.raw_ungapped_xdrop = rawscore(12.3 bits), config.cpp:428,853
	int max_shift = 2000;                 // chaining_maxgap
	size_t range_cover = 8;               // chaining_range_cover
	size_t maxnodes = 0;                  // chaining_maxnodes
	double len_cap = 2.0;                 // chaining_len_cap
	size_t min_nodes = 200;               // chaining_min_nodes
	double stacked_hsp_ratio = 0.5;       // chaining_stacked_hsp_ratio
	int cutoff = 19;                      // Aligner::run(..., SPACE_PENALTY, 19, band)
	double space_penalty = 0.1;
};

// ---- x-drop ungapped extension ------------------------------------------------------------------------------------
inline Seg xdrop_ungapped(const ScoreTable& S, const SeqRef& q, const int8_t* cbs, const SeqRef& t, int qa, int sa, int xdrop)
{
	int score = 0, st = 0, n = 1, delta = 0, len = 0;
	int a = qa - 1, b = sa - 1, ql, sl;
	// reads one position beyond either end: the blocks carry delimiters/padding there (string_set.h:34-47)
	while (score - st < xdrop && (ql = q[a]) != 31 && (sl = t[b]) != 31) {
		st += S.at(ql, sl);
		if (cbs) st += cbs[a];
		if (st > score) { score = st; delta = n; }
		--a; --b; ++n;
	}
	a = qa; b = sa; st = score; n = 1;
	while (score - st < xdrop && (ql = q[a]) != 31 && (sl = t[b]) != 31) {
		st += S.at(ql, sl);
		if (cbs) st += cbs[a];
		if (st > score) { score = st; len = n; }
		++a; ++b; ++n;
	}
	return Seg{ qa - delta, sa - delta, len + delta, score };
}

// ---- chaining -----------------------------------------------------------------------------------------------------
struct ChainWorkspace {
	struct Node : Seg {
		int link_idx, prefix_score, path_max, path_min;
		int rel_score() const { return prefix_score == path_max ? prefix_score : prefix_score - path_min; }
	};
	struct Edge { int prefix_score, path_max, j, path_min, prefix_score_begin; unsigned node_in, node_out; };
	struct Link { int s_pos1, q_pos1, s_pos2, q_pos2, score1, score2; };

	std::vector<Node> nodes;
	std::vector<Edge> edges;
	std::map<int, unsigned> window;
	const ScoreTable* S = nullptr;
	SeqRef query, subject;
	ChainCfg cfg;

	static Node make_node(const Seg& s) { Node n; (Seg&)n = s; n.link_idx = -1; n.prefix_score = n.path_max = n.path_min = s.score; return n; }

	int score_range(const SeqRef& a, const SeqRef& b, int i, int j, int j_end) const
	{
		int s = 0;
		for (; j < j_end; ++i, ++j) s += S->at(a[i], b[j]);
		return s;
	}

	// DiagGraph::load: drop segments that start inside the previous kept stretch of the same diagonal
	void load(const Seg* begin, const Seg* end)
	{
		nodes.clear(); edges.clear();
		int d = INT_MIN, max_j_end = INT_MIN;
		for (const Seg* s = begin; s < end; ++s) {
			if (s->diag() != d) { d = s->diag(); nodes.push_back(make_node(*s)); max_j_end = s->j_end(); }
			else if (max_j_end < s->j) { nodes.push_back(make_node(*s)); max_j_end = std::max(max_j_end, s->j_end()); }
		}
	}

	void prune()
	{
		std::vector<Node> finished, win;          // `win` keeps insertion order like the reference's std::list
		for (const Node& d : nodes) {
			size_t n = 0;
			for (size_t x = 0; x < win.size();) {
				if (win[x].j_end() > d.j) {
					if (win[x].score >= d.score && win[x].j <= d.j && win[x].j_end() >= d.j_end()) ++n;
					++x;
				}
				else { finished.push_back(win[x]); win.erase(win.begin() + (ptrdiff_t)x); }
			}
			if (n <= cfg.range_cover) win.push_back(d);
		}
		for (const Node& d : win) finished.push_back(d);
		nodes.swap(finished);
	}

	// DiagGraph::add_edge: edges of one node are contiguous, inserted at link_idx; later nodes' indices shift
	void add_edge(const Edge& e)
	{
		for (size_t k = e.node_in + 1; k < nodes.size(); ++k) {
			if (nodes[k].link_idx == -1) break;
			++nodes[k].link_idx;
		}
		Node& d = nodes[e.node_in];
		if (e.prefix_score > d.prefix_score) { d.prefix_score = e.prefix_score; d.path_max = e.path_max; d.path_min = e.path_min; }
		edges.insert(edges.begin() + d.link_idx++, e);
	}

	// DiagGraph::get_edge: best incoming edge of `node` ending before column j; -1 = none
	ptrdiff_t get_edge(size_t node, int j) const
	{
		const Node& d = nodes[node];
		if (d.score == 0) return (ptrdiff_t)d.link_idx - 1;
		if (edges.empty()) return -1;
		int max_score = d.score;
		ptrdiff_t best = -1;
		for (ptrdiff_t i = (ptrdiff_t)d.link_idx - 1; i >= 0 && edges[(size_t)i].node_in == node; --i)
			if (edges[(size_t)i].j < j && edges[(size_t)i].prefix_score > max_score) { best = i; max_score = edges[(size_t)i].prefix_score; }
		return best;
	}

	int prefix_score_at(size_t node, int j, int& path_max, int& path_min) const
	{
		const ptrdiff_t e = get_edge(node, j);
		const int sc = nodes[node].score;
		if (e < 0) { path_max = sc; path_min = sc; return sc; }
		path_max = std::max(sc, edges[(size_t)e].path_max);
		path_min = edges[(size_t)e].path_min;
		return std::max(sc, edges[(size_t)e].prefix_score);
	}

	// get_hgap_link (greedy_align.cpp:153-215): best junction between d1 (left/upper) and d2 on a lower diagonal
	int hgap_link(const Seg& d1, const Seg& d2, const SeqRef& q, const SeqRef& s, Link& l, int padding) const
	{
		const int d = d1.diag() - d2.diag(),
			j2_end = std::min(std::max(d2.j, d1.j_last() + d + 1 + padding), d2.j_last());
		int j1;
		bool space;
		if (d1.j_last() < d2.j - d - 1) { j1 = d1.j_last(); space = true; }
		else { j1 = std::max(d2.j - d - 1 - padding, d1.j); space = false; }
		int j2 = j1 + d + 1, i1 = d1.i + (j1 - d1.j), i2 = i1 + 1;
		if (j2 > d2.j_last()) { l.s_pos1 = -1; l.score1 = 0; l.score2 = 0; return INT_MIN; }
		int score1 = 0, score2 = score_range(q, s, i2, j2, d2.j) + d2.score - score_range(q, s, d2.i, d2.j, j2);
		int max_score = INT_MIN;
		for (;;) {
			if (score1 + score2 > max_score) {
				max_score = score1 + score2;
				l.q_pos1 = i1; l.s_pos1 = j1; l.q_pos2 = i2; l.s_pos2 = j2; l.score1 = score1; l.score2 = score2;
			}
			score2 -= S->at(q[i2], s[j2]);
			++i1; ++i2; ++j1; ++j2;
			if (j2 > j2_end) break;
			score1 += S->at(q[i1], s[j1]);
		}
		const int j1_end = j2_end - d;
		if (space) l.score1 += d1.score;
		else l.score1 += d1.score - score_range(q, s, d1.diag() + j1_end, j1_end, d1.j_end()) + score_range(q, s, d1.i_end(), d1.j_end(), j1_end) - score1;
		return max_score;
	}

	int link(const Seg& d1, const Seg& d2, Link& l, int padding) const
	{
		if (d1.diag() < d2.diag()) {      // vertical gap: the transposed problem
			const Seg t1{ d1.j, d1.i, d1.len, d1.score }, t2{ d2.j, d2.i, d2.len, d2.score };
			const int s = hgap_link(t1, t2, subject, query, l, padding);
			std::swap(l.s_pos1, l.q_pos1); std::swap(l.s_pos2, l.q_pos2);
			return s;
		}
		return hgap_link(d1, d2, query, subject, l, padding);
	}

	int approximate_link(unsigned d_idx, unsigned e_idx)
	{
		Node& d = nodes[d_idx];
		Node& e = nodes[e_idx];
		const int shift = d.diag() - e.diag();
		const int gap_score = shift != 0 ? -S->gap_open - std::abs(shift) * S->gap_extend : 0;
		const int space = shift > 0 ? d.j - e.j_last() : d.i - e.i_last();
		int prefix_score = 0, link_j = 0, path_max = 0, path_min = 0, prefix_score_begin = 0;
		if (space <= 0 || cfg.space_penalty == 0.0) {
			const ptrdiff_t ed = get_edge(d_idx, d.j);
			if (ed >= 0 && edges[(size_t)ed].prefix_score > e.prefix_score + gap_score + d.score) return 0;
			Link l;
			if (link(e, d, l, 10) > 0) {
				const int diff1 = e.score - l.score1;
				const int prefix_e = prefix_score_at(e_idx, l.s_pos1, path_max, path_min);
				prefix_score = prefix_e - diff1 + gap_score + l.score2;
				const ptrdiff_t ed2 = get_edge(d_idx, l.s_pos2);
				if (ed2 >= 0 && edges[(size_t)ed2].prefix_score > prefix_score) return 0;
				prefix_score_begin = prefix_score - l.score2;
				path_min = std::min(path_min, prefix_score - l.score2);
				if (prefix_e == path_max) path_max -= diff1;
				link_j = l.s_pos2;
			}
		}
		else {
			prefix_score = e.prefix_score + gap_score - int(cfg.space_penalty * std::max(space - 1, 0)) + d.score;
			const ptrdiff_t ed = get_edge(d_idx, d.j);
			if (ed >= 0 && edges[(size_t)ed].prefix_score > prefix_score) return 0;
			prefix_score_begin = prefix_score - d.score;
			path_max = e.path_max;
			path_min = std::min(e.path_min, prefix_score - d.score);
			link_j = d.j;
		}
		if (prefix_score > d.score) {
			path_max = std::max(path_max, prefix_score);
			add_edge(Edge{ prefix_score, path_max, link_j, prefix_score == path_max ? prefix_score : path_min, prefix_score_begin, d_idx, e_idx });
		}
		return prefix_score;
	}

	void forward_pass()
	{
		window.clear();
		const double sp = cfg.space_penalty;
		for (unsigned node = 0; node < nodes.size(); ++node) {
			nodes[node].link_idx = (int)edges.size();                 // DiagGraph::init(node)
			const int dd = nodes[node].diag();
			auto i = window.find(dd);
			if (i == window.end()) i = window.insert(std::make_pair(dd, node)).first;
			auto j = i;
			int max_j = 0;
			if (i != window.begin()) {
				do {
					--j;
					const Node& d = nodes[node];
					const Node& e = nodes[j->second];
					if (e.prefix_score - int(sp * std::max(d.j - e.j_end(), 0)) <= 0) {
						if (j == window.begin()) { window.erase(j); break; }
						auto k = j; ++k;
						window.erase(j);
						j = k;
						continue;
					}
					if (e.j_end() < max_j) continue;
					const unsigned e_idx = j->second;
					approximate_link(node, e_idx);
					{
						const Node& d2 = nodes[node];
						const Node& e2 = nodes[e_idx];
						max_j = std::max(max_j, std::min(d2.j, e2.j_end()));
						if (e2.j_end() - (d2.j_end() - std::min(e2.diag() - d2.diag(), 0)) >= 10)
							approximate_link(e_idx, node);
					}
				} while (j != window.begin());
			}
			j = i;
			if (j->second == node) ++j;
			int max_i = 0;
			while (j != window.end()) {
				const Node& d = nodes[node];
				const Node& e = nodes[j->second];
				if (e.prefix_score - int(sp * std::max(d.j - e.j_end(), 0)) <= 0 && j != i) {
					auto k = j; ++k;
					window.erase(j);
					j = k;
					continue;
				}
				if (e.i_end() < max_i) { ++j; continue; }
				const unsigned e_idx = j->second;
				approximate_link(node, e_idx);
				{
					const Node& d2 = nodes[node];
					const Node& e2 = nodes[e_idx];
					if (e2.i < d2.i) max_i = std::max(max_i, std::min(e2.i_end(), d2.i));
					if (e2.j_end() - (d2.j_end() - std::min(e2.diag() - d2.diag(), 0)) >= 10)
						approximate_link(e_idx, node);
				}
				++j;
			}
			i->second = node;
		}
	}

	static double overlap_factor(int a0, int a1, int b0, int b1)       // Interval(a).overlap_factor(Interval(b))
	{
		const int lo = std::max(a0, b0), hi = std::min(a1, b1);
		const int ov = hi > lo ? hi - lo : 0, len = a1 > a0 ? a1 - a0 : 0;
		return (double)(unsigned)ov / (double)len;
	}

	bool disjoint(const std::vector<Chain>& ts, size_t begin, int q0, int q1, int s0, int s1, int score) const
	{
		for (size_t x = begin; x < ts.size(); ++x) {
			const double ot = overlap_factor(s0, s1, ts[x].s0, ts[x].s1), oq = overlap_factor(q0, q1, ts[x].q0, ts[x].q1);
			if ((1.0 - std::min(ot, oq)) * score / ts[x].score >= cfg.stacked_hsp_ratio) continue;
			if ((1.0 - std::max(ot, oq)) * score < cfg.cutoff) return false;
		}
		return true;
	}

	// Aligner::backtrace_old (the recursive form the reference calls, backtrace.cpp:78-167), without transcript output
	bool walk(size_t node, int j_end, Chain& t, int score_max, int score_min, unsigned& next) const
	{
		const Node& d = nodes[node];
		const ptrdiff_t f = get_edge(node, j_end);
		bool at_end = f < 0 || (size_t)f >= edges.size();
		const int prefix_score = at_end ? d.score : edges[(size_t)f].prefix_score;
		if (prefix_score > score_max) return false;
		score_min = std::min(score_min, at_end ? 0 : edges[(size_t)f].prefix_score_begin);
		if (!at_end) {
			const Edge& ef = edges[(size_t)f];
			const Node& e = nodes[ef.node_out];
			const int shift = d.diag() - e.diag();
			const int j = ef.j;
			if (std::abs(shift) <= cfg.max_shift) {
				if (!walk(ef.node_out, shift > 0 ? j : j + shift, t, score_max, score_min, next)) {
					if (ef.prefix_score_begin > score_min) return false;
					at_end = true;
				}
			}
			else { next = ef.node_out; at_end = true; }
		}
		if (at_end) { t.q0 = d.i; t.s0 = d.j; t.score = score_max - score_min; }
		const int dd = d.diag();
		t.d_max = std::max(t.d_max, dd);
		t.d_min = std::min(t.d_min, dd);
		return true;
	}

	int backtrace_all(std::vector<Chain>& ts)
	{
		std::vector<const Node*> top;
		for (const Node& d : nodes)
			if (d.rel_score() >= cfg.cutoff) top.push_back(&d);
		std::sort(top.begin(), top.end(), [](const Node* x, const Node* y) { return x->rel_score() > y->rel_score(); });
		int max_score = 0;
		size_t t_begin = ts.size();                       // == ts.end() until the first chain is stored
		bool have_begin = false;
		for (const Node* n : top) {
			const size_t first = have_begin ? t_begin : ts.size();
			if (!disjoint(ts, first, n->i, n->i + n->len, n->j, n->j + n->len, n->score)) continue;
			// Aligner::backtrace(top_node, hsps, ts, t_begin, cutoff, max_shift)
			size_t top_node = (size_t)(n - nodes.data());
			unsigned next;
			int max_j = subject.len;
			do {
				Chain t{ INT_MAX, INT_MIN, 0, 0, 0, 0, 0 };
				next = UINT_MAX;
				{
					const Node& d = nodes[top_node];
					t.s1 = d.j_end(); t.q1 = d.i_end();
					walk(top_node, std::min(d.j_end(), max_j), t, d.prefix_score, d.prefix_score, next);
				}
				if (t.score > 0) max_j = t.s0;
				const size_t db = have_begin ? t_begin : ts.size();
				if (t.score >= cfg.cutoff && disjoint(ts, db, t.q0, t.q1, t.s0, t.s1, t.score)) {
					if (!have_begin) { t_begin = ts.size(); have_begin = true; }
					ts.push_back(t);
					max_score = std::max(max_score, t.score);
				}
				top_node = next;
			} while (next != UINT_MAX);
		}
		return max_score;
	}

	static int merge_score(const Chain& a, const Chain& b)
	{
		const int gq = b.q0 - a.q1, gt = b.s0 - a.s1;
		if (gq < 0 || gt < 0) return 0;
		const int s = a.score + b.score;
		return gq > gt ? int(s - gq * 0.5 - gt * 0.1) : int(s - gt * 0.5 - gq * 0.1);
	}

	static void merge_chains(std::vector<Chain>& h)
	{
		for (size_t a = 0; a < h.size(); ++a)
			for (size_t b = a + 1; b < h.size();) {
				const int m1 = merge_score(h[a], h[b]), m2 = merge_score(h[b], h[a]), mx = std::max(h[a].score, h[b].score);
				if (m1 > mx || m2 > mx) {
					const Chain& x = m1 > mx ? h[a] : h[b];
					const Chain& y = m1 > mx ? h[b] : h[a];
					Chain m{ std::min(x.d_min, y.d_min), std::max(x.d_max, y.d_max), merge_score(x, y), x.q0, y.q1, x.s0, y.s1 };
					h[a] = m;
					h.erase(h.begin() + (ptrdiff_t)b);
				}
				else ++b;
			}
	}

	// Chaining::run: segments sorted by (diagonal, j) in, chains out (unsorted)
	void run(const ScoreTable& st, const SeqRef& q, const SeqRef& s, const std::vector<Seg>& segs, std::vector<Chain>& out)
	{
		out.clear();
		if (segs.size() == 1) {
			const Seg& g = segs[0];
			out.push_back(Chain{ g.diag(), g.diag(), g.score, g.i, g.i + g.len, g.j, g.j + g.len });
			return;
		}
		S = &st; query = q; subject = s;
		load(segs.data(), segs.data() + segs.size());
		if (cfg.maxnodes > 0) {
			std::sort(nodes.begin(), nodes.end(), [](const Seg& x, const Seg& y) { return x.score > y.score; });
			if (nodes.size() > cfg.maxnodes) nodes.erase(nodes.begin() + (ptrdiff_t)cfg.maxnodes, nodes.end());
		}
		if (cfg.len_cap > 0.0 && nodes.size() > cfg.min_nodes) {
			std::sort(nodes.begin(), nodes.end(), [](const Seg& x, const Seg& y) { return x.score > y.score; });
			const double cap = q.len * cfg.len_cap;
			double total = 0.0;
			size_t it = 0;
			while (it < nodes.size() && total < cap) { total += nodes[it].len; ++it; }
			nodes.erase(nodes.begin() + (ptrdiff_t)std::max(cfg.min_nodes, it), nodes.end());
		}
		std::sort(nodes.begin(), nodes.end(), [](const Seg& x, const Seg& y) { return x.j < y.j || (x.j == y.j && x.i < y.i); });
		prune();
		forward_pass();
		backtrace_all(out);
		merge_chains(out);
	}
};

}  // namespace dmnd_ref
