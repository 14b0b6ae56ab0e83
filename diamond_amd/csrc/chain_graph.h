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

// DMND_XD (xdrop_core.h): __host__ __device__ under hipcc. The chaining below is ONE source for both sides: the host instantiates
// it over std::vector (ChainWorkspace), the device planner (plan_kernels.hip, round 6) over fixed-capacity arrays in a lane's
// private memory (ChainWorkspaceT<FixedChainPolicy>); a target that does not fit those arrays is chained on the host.
namespace dmnd {

// std::vector's interface as far as the chaining uses it, over a fixed array; running out of room sets `overflow` (the caller
// then discards the result) instead of growing
template<typename T, int CAP>
struct FixedVec {
	T a[CAP];
	int n = 0;
	bool overflow = false;
	DMND_XD size_t size() const { return (size_t)n; }
	DMND_XD bool empty() const { return n == 0; }
	DMND_XD void clear() { n = 0; }
	DMND_XD void reserve(size_t) {}
	DMND_XD T* begin() { return a; }
	DMND_XD T* end() { return a + n; }
	DMND_XD const T* begin() const { return a; }
	DMND_XD const T* end() const { return a + n; }
	DMND_XD T& operator[](size_t i) { return a[i]; }
	DMND_XD const T& operator[](size_t i) const { return a[i]; }
	DMND_XD T& back() { return a[n - 1]; }
	DMND_XD const T& back() const { return a[n - 1]; }
	DMND_XD void push_back(const T& x) { if (n < CAP) a[n++] = x; else overflow = true; }
	DMND_XD void pop_back() { --n; }
	DMND_XD void resize(size_t m) { if ((int)m <= CAP) n = (int)m; else overflow = true; }      // (only ever shrinks in the chaining)
	DMND_XD void insert(T* at, const T& x)
	{
		if (n >= CAP) { overflow = true; return; }
		for (T* p = a + n; p > at; --p) *p = *(p - 1);
		*at = x; ++n;
	}
	DMND_XD void erase(T* at) { for (T* p = at; p + 1 < a + n; ++p) *p = *(p + 1); --n; }
	DMND_XD void reset() { n = 0; overflow = false; }      // an instance in memory no constructor ran on (LDS)
	DMND_XD void swap(FixedVec& o)                      // element by element: a whole-array temporary would be 0.5 KB of private memory on the device
	{
		const int m = n > o.n ? n : o.n;
		for (int i = 0; i < m; ++i) { const T t = a[i]; a[i] = o.a[i]; o.a[i] = t; }
		const int tn = n; n = o.n; o.n = tn;
		const bool tv = overflow; overflow = o.overflow; o.overflow = tv;
	}
	template<typename It> DMND_XD void append(It b, It e) { for (; b != e; ++b) push_back(*b); }
};

// what std::sort does for up to 16 elements (libstdc++: introsort leaves ranges of <= _S_threshold = 16 elements to ONE insertion
// sort, which keeps equal elements in their order) -- the device's sort; longer ranges do not occur there (FixedChainPolicy caps)
template<typename T, typename Cmp>
DMND_XD void insertion_sort(T* b, T* e, Cmp less)
{
	for (T* i = b == e ? e : b + 1; i < e; ++i) {
		const T v = *i;
		T* p = i;
		while (p > b && less(v, *(p - 1))) { *p = *(p - 1); --p; }
		*p = v;
	}
}

struct HostChainPolicy {
	template<typename T, int CAP> struct Vec : std::vector<T> {
		template<typename It> void append(It b, It e) { this->insert(this->end(), b, e); }
		bool overflowed() const { return false; }
	};
	template<typename It, typename Cmp> static void sort(It b, It e, Cmp c) { std::sort(b, e, c); }
	template<typename V> static bool overflowed(const V&) { return false; }
};

struct FixedChainPolicy {
	enum { SORT_MAX = 16 };
	template<typename T, int CAP> using Vec = FixedVec<T, CAP>;
	template<typename T, typename Cmp> DMND_XD static void sort(T* b, T* e, Cmp c) { insertion_sort(b, e, c); }
	template<typename V> DMND_XD static bool overflowed(const V& v) { return v.overflow; }
};

struct ScoreTable {                // ScoreMatrix::operator()(a,b) = matrix32[a*32+b] on masked letters
	int m[32 * 32];
	int gap_open, gap_extend;
	DMND_XD int at(int a, int b) const { return m[(a << 5) + b]; }
};

struct SeqRef {                    // Sequence: letters are read through & 31 (basic/sequence.h:80-87)
	const int8_t* p;
	int len;
	DMND_XD int operator[](int i) const { return p[i] & 31; }
};

struct Seg {                       // one ungapped diagonal segment: cells (i + k, j + k), k < len
	int i, j, len, score;
	DMND_XD int diag() const { return i - j; }
	DMND_XD int j_end() const { return j + len; }
	DMND_XD int j_last() const { return j + len - 1; }
	DMND_XD int i_end() const { return i + len; }
	DMND_XD int i_last() const { return i + len - 1; }
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

DMND_XD int chain_min(int a, int b) { return b < a ? b : a; }
DMND_XD int chain_max(int a, int b) { return a < b ? b : a; }
DMND_XD double chain_min(double a, double b) { return b < a ? b : a; }      // std::min / std::max: the second argument only when it is strictly smaller / larger
DMND_XD double chain_max(double a, double b) { return a < b ? b : a; }
DMND_XD int chain_abs(int a) { return a < 0 ? -a : a; }

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
// P: HostChainPolicy (std::vector, std::sort) or FixedChainPolicy (arrays of NODES segments, LINKS links, CHAINS chains; insertion sort)
template<typename P, int NODES = 16, int LINKS = 96, int CHAINS = 16>
struct ChainWorkspaceT {
	struct Node : Seg {
		int best, peak, dip;          // best chain score ending here; the highest and lowest running score on that chain's way
		int newest;                   // most recent incoming link (index into links), -1 = none
		DMND_XD int rel() const { return best == peak ? best : best - dip; }
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
	typedef typename P::template Vec<Node, NODES> NodeVec;
	typedef typename P::template Vec<Chain, CHAINS> ChainVec;

	NodeVec nodes, closed_, open_;
	typename P::template Vec<InLink, LINKS> links;
	typename P::template Vec<Front, NODES> front;
	typename P::template Vec<Frame, NODES + 1> frames;
	typename P::template Vec<unsigned, NODES> tops;
	const ScoreTable* S = nullptr;
	SeqRef query, subject;
	ChainCfg cfg;

	DMND_XD static Node node_of(const Seg& s) { Node n; n.i = s.i; n.j = s.j; n.len = s.len; n.score = s.score; n.best = n.peak = n.dip = s.score; n.newest = -1; return n; }

	// FixedChainPolicy only: puts an instance that lives in raw memory (the device planner's LDS) into its constructed state
	DMND_XD void reset_fixed()
	{
		nodes.reset(); closed_.reset(); open_.reset(); links.reset(); front.reset(); frames.reset(); tops.reset();
		S = nullptr; cfg = ChainCfg();
	}

	// true: some array of a fixed-capacity instance was too small -- the result is void (never with HostChainPolicy)
	DMND_XD bool overflowed() const
	{
		return P::overflowed(nodes) || P::overflowed(closed_) || P::overflowed(open_) || P::overflowed(links) || P::overflowed(front) || P::overflowed(frames) || P::overflowed(tops);
	}

	// Segments arrive sorted by (diagonal, j): a segment that begins inside the stretch already covered on its diagonal is dropped
	DMND_XD void load(const Seg* segs, size_t n_segs)
	{
		nodes.clear(); links.clear();
		bool have = false;
		int d = 0, covered = 0;
		for (size_t x = 0; x < n_segs; ++x) {
			const Seg& s = segs[x];
			if (have && s.diag() == d && s.j <= covered) continue;
			covered = have && s.diag() == d ? chain_max(covered, s.j_end()) : s.j_end();
			d = s.diag(); have = true;
			nodes.push_back(node_of(s));
		}
	}

	// Drops a segment when more than range_cover still-open segments (in subject order) score at least as much and cover its
	// whole subject range. The survivors come out in the order in which they stop being open.
	DMND_XD void thin_out()
	{
		NodeVec& closed = closed_;
		NodeVec& open = open_;
		closed.clear(); open.clear();
		closed.reserve(nodes.size());
		for (size_t k = 0; k < nodes.size(); ++k) {
			const Node d = nodes[k];
			size_t covering = 0, keep = 0;
			for (size_t x = 0; x < open.size(); ++x) {
				const Node w = open[x];
				if (w.j_end() <= d.j) { closed.push_back(w); continue; }
				covering += w.score >= d.score && w.j <= d.j && w.j_end() >= d.j_end();
				if (keep != x) open[keep] = w;
				++keep;
			}
			open.resize(keep);
			if (covering <= cfg.range_cover) open.push_back(d);
		}
		closed.append(open.begin(), open.end());
		nodes.swap(closed);
	}

	// Best incoming link of `node` among those whose downstream part begins before column j and that beat the bare
	// segment; of equally good ones the newest. -1 = none.
	DMND_XD int best_in(unsigned node, int j) const
	{
		int pick = -1, top = nodes[node].score;
		for (int l = nodes[node].newest; l >= 0; l = links[(size_t)l].older)
			if (links[(size_t)l].j < j && links[(size_t)l].best > top) { pick = l; top = links[(size_t)l].best; }
		return pick;
	}

	DMND_XD void add_link(unsigned to, const InLink& l)
	{
		Node& d = nodes[to];
		if (l.best > d.best) { d.best = l.best; d.peak = l.peak; d.dip = l.dip; }
		const int at = (int)links.size();
		links.push_back(l);
		if (P::overflowed(links)) return;
		links.back().older = d.newest;
		d.newest = at;
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
	struct Axis { int x, off, len, score; DMND_XD int end() const { return x + len; } DMND_XD int last() const { return x + len - 1; } };

	template<typename Cell>
	DMND_XD static bool junction_on(const Axis& up, const Axis& down, int pad, Cell W, int& c_best, int& t_best, int& r_best, int& total)
	{
		const int g = up.off - down.off;
		const bool apart = up.last() < down.x - g - 1;
		const int c0 = apart ? up.last() : chain_max(down.x - g - 1 - pad, up.x);
		const int entry_max = chain_min(chain_max(down.x, up.last() + g + 1 + pad), down.last());
		if (c0 + g + 1 > down.last()) return false;
		auto span = [&](int off, int a, int b) { int s = 0; for (int x = a; x < b; ++x) s += W(off, x); return s; };
		int run = 0;                                                                              // sum W(up.off, (c0, c])
		int r = down.score + span(down.off, c0 + g + 1, down.x) - span(down.off, down.x, c0 + g + 1);      // R(c + g + 1)
		total = INT_MIN;
		for (int c = c0;; ++c) {
			if (run + r > total) { total = run + r; c_best = c; r_best = r; }
			if (c + g + 1 >= entry_max) break;
			r -= W(down.off, c + g + 1);
			run += W(up.off, c + 1);
		}
		// T(c_best + 1)
		t_best = up.score + span(up.off, up.end(), c_best + 1) - span(up.off, c_best + 1, up.end());
		return true;
	}

	// Junction from segment e into segment d (e is the upstream one). Lower or equal diagonal of d: the segments advance
	// along the subject; higher diagonal: along the query (same computation with the roles of the sequences swapped).
	DMND_XD bool junction(const Seg& e, const Seg& d, Junction& out) const
	{
		int c = 0, t = 0, r = 0, total = 0;
		const ScoreTable* st = S;
		const SeqRef qs = query, ss = subject;
		if (e.diag() < d.diag()) {
			const Axis up{ e.i, -e.diag(), e.len, e.score }, down{ d.i, -d.diag(), d.len, d.score };
			auto cell = [&](int off, int x) { return st->at(ss[x + off], qs[x]); };
			if (!junction_on(up, down, cfg.link_padding, cell, c, t, r, total)) return false;
			const int g = up.off - down.off;
			out = Junction{ total, c + up.off, c, c + g + 1 + down.off, c + g + 1, t, r };
		}
		else {
			const Axis up{ e.j, e.diag(), e.len, e.score }, down{ d.j, d.diag(), d.len, d.score };
			auto cell = [&](int off, int x) { return st->at(qs[x + off], ss[x]); };
			if (!junction_on(up, down, cfg.link_padding, cell, c, t, r, total)) return false;
			const int g = up.off - down.off;
			out = Junction{ total, c, c + up.off, c + g + 1, c + g + 1 + down.off, t, r };
		}
		return true;
	}

	// Considers continuing the best chain of `from` with segment `to`, and records the link if it improves on the bare segment.
	DMND_XD void try_link(unsigned to, unsigned from)
	{
		const Node d = nodes[to];
		const Node e = nodes[from];
		const int shift = d.diag() - e.diag();
		const int gap = shift ? -(S->gap_open + chain_abs(shift) * S->gap_extend) : 0;
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
			const int e_best = in < 0 ? e_score : chain_max(e_score, links[(size_t)in].best);
			int peak = in < 0 ? e_score : chain_max(e_score, links[(size_t)in].peak);
			int dip = in < 0 ? e_score : links[(size_t)in].dip;
			l.best = e_best - cut + gap + jn.down;
			const int have2 = best_in(to, jn.s2);
			if (have2 >= 0 && links[(size_t)have2].best > l.best) return;
			l.begin = l.best - jn.down;
			dip = chain_min(dip, l.begin);
			if (e_best == peak) peak -= cut;
			l.peak = peak; l.dip = dip; l.j = jn.s2;
		}
		else {
			// free space between the segments: a flat penalty per skipped position
			l.best = e.best + gap - int(cfg.space_penalty * chain_max(apart - 1, 0)) + d.score;
			const int have = best_in(to, d.j);
			if (have >= 0 && links[(size_t)have].best > l.best) return;
			l.begin = l.best - d.score;
			l.peak = e.peak;
			l.dip = chain_min(e.dip, l.begin);
			l.j = d.j;
		}
		if (l.best <= d.score) return;
		l.peak = chain_max(l.peak, l.best);
		if (l.best == l.peak) l.dip = l.best;
		add_link(to, l);
	}

	DMND_XD bool faded(const Node& e, const Node& d) const { return e.best - int(cfg.space_penalty * chain_max(d.j - e.j_end(), 0)) <= 0; }
	DMND_XD bool overhangs(const Node& e, const Node& d) const { return e.j_end() - (d.j_end() - chain_min(e.diag() - d.diag(), 0)) >= cfg.reverse_overhang; }

	// Segments in (j, i) order; `front` holds for every diagonal the latest segment seen on it while that segment can still
	// contribute (its chain score has not faded over the distance). A new segment is linked to the frontier segments below
	// it (nearest diagonal first), then to those at and above it.
	DMND_XD void sweep()
	{
		front.clear();
		for (unsigned n = 0; n < nodes.size(); ++n) {
			const int dd = nodes[n].diag();
			size_t pos = 0;
			{       // first frontier entry whose diagonal is not below dd (std::lower_bound)
				size_t lo = 0, hi = front.size();
				while (lo < hi) { const size_t mid = (lo + hi) / 2; if (front[mid].diag < dd) lo = mid + 1; else hi = mid; }
				pos = lo;
			}
			const bool fresh = pos == front.size() || front[pos].diag != dd;
			if (fresh) front.insert(front.begin() + (ptrdiff_t)pos, Front{ dd, n });
			if (P::overflowed(front)) return;
			int reach_j = 0;
			for (size_t k = pos; k-- > 0;) {
				const unsigned en = front[k].node;
				if (faded(nodes[en], nodes[n])) { front.erase(front.begin() + (ptrdiff_t)k); --pos; continue; }
				if (nodes[en].j_end() < reach_j) continue;
				try_link(n, en);
				reach_j = chain_max(reach_j, chain_min(nodes[n].j, nodes[en].j_end()));
				if (overhangs(nodes[en], nodes[n])) try_link(en, n);
			}
			int reach_i = 0;
			for (size_t k = fresh ? pos + 1 : pos; k < front.size();) {
				const unsigned en = front[k].node;
				if (k != pos && faded(nodes[en], nodes[n])) { front.erase(front.begin() + (ptrdiff_t)k); continue; }
				if (nodes[en].i_end() >= reach_i) {
					try_link(n, en);
					if (nodes[en].i < nodes[n].i) reach_i = chain_max(reach_i, chain_min(nodes[en].i_end(), nodes[n].i));
					if (overhangs(nodes[en], nodes[n])) try_link(en, n);
				}
				++k;
			}
			front[pos].node = n;
		}
	}

	DMND_XD static double share(int a0, int a1, int b0, int b1)            // part of [a0, a1) that lies in [b0, b1)
	{
		const int lo = chain_max(a0, b0), hi = chain_min(a1, b1);
		return (double)(unsigned)(hi > lo ? hi - lo : 0) / (double)(a1 > a0 ? a1 - a0 : 0);
	}

	// May a candidate (ranges + score) coexist with the chains ts[first..)? A chain that the candidate mostly stacks on (in
	// either sequence) without being dwarfed by it does not count; otherwise what remains of the candidate outside that chain
	// must still be worth the cutoff.
	template<typename CV>
	DMND_XD bool compatible(const CV& ts, size_t first, int q0, int q1, int s0, int s1, int score) const
	{
		for (size_t x = first; x < ts.size(); ++x) {
			const double in_s = share(s0, s1, ts[x].s0, ts[x].s1), in_q = share(q0, q1, ts[x].q0, ts[x].q1);
			if ((1.0 - chain_min(in_s, in_q)) * score / ts[x].score >= cfg.stacked_hsp_ratio) continue;
			if ((1.0 - chain_max(in_s, in_q)) * score < cfg.cutoff) return false;
		}
		return true;
	}

	// Follows the best links from `top` back to where its chain begins. The chain may not pass through a point whose
	// running score exceeds the chain's final score; a part reached through such a link is cut off if the chain is still worth
	// more from there on. `jump` receives the segment behind a link that crosses more than max_shift diagonals (the walk stops
	// in front of it and the caller continues from there), else UINT_MAX. Returns false if nothing could be followed.
	DMND_XD bool follow(unsigned top, int j_bound, Chain& t, unsigned& jump)
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
			dip = chain_min(dip, l < 0 ? 0 : links[(size_t)l].begin);
			frames.push_back(Frame{ cur, l, dip });
			if (P::overflowed(frames)) return false;
			if (l < 0) break;
			const InLink& in = links[(size_t)l];
			const int shift = nodes[cur].diag() - nodes[in.from].diag();
			if (chain_abs(shift) > cfg.max_shift) { jump = in.from; break; }
			j_bound = shift > 0 ? in.j : in.j + shift;
			cur = in.from;
		}
		if (rejected)       // drop the frames that cannot do without what lies behind them
			while (!frames.empty() && links[(size_t)frames.back().link].begin > frames.back().dip) frames.pop_back();
		if (frames.empty()) return false;
		const Node& first = nodes[frames.back().node];
		t.q0 = first.i; t.s0 = first.j; t.score = final_score - frames.back().dip;
		for (size_t k = 0; k < frames.size(); ++k) {
			const int dd = nodes[frames[k].node].diag();
			t.d_min = chain_min(t.d_min, dd);
			t.d_max = chain_max(t.d_max, dd);
		}
		return true;
	}

	template<typename CV>
	DMND_XD void collect(CV& ts)
	{
		tops.clear();
		for (unsigned n = 0; n < nodes.size(); ++n)
			if (nodes[n].rel() >= cfg.cutoff) tops.push_back(n);
		// same comparison sequence as the reference's sort of its candidate list, so equal scores end up in the same order
		const NodeVec& nd = nodes;
		P::sort(tops.begin(), tops.end(), [&nd](unsigned x, unsigned y) { return nd[x].rel() > nd[y].rel(); });
		const size_t none = (size_t)-1;
		size_t first = none;                              // first chain of this call in ts
		for (size_t k = 0; k < tops.size(); ++k) {
			const unsigned n = tops[k];
			const Node d = nodes[n];
			if (!compatible(ts, first == none ? ts.size() : first, d.i, d.i_end(), d.j, d.j_end(), d.score)) continue;
			unsigned from = n;
			int j_bound = subject.len;
			while (from != UINT_MAX) {
				const Node top = nodes[from];
				Chain t{ INT_MAX, INT_MIN, 0, 0, top.i_end(), 0, top.j_end() };
				unsigned jump;
				follow(from, chain_min(top.j_end(), j_bound), t, jump);
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
	DMND_XD static int joined_score(const Chain& a, const Chain& b)
	{
		const int dq = b.q0 - a.q1, ds = b.s0 - a.s1;
		if (dq < 0 || ds < 0) return 0;
		const int longer = chain_max(dq, ds), shorter = chain_min(dq, ds);
		// the reference subtracts the two costs in this order (double arithmetic, truncated)
		return int((a.score + b.score) - longer * 0.5 - shorter * 0.1);
	}

	template<typename CV>
	DMND_XD static void join_chains(CV& h)
	{
		for (size_t a = 0; a < h.size(); ++a)
			for (size_t b = a + 1; b < h.size();) {
				const int limit = chain_max(h[a].score, h[b].score);
				const bool ab = joined_score(h[a], h[b]) > limit, ba = !ab && joined_score(h[b], h[a]) > limit;
				if (!ab && !ba) { ++b; continue; }
				const Chain x = ab ? h[a] : h[b], y = ab ? h[b] : h[a];
				h[a] = Chain{ chain_min(x.d_min, y.d_min), chain_max(x.d_max, y.d_max), joined_score(x, y), x.q0, y.q1, x.s0, y.s1 };
				h.erase(h.begin() + (ptrdiff_t)b);
			}
	}

	// segments sorted by (diagonal, j) in, chains out (unsorted). With FixedChainPolicy the caller checks overflowed() and
	// P::overflowed(out) afterwards.
	template<typename CV>
	DMND_XD void run_segs(const ScoreTable& st, const SeqRef& q, const SeqRef& s, const Seg* segs, size_t n_segs, CV& out)
	{
		out.clear();
		if (n_segs == 1) {
			const Seg& g = segs[0];
			out.push_back(Chain{ g.diag(), g.diag(), g.score, g.i, g.i_end(), g.j, g.j_end() });
			return;
		}
		S = &st; query = q; subject = s;
		load(segs, n_segs);
		auto by_score = [](const Seg& x, const Seg& y) { return x.score > y.score; };
		if (cfg.maxnodes > 0) {
			P::sort(nodes.begin(), nodes.end(), by_score);
			if (nodes.size() > cfg.maxnodes) nodes.resize(cfg.maxnodes);
		}
		if (cfg.len_cap > 0.0 && nodes.size() > cfg.min_nodes) {
			// many segments: the best-scoring ones until their lengths add up to len_cap query lengths, at least min_nodes
			P::sort(nodes.begin(), nodes.end(), by_score);
			const double cap = q.len * cfg.len_cap;
			double total = 0.0;
			size_t n = 0;
			while (n < nodes.size() && total < cap) total += nodes[n++].len;
			nodes.resize(min_nodes_or(n));
		}
		P::sort(nodes.begin(), nodes.end(), [](const Seg& x, const Seg& y) { return x.j < y.j || (x.j == y.j && x.i < y.i); });
		thin_out();
		sweep();
		collect(out);
		join_chains(out);
	}
	DMND_XD size_t min_nodes_or(size_t n) const { return cfg.min_nodes < n ? n : cfg.min_nodes; }

	void run(const ScoreTable& st, const SeqRef& q, const SeqRef& s, const std::vector<Seg>& segs, std::vector<Chain>& out)
	{
		run_segs(st, q, s, segs.data(), segs.size(), out);
	}
};

typedef ChainWorkspaceT<HostChainPolicy> ChainWorkspace;

}  // namespace dmnd
