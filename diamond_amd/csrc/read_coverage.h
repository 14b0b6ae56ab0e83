// read_coverage.h -- an interval of a read and the coverage of the read by the alignments kept so far: what the reference's
// RangeCulling asks of IntervalPartition (/root/reference/src/output/target_culling.h:107-163, util/geo/interval_partition.h:46-187).
// Shared by the frameshift extension (frameshift_host.hip) and the range-culling block join (extend_host.hip, dmnd_join_blocks_range).
#pragma once
#include <algorithm>
#include <climits>
#include <cstdint>
#include <iterator>
#include <map>

namespace dmnd {

struct Interval {
	int b, e;
	int length() const { return e > b ? e - b : 0; }
	int overlap(const Interval& r) const { const int lo = std::max(b, r.b), hi = std::min(e, r.e); return hi > lo ? hi - lo : 0; }
};

// Coverage of the read by the alignments kept so far, as a step function over read positions: the step that begins at a key holds
// how many alignments cover it and the lowest / highest of their scores (the lowest only among the first `cap` of them).
// Behaviour of IntervalPartition (util/geo/interval_partition.h:46-187) incl. what a new boundary inherits.
struct Coverage {
	struct Step { int64_t count = 0; int min_score = INT_MAX, max_score = 0; };
	std::map<int, Step> steps;
	int64_t cap;
	explicit Coverage(int64_t cap) : cap(cap) { steps[0] = Step(); }
	void insert(Interval k, int score)
	{
		auto i = steps.lower_bound(k.b);
		if (i == steps.end()) i = steps.insert({ k.b, Step() }).first;            // (a boundary behind the last one starts from an empty step)
		else if (i->first != k.b) { const Step before = std::prev(i)->second; i = steps.insert({ k.b, before }).first; }
		Step last;
		for (; i != steps.end() && i->first < k.e; ++i) {
			last = i->second;
			Step& s = i->second;
			if (s.count < cap) s.min_score = std::min(s.min_score, score);
			s.max_score = std::max(s.max_score, score);
			++s.count;
		}
		if (i == steps.end() || i->first != k.e) steps[k.e] = last;
	}
	template<typename Pred>
	int covered_if(Interval k, Pred p) const
	{
		auto i = steps.lower_bound(k.b);
		if (i == steps.end() || i->first != k.b) --i;
		int c = 0;
		for (; i != steps.end() && i->first < k.e; ++i) {
			const auto next = std::next(i);
			if (p(i->second)) c += k.overlap(Interval{ i->first, next == steps.end() ? INT_MAX : next->first });
		}
		return c;
	}
	int covered(Interval k) const { return covered_if(k, [&](const Step& s) { return s.count >= cap; }); }
	int covered_max(Interval k, int score) const { return covered_if(k, [&](const Step& s) { return s.max_score >= score; }); }
	int covered_min(Interval k, int score) const { return covered_if(k, [&](const Step& s) { return s.count >= cap && s.min_score >= score; }); }
};

}  // namespace dmnd
