// seg_mask.h -- SEG low-complexity segments of a protein sequence (Wootton & Federhen 1993/1996) as the reference's `--masking seg`
// finds them: NCBI's SEG with window 10, K2 trigger 1.8, extension 2.1, trimming of up to 50 letters, at most 2 letters outside
// the 20 standard residues per window, no merging of segments (SeqBufferSeg + SegParametersNewAa,
// /root/reference/src/lib/blast/blast_seg.cpp:1732-2314; applied by Masking::operator(), src/masking/masking.cpp:172-192).
// Host code (the reference masks with SEG on the host too); integer state + double arithmetic in the reference's operation order.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace dmnd { namespace seg {

enum { WINDOW = 10, MAX_TRIM = 50, MAX_BOGUS = 2, ALPHA = 20, DOWNSET = (WINDOW + 1) / 2 - 1, UPSET = WINDOW - DOWNSET };
constexpr double LOCUT = 1.8, HICUT = 2.1, LN2 = 0.69314718055994530941723212145818, LN20 = 2.9957322735539909;

// ln(n!): the reference reads a table of 10001 values printed with six decimals (blast_seg.cpp:53-1308) and switches to Stirling's
// formula above it (s_lnfact, :1833-1838). The table is rebuilt here from exact sums, rounded the same way.
struct LnFact {
	std::vector<double> table;
	LnFact() : table(10001)
	{
		long double acc = 0.0L;
		for (int n = 0; n <= 10000; ++n) {
			if (n >= 2) acc += std::log((long double)n);
			char b[40];
			std::snprintf(b, sizeof b, "%.6Lf", acc);
			table[(size_t)n] = std::strtod(b, nullptr);
		}
	}
	double operator()(uint32_t n) const { return n < table.size() ? table[n] : ((n + 0.5) * std::log((double)n) - n + 0.9189385332); }
};

inline const LnFact& lnfact() { static const LnFact t; return t; }

// State vector of a window: the counts of the standard residues in it, descending, zero-terminated (slots ALPHA + 1); letters
// outside the 20 standard residues are "bogus" and only counted.
struct Window {
	int comp[ALPHA];
	int state[ALPHA + 1];
	int bogus;
	void open(const int8_t* s, int len)
	{
		std::fill(comp, comp + ALPHA, 0);
		bogus = 0;
		for (int i = 0; i < len; ++i) { const int l = s[i] & 31; if (l < ALPHA) ++comp[l]; else ++bogus; }
		int n = 0;
		for (int l = 0; l < ALPHA; ++l) if (comp[l]) state[n++] = comp[l];
		std::sort(state, state + n, [](int a, int b) { return a > b; });
		std::fill(state + n, state + ALPHA + 1, 0);
	}
	// one letter leaves on the left, one enters on the right (s_ShiftWin1: s_DecrementSV / s_IncrementSV keep the order)
	void shift(int out, int in)
	{
		out &= 31; in &= 31;
		if (out < ALPHA) {
			const int cl = comp[out]--;
			for (int* sv = state; *sv != 0; ++sv) if (*sv == cl && sv[1] < cl) { *sv = cl - 1; break; }
		}
		else --bogus;
		if (in < ALPHA) {
			const int cl = comp[in]++;
			for (int* sv = state;; ++sv) if (*sv == cl) { ++*sv; break; }
		}
		else ++bogus;
	}
};

// K2 entropy of a state vector in bits (s_Entropy, :1586-1607)
inline double entropy(const int* sv)
{
	int total = 0;
	for (int i = 0; sv[i] != 0; ++i) total += sv[i];
	if (total == 0) return 0.0;
	double ent = 0.0;
	for (int i = 0; sv[i] != 0; ++i) ent += ((double)sv[i]) * std::log(((double)sv[i]) / (double)total) / LN2;
	return std::fabs(ent / (double)total);
}

// The entropy of a trigger window only depends on its state vector, and a window of 10 letters has 139 of them: the values are
// kept per thread (same function of the same input, so the numbers are the ones entropy() returns).
inline double window_entropy(const int* sv)
{
	struct Entry { uint64_t key; double value; };
	thread_local Entry cache[512] = {};
	uint64_t key = 1;                                   // counts are at most WINDOW = 10: four bits each, at most ten of them
	for (int i = 0; sv[i] != 0; ++i) key = (key << 4) | (uint64_t)sv[i];
	Entry& e = cache[(key * 0x9E3779B97F4A7C15ULL) >> 55];
	if (e.key != key) { e.key = key; e.value = entropy(sv); }
	return e.value;
}

// ln of the number of compositions of a complexity state (s_LnAss, :1871-1911): 20! / (product over the groups of equal counts
// of their multiplicity!, the zero counts being one group)
inline double ln_ass(const int* sv)
{
	const LnFact& F = lnfact();
	double ans = F.table[ALPHA];
	if (sv[0] == 0) return ans;
	int total = ALPHA, cl = 1, i = 0;
	int svi = sv[0], svim1 = sv[0];
	for (;; svim1 = svi) {
		if (++i == ALPHA) { ans -= F((uint32_t)cl); break; }
		else if ((svi = *++sv) == svim1) { ++cl; continue; }
		else {
			total -= cl;
			ans -= F((uint32_t)cl);
			if (svi == 0) { ans -= F((uint32_t)total); break; }
			cl = 1;
		}
	}
	return ans;
}

// ln P0 of a window (s_GetProb, :1922-1944): compositions x permutations / 20^length
inline double get_prob(const int* sv, int total)
{
	const LnFact& F = lnfact();
	const double totseq = ((double)total) * LN20;
	const double ans1 = ln_ass(sv);
	double ans2 = 0;
	if (ans1 > -100000.0) {
		ans2 = F((uint32_t)total);                       // s_LnPerm
		for (int i = 0; sv[i] != 0; ++i) ans2 -= F((uint32_t)sv[i]);
	}
	return ans1 + ans2 - totseq;
}

struct Range { int begin, end; };        // inclusive

// s_Trim (:1952-1998): the sub-window of s[0, len) -- at most MAX_TRIM letters shorter -- with the lowest P0
inline void trim(const int8_t* s, int len, int& leftend, int& rightend)
{
	int lend = 0, rend = len - 1, minlen = 1;
	if (len - MAX_TRIM > minlen) minlen = len - MAX_TRIM;
	double minprob = 1.0;
	Window w;
	for (int l = len; l > minlen; --l) {
		w.open(s, l);
		for (int i = 0;; ++i) {
			const double prob = get_prob(w.state, l);
			if (prob < minprob) { minprob = prob; lend = i; rend = l + i - 1; }
			if (i + 1 + l > len) break;
			w.shift(s[i], s[i + l]);
		}
	}
	leftend += lend;
	rightend -= len - rend - 1;
}

// s_SegSeq (:2008-2094). out receives the segments in the reference's list order (newest first); of the segments found in the part
// that a trim cut off on the left only the list head survives there (`leftsegs->next = *segs`, :2071-2075) -- kept as it is.
inline void seg_seq(const int8_t* s, int len, int offset, std::vector<Range>& out)
{
	if (WINDOW > len) return;
	const int first = DOWNSET, last = len - UPSET;
	std::vector<double> H((size_t)len, -1.0);
	{
		Window w;
		w.open(s, WINDOW);
		for (int i = first; i <= last; ++i) {
			if (w.bogus <= MAX_BOGUS) H[(size_t)i] = window_entropy(w.state);
			if (i < last) w.shift(s[i - first], s[i - first + WINDOW]);
		}
	}
	int lowlim = first;
	for (int i = first; i <= last; ++i) {
		if (!(H[(size_t)i] <= LOCUT && H[(size_t)i] != -1.0)) continue;
		int loi = i, hii = i;
		for (; loi >= lowlim; --loi) if (H[(size_t)loi] == -1.0 || H[(size_t)loi] > HICUT) break;      // s_FindLow
		++loi;
		for (; hii <= last; ++hii) if (H[(size_t)hii] == -1.0 || H[(size_t)hii] > HICUT) break;        // s_FindHigh
		--hii;
		int leftend = loi - DOWNSET, rightend = hii + UPSET - 1;
		trim(s + leftend, rightend - leftend + 1, leftend, rightend);
		if (i + UPSET - 1 < leftend) {                     // the trigger window lies in what the trim cut off on the left
			const int lend = loi - DOWNSET, rend = leftend - 1;
			std::vector<Range> left;
			seg_seq(s + lend, rend - lend + 1, offset + lend, left);
			if (!left.empty()) out.insert(out.begin(), left.front());
		}
		out.insert(out.begin(), Range{ leftend + offset, rightend + offset });
		i = std::min(hii, rightend + DOWNSET);
		lowlim = i + 1;
	}
}

// the segments of a sequence, ascending as the reference hands them to the masking (s_SegsToBlastSeqLoc reverses the list)
inline std::vector<Range> segments(const int8_t* s, int len)
{
	std::vector<Range> r;
	seg_seq(s, len, 0, r);
	std::reverse(r.begin(), r.end());
	return r;
}

}}  // namespace dmnd::seg
