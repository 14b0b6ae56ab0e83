// oracle/ref_tap.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Known-answer tap for the reference's own dispatch seam. Linked into oracle/_ref/diamond_tap with
// GNU ld --wrap on DP::BandedSwipe::swipe (dp/dp.h:287; dispatcher emitted at
// dp/swipe/swipe_wrapper.cpp:487), so every call the genuine reference makes is forwarded
// unchanged to __real_ and its inputs (query, Hauser bias, every DpTarget of all 6 bins) and the
// returned std::list<Hsp> are appended to the file named by $DIAMOND_TAP_FILE.
// No reference source is modified or copied; this file only *uses* the reference's headers.
//
// Record layout (little endian, all int32 unless noted). First record of the file:
//   magic 'MTX1' | gap_open | gap_extend | f64 lambda | f64 ln_k | f64 db_letters | f64 max_evalue
//   | matrix8[32*32] (int8, score_matrix.matrix8(), stats/score_matrix.h:69)
// then one record per swipe() call:
//   magic 'SWP1' | flags | hsp_values | frame | query_source_len | qlen | has_cbs
//   | query[qlen] (int8, raw letters incl. mask bit) | cbs[qlen] (int8, if has_cbs)
//   | n_targets | n_targets x { bin, target_idx, d_begin, d_end, cols, true_target_len, tlen, seq[tlen] }
//   | n_hsps   | n_hsps x { swipe_target, swipe_bin, score, frame, d_begin, d_end,
//                           q_begin, q_end, s_begin, s_end, length, identities, mismatches,
//                           positives, gap_openings, gaps, backtraced, f64 evalue, f64 bit_score,
//                           n_transcript, transcript[n_transcript] (uint8 PackedOperation codes) }
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <atomic>
#include <vector>
#include <list>
#include "dp/dp.h"
#include "stats/score_matrix.h"

namespace {
std::mutex tap_mtx;
FILE* tap_file() {
	static FILE* f = nullptr;
	static bool init = false;
	if (!init) {
		init = true;
		const char* p = getenv("DIAMOND_TAP_FILE");
		if (p) f = fopen(p, "wb");
	}
	return f;
}
long tap_limit() {
	static long lim = -2;
	if (lim == -2) {
		const char* p = getenv("DIAMOND_TAP_MAX_CALLS");
		lim = p ? atol(p) : -1;
	}
	return lim;
}
// Cell counting mode ($DIAMOND_TAP_CELLS=file): no record dump, only the number of swipe() calls, DpTargets and
// DP cells (sum of DpTarget::cells(), dp/dp.h:121-124 = the reference's NET_DP_CELLS definition; its own
// -DDP_STAT counters do not compile in this revision) -- written at exit. Used for the GCUPS cpu_baseline.
std::atomic<long long> g_cells(0), g_targets(0), g_calls(0);
void write_cells() {
	const char* p = getenv("DIAMOND_TAP_CELLS");
	if (!p) return;
	if (FILE* f = fopen(p, "w")) {
		fprintf(f, "{\"calls\": %lld, \"targets\": %lld, \"cells\": %lld}\n", g_calls.load(), g_targets.load(), g_cells.load());
		fclose(f);
	}
}
struct Buf {
	std::vector<char> d;
	void i32(int32_t v) { d.insert(d.end(), (char*)&v, (char*)&v + 4); }
	void f64(double v) { d.insert(d.end(), (char*)&v, (char*)&v + 8); }
	void bytes(const void* p, size_t n) { d.insert(d.end(), (const char*)p, (const char*)p + n); }
};
}

// The wrapped symbol is a mangled C++ name, so bind the __real_/__wrap_ linker names with asm labels.
std::list<Hsp> real_swipe(const DP::Targets& targets, DP::Params& params) asm("__real__ZN2DP11BandedSwipe5swipeB5cxx11ERKSt5arrayINS_9TargetVecELm6EERNS_6ParamsE");
std::list<Hsp> wrap_swipe(const DP::Targets& targets, DP::Params& params) asm("__wrap__ZN2DP11BandedSwipe5swipeB5cxx11ERKSt5arrayINS_9TargetVecELm6EERNS_6ParamsE");

std::list<Hsp> wrap_swipe(const DP::Targets& targets, DP::Params& params)
{
	static long calls = 0;
	static const bool count_cells = getenv("DIAMOND_TAP_CELLS") != nullptr;
	if (count_cells) {
		static std::once_flag once;
		std::call_once(once, [] { atexit(write_cells); });
		long long c = 0, n = 0;
		for (int bin = 0; bin < DP::BINS; ++bin)
			for (const DpTarget& t : targets[bin]) { c += t.cells(params.flags, params.query.length()); ++n; }
		g_cells += c; g_targets += n; ++g_calls;
	}
	Buf b;
	FILE* f = tap_file();
	if (f) {
		b.i32(0x31505753);
		b.i32((int32_t)params.flags);
		b.i32((int32_t)params.v);
		b.i32(params.frame.index());
		b.i32(params.query_source_len);
		const int qlen = params.query.length();
		b.i32(qlen);
		b.i32(params.composition_bias ? 1 : 0);
		b.bytes(params.query.data(), qlen);
		if (params.composition_bias)
			b.bytes(params.composition_bias, qlen);
		int32_t n = 0;
		for (int bin = 0; bin < DP::BINS; ++bin)
			n += (int32_t)targets[bin].size();
		b.i32(n);
		for (int bin = 0; bin < DP::BINS; ++bin)
			for (const DpTarget& t : targets[bin]) {
				b.i32(bin);
				b.i32((int32_t)t.target_idx);
				b.i32(t.d_begin);
				b.i32(t.d_end);
				b.i32(t.cols);
				b.i32(t.true_target_len);
				b.i32(t.seq.length());
				b.bytes(t.seq.data(), t.seq.length());
			}
	}
	std::list<Hsp> out = real_swipe(targets, params);
	if (f) {
		b.i32((int32_t)out.size());
		for (const Hsp& h : out) {
			b.i32(h.swipe_target);
			b.i32(h.swipe_bin);
			b.i32(h.score);
			b.i32(h.frame);
			b.i32(h.d_begin);
			b.i32(h.d_end);
			b.i32(h.query_range.begin_);
			b.i32(h.query_range.end_);
			b.i32(h.subject_range.begin_);
			b.i32(h.subject_range.end_);
			b.i32(h.length);
			b.i32(h.identities);
			b.i32(h.mismatches);
			b.i32(h.positives);
			b.i32(h.gap_openings);
			b.i32(h.gaps);
			b.i32(h.backtraced ? 1 : 0);
			b.f64(h.evalue);
			b.f64(h.bit_score);
			const auto& tr = h.transcript.data();
			b.i32((int32_t)tr.size());
			for (const PackedOperation& op : tr) {
				const uint8_t c = op.code;
				b.bytes(&c, 1);
			}
		}
		std::lock_guard<std::mutex> lock(tap_mtx);
		if (calls == 0) {
			Buf h;
			h.i32(0x3158544d);
			h.i32(score_matrix.gap_open());
			h.i32(score_matrix.gap_extend());
			h.f64(score_matrix.lambda());
			h.f64(score_matrix.ln_k());
			h.f64(score_matrix.db_letters());
			h.f64(config.max_evalue);
			h.bytes(score_matrix.matrix8(), 32 * 32);
			fwrite(h.d.data(), 1, h.d.size(), f);
		}
		const long lim = tap_limit();
		if (lim < 0 || calls < lim) {
			fwrite(b.d.data(), 1, b.d.size(), f);
			fflush(f);
		}
		++calls;
	}
	return out;
}
