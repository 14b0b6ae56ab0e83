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
//     ($DIAMOND_TAP_MATRICES set: magic 'SWP2' and every target is followed by has_matrix | int8 scores[32*26] if has_matrix:
//      the target's composition-adjusted matrix, DpTarget::matrix, --comp-based-stats 2..5)
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
#include "align/extend.h"
#include "search/hit.h"
#include "run/config.h"
#include "data/block/block.h"
#include "basic/shape_config.h"
#include "basic/reduction.h"

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
	static const bool with_matrices = getenv("DIAMOND_TAP_MATRICES") != nullptr;      // 'SWP2': every DpTarget followed by its adjusted matrix
	if (f) {
		b.i32(with_matrices ? 0x32505753 : 0x31505753);
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
				if (with_matrices) {
					b.i32(t.adjusted_matrix() ? 1 : 0);
					if (t.adjusted_matrix()) b.bytes(t.matrix->scores.data(), 32 * AMINO_ACID_COUNT);
				}
			}
	}
	std::list<Hsp> out;
	try { out = real_swipe(targets, params); }
	catch (...) {
		// a call the reference itself refuses (e.g. "Traceback with adjusted matrix not supported", full_swipe.h:102): its inputs
		// are still written, with -1 for the number of HSPs
		if (f) {
			b.i32(-1);
			std::lock_guard<std::mutex> lock(tap_mtx);
			fwrite(b.d.data(), 1, b.d.size(), f);
			fflush(f);
		}
		throw;
	}
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


// ---- second seam: Extension::extend(query_id, Hit* begin, Hit* end, cfg, ...) (align/extend.cpp:346), called per
// query from align_worker (align/align.cpp:157). Its input is the query's stage-2 seed hits (the output of the whole
// seed stage, all shapes and index chunks), its output the query's final Match list. $DIAMOND_TAP_EXT=file:
//   first record 'BLK1': seed-stage configuration + both sequence blocks (letters as they are at extension time):
//     seedp_bits index_chunks hamming_filter_id n_shapes | per shape: length weight mask positions[weight]
//     | reduction map[32] (int32) | f64 seed_complexity_cut | f64 ungapped_evalue | f64 gapped_filter_evalue
//     | query_contexts | per block (query, target): n_seqs, i64 raw_len, data[raw_len], i64 limits[n_seqs+1]
//   then per call 'EXT1': query_id n_hits | n_hits x { u32 query, i64 subject, i32 seed_offset, i32 score }
//     | n_matches x { target_block_id filter_score f64 filter_evalue ungapped_score n_hsp | hsp records as above }
namespace {
FILE* ext_file() {
	static FILE* f = nullptr;
	static bool init = false;
	if (!init) {
		init = true;
		const char* p = getenv("DIAMOND_TAP_EXT");
		if (p) f = fopen(p, "wb");
	}
	return f;
}
void i64(Buf& b, int64_t v) { b.bytes(&v, 8); }
void dump_block(Buf& b, const SequenceSet& s) {
	const int64_t n = s.size();
	b.i32((int32_t)n);
	const int64_t raw = s.position(n - 1, 0) + s.length(n - 1) + 1 + 256;     // last delimiter + perimeter padding
	i64(b, raw);
	b.bytes(s.data(0), (size_t)raw);
	for (int64_t i = 0; i < n; ++i) i64(b, s.position(i, 0));
	i64(b, s.position(n - 1, 0) + s.length(n - 1) + 1);
}
void dump_hsp(Buf& b, const Hsp& h) {
	b.i32(h.swipe_target); b.i32(h.swipe_bin); b.i32(h.score); b.i32(h.frame); b.i32(h.d_begin); b.i32(h.d_end);
	b.i32(h.query_range.begin_); b.i32(h.query_range.end_); b.i32(h.subject_range.begin_); b.i32(h.subject_range.end_);
	b.i32(h.length); b.i32(h.identities); b.i32(h.mismatches); b.i32(h.positives); b.i32(h.gap_openings); b.i32(h.gaps);
	b.i32(h.backtraced ? 1 : 0); b.f64(h.evalue); b.f64(h.bit_score);
	const auto& tr = h.transcript.data();
	b.i32((int32_t)tr.size());
	for (const PackedOperation& op : tr) { const uint8_t c = op.code; b.bytes(&c, 1); }
}
}

std::vector<Extension::Match> real_extend(BlockId query_id, Search::Hit* begin, Search::Hit* end, const Search::Config& cfg, Statistics& stat, DP::Flags flags, std::pmr::monotonic_buffer_resource& pool)
	asm("__real__ZN9Extension6extendEjPN6Search3HitES2_RKNS0_6ConfigER10StatisticsN2DP5FlagsERNSt3pmr25monotonic_buffer_resourceE");
std::vector<Extension::Match> wrap_extend(BlockId query_id, Search::Hit* begin, Search::Hit* end, const Search::Config& cfg, Statistics& stat, DP::Flags flags, std::pmr::monotonic_buffer_resource& pool)
	asm("__wrap__ZN9Extension6extendEjPN6Search3HitES2_RKNS0_6ConfigER10StatisticsN2DP5FlagsERNSt3pmr25monotonic_buffer_resourceE");

std::vector<Extension::Match> wrap_extend(BlockId query_id, Search::Hit* begin, Search::Hit* end, const Search::Config& cfg, Statistics& stat, DP::Flags flags, std::pmr::monotonic_buffer_resource& pool)
{
	FILE* f = ext_file();
	Buf b;
	if (f) {
		b.i32(0x31545845);
		b.i32((int32_t)query_id);
		b.i32((int32_t)(end - begin));
		for (const Search::Hit* h = begin; h < end; ++h) {
			b.i32((int32_t)h->query_);
			i64(b, (int64_t)(uint64_t)h->subject_);
			b.i32((int32_t)h->seed_offset_);
			b.i32((int32_t)h->score_);
		}
	}
	std::vector<Extension::Match> out = real_extend(query_id, begin, end, cfg, stat, flags, pool);
	if (f) {
		b.i32((int32_t)out.size());
		for (const Extension::Match& m : out) {
			b.i32((int32_t)m.target_block_id); b.i32(m.filter_score); b.f64(m.filter_evalue); b.i32(m.ungapped_score);
			b.i32((int32_t)m.hsp.size());
			for (const Hsp& h : m.hsp) dump_hsp(b, h);
		}
		static std::mutex mtx;
		static bool header = false;
		std::lock_guard<std::mutex> lock(mtx);
		if (!header) {
			header = true;
			Buf h;
			h.i32(0x314b4c42);
			h.i32(cfg.seedp_bits); h.i32((int32_t)cfg.index_chunks); h.i32((int32_t)cfg.hamming_filter_id); h.i32(shapes.count());
			for (int i = 0; i < shapes.count(); ++i) {
				h.i32(shapes[i].length_); h.i32(shapes[i].weight_); h.i32((int32_t)shapes[i].mask_);
				for (int k = 0; k < shapes[i].weight_; ++k) h.i32(shapes[i].positions_[k]);
			}
			for (int i = 0; i < 32; ++i) h.i32((int32_t)Reduction::get_reduction()((size_t)i));
			h.f64(cfg.seed_complexity_cut); h.f64(cfg.ungapped_evalue); h.f64(cfg.gapped_filter_evalue);
			h.i32(align_mode.query_contexts);
			dump_block(h, cfg.query->seqs());
			dump_block(h, cfg.target->seqs());
			fwrite(h.d.data(), 1, h.d.size(), f);
		}
		fwrite(b.d.data(), 1, b.d.size(), f);
		fflush(f);
	}
	return out;
}

// ---- third seam: the gapped filter (align/gapped_filter.cpp:80; called from Extension::extend, align/extend.cpp:206,
// only when gapped_filter_evalue > 0, i.e. --sensitive and above). $DIAMOND_TAP_GF=file, per call one 'GFL1' record:
//   f64 gapped_filter_evalue | f64 gapped_filter_evalue1 | i32 diag_score window gap_open gap_extend
//   | i64 query offset inside the query block | i32 qlen | i32 has_cbs | int8 cbs[qlen]
//   | i32 n_targets x { u32 block_id, i32 cutoff1, i32 cutoff2, i32 n_hits x { i32 i j score frame } }
//   | i32 n_out x u32 surviving block ids
#include "align/target.h"
#include "stats/hauser_correction.h"
#define GF_SYM "_ZN9Extension13gapped_filterEPK8SequencePK16HauserCorrectionN9FlatArrayINS_7SeedHitEmE8IteratorES9_N9__gnu_cxx17__normal_iteratorIPKjSt6vectorIjSaIjEEEER10StatisticsN2DP5FlagsERKN6Search6ConfigE"
using GfResult = std::pair<FlatArray<Extension::SeedHit>, std::vector<uint32_t>>;
GfResult real_gf(const Sequence* query, const HauserCorrection* query_cbs, FlatArray<Extension::SeedHit>::Iterator seed_hits, FlatArray<Extension::SeedHit>::Iterator seed_hits_end,
	std::vector<uint32_t>::const_iterator target_block_ids, Statistics& stat, DP::Flags flags, const Search::Config& params) asm("__real_" GF_SYM);
GfResult wrap_gf(const Sequence* query, const HauserCorrection* query_cbs, FlatArray<Extension::SeedHit>::Iterator seed_hits, FlatArray<Extension::SeedHit>::Iterator seed_hits_end,
	std::vector<uint32_t>::const_iterator target_block_ids, Statistics& stat, DP::Flags flags, const Search::Config& params) asm("__wrap_" GF_SYM);

GfResult wrap_gf(const Sequence* query, const HauserCorrection* query_cbs, FlatArray<Extension::SeedHit>::Iterator seed_hits, FlatArray<Extension::SeedHit>::Iterator seed_hits_end,
	std::vector<uint32_t>::const_iterator target_block_ids, Statistics& stat, DP::Flags flags, const Search::Config& params)
{
	static FILE* f = getenv("DIAMOND_TAP_GF") ? fopen(getenv("DIAMOND_TAP_GF"), "wb") : nullptr;
	Buf b;
	if (f) {
		const int64_t n = seed_hits_end - seed_hits;
		const int qlen = (int)query[0].length();
		b.i32(0x314c4647);
		b.f64(params.gapped_filter_evalue); b.f64(config.gapped_filter_evalue1);
		b.i32(config.gapped_filter_diag_score); b.i32(config.gapped_filter_window); b.i32(score_matrix.gap_open()); b.i32(score_matrix.gap_extend());
		i64(b, (int64_t)(query[0].data() - params.query->seqs().data(0)));
		b.i32(qlen);
		const bool has_cbs = Stats::CBS::hauser(config.comp_based_stats);
		b.i32(has_cbs ? 1 : 0);
		if (has_cbs) b.bytes(query_cbs[0].int8.data(), (size_t)qlen);
		b.i32((int32_t)n);
		for (int64_t t = 0; t < n; ++t) {
			const uint32_t id = target_block_ids[t];
			const int slen = (int)params.target->seqs()[id].length();
			b.i32((int32_t)id); b.i32(params.cutoff_gapped1_new(qlen, slen)); b.i32(params.cutoff_gapped2_new(qlen, slen));
			b.i32((int32_t)(seed_hits.end(t) - seed_hits.begin(t)));
			for (auto h = seed_hits.begin(t); h < seed_hits.end(t); ++h) { b.i32(h->i); b.i32(h->j); b.i32(h->score); b.i32((int32_t)h->frame); }
		}
	}
	GfResult out = real_gf(query, query_cbs, seed_hits, seed_hits_end, target_block_ids, stat, flags, params);
	if (f) {
		b.i32((int32_t)out.second.size());
		for (uint32_t id : out.second) b.i32((int32_t)id);
		static std::mutex mtx;
		std::lock_guard<std::mutex> lock(mtx);
		fwrite(b.d.data(), 1, b.d.size(), f);
		fflush(f);
	}
	return out;
}

// ---- fourth seam: tantan repeat masking (masking/tantan.cpp:112, dispatcher Util::tantan::mask; called per sequence from
// Masking::operator(), masking/masking.cpp:176, for the query and the reference block when --masking is tantan = default).
// $DIAMOND_TAP_TANTAN=file: header 'TANH' + 32x32 float likelihood-ratio matrix + p_repeat p_repeat_end repeat_growth p_mask
// (floats), then per call 'TAN1': len | letters before | letters after (mask_mode as passed).
#include "masking/def.h"
#define TANTAN_SYM "_ZN4Util6tantan4maskEPaiPPKfffffi"
namespace Util { namespace tantan { } }
Mask::Ranges real_tantan(Letter* seq, int len, const float** lr, float p_repeat, float p_repeat_end, float repeat_growth, float p_mask, int mask_mode) asm("__real_" TANTAN_SYM);
Mask::Ranges wrap_tantan(Letter* seq, int len, const float** lr, float p_repeat, float p_repeat_end, float repeat_growth, float p_mask, int mask_mode) asm("__wrap_" TANTAN_SYM);

Mask::Ranges wrap_tantan(Letter* seq, int len, const float** lr, float p_repeat, float p_repeat_end, float repeat_growth, float p_mask, int mask_mode)
{
	static FILE* f = getenv("DIAMOND_TAP_TANTAN") ? fopen(getenv("DIAMOND_TAP_TANTAN"), "wb") : nullptr;
	static std::mutex mtx;
	static bool header = false;
	static std::atomic<int64_t> budget(getenv("DIAMOND_TAP_TANTAN_MAX") ? atoll(getenv("DIAMOND_TAP_TANTAN_MAX")) : (int64_t)1 << 62);
	if (!f || budget.fetch_sub(1) <= 0)
		return real_tantan(seq, len, lr, p_repeat, p_repeat_end, repeat_growth, p_mask, mask_mode);
	Buf b;
	b.i32(0x314e4154); b.i32(len); b.i32(mask_mode);
	b.bytes(seq, (size_t)len);
	Mask::Ranges r = real_tantan(seq, len, lr, p_repeat, p_repeat_end, repeat_growth, p_mask, mask_mode);
	b.bytes(seq, (size_t)len);
	b.i32((int32_t)r.size());
	for (const auto& x : r) { b.i32(x.first); b.i32(x.second); }
	std::lock_guard<std::mutex> lock(mtx);
	if (!header) {
		header = true;
		Buf h;
		h.i32(0x484e4154);
		for (int i = 0; i < 32; ++i) h.bytes(i < AMINO_ACID_COUNT ? lr[i] : lr[0], 32 * sizeof(float));
		const float p[4] = { p_repeat, p_repeat_end, repeat_growth, p_mask };
		h.bytes(p, sizeof p);
		fwrite(h.d.data(), 1, h.d.size(), f);
	}
	fwrite(b.d.data(), 1, b.d.size(), f);
	fflush(f);
	return r;
}

// ---- fifth seam: composition-based matrix adjustment (--comp-based-stats 2..5). Two wrapped symbols, both defined in
// stats/cbs.cpp and called per (query, target) from WorkTarget::WorkTarget (align/ungapped.cpp:44-58):
//   Stats::adjust_matrix (cbs.cpp:94-112)            -> which rule, or eDontAdjustMatrix
//   Stats::TargetMatrix::TargetMatrix (cbs.cpp:114-173) -> the target's 32 x 26 int8 score table
// $DIAMOND_TAP_CBS=file. Header 'CBH1' once: cbs mode | cbs_matrix_scale | f64 ideal_lambda | f64 ungapped_lambda | f64 cbs_angle
//   | f64 joint_probs[400] | f64 background[20] | int8 matrix8[32*32]
// then 'ADJ1': f64 query_comp[20] | query_len | cbs | tlen | target[tlen] | rule
// and  'TMX1': f64 query_comp[20] | query_len | cbs | rule | tlen | target[tlen] | int8 scores[32*26] | score_min | score_max
#include "stats/cbs.h"
#define ADJ_SYM "_ZN5Stats13adjust_matrixERKSt5arrayIdLm20EEijRK8Sequence"
#define TMX_SYM "_ZN5Stats12TargetMatrixC1ERKSt5arrayIdLm20EEijRK8SequenceR10StatisticsRNSt3pmr25monotonic_buffer_resourceENS_17EMatrixAdjustRuleE"
namespace {
std::mutex cbs_mtx;
FILE* cbs_file() {
	static FILE* f = getenv("DIAMOND_TAP_CBS") ? fopen(getenv("DIAMOND_TAP_CBS"), "wb") : nullptr;
	return f;
}
std::atomic<int64_t> cbs_budget(getenv("DIAMOND_TAP_CBS_MAX") ? atoll(getenv("DIAMOND_TAP_CBS_MAX")) : (int64_t)1 << 62);
void cbs_write(const Buf& b) {
	FILE* f = cbs_file();
	std::lock_guard<std::mutex> lock(cbs_mtx);
	static bool header = false;
	if (!header) {
		header = true;
		Buf h;
		h.i32(0x31484243); h.i32((int32_t)config.comp_based_stats); h.i32((int32_t)config.cbs_matrix_scale);
		h.f64(score_matrix.ideal_lambda()); h.f64(score_matrix.ungapped_lambda()); h.f64(Stats::comp_based_stats.angle);
		h.bytes(score_matrix.joint_probs(), 400 * sizeof(double));
		h.bytes(score_matrix.background_freqs(), 20 * sizeof(double));
		h.bytes(score_matrix.matrix8(), 32 * 32);
		fwrite(h.d.data(), 1, h.d.size(), f);
	}
	fwrite(b.d.data(), 1, b.d.size(), f);
	fflush(f);
}
}
Stats::EMatrixAdjustRule real_adjust(const Stats::Composition& query_comp, int query_len, unsigned cbs, const Sequence& target) asm("__real_" ADJ_SYM);
Stats::EMatrixAdjustRule wrap_adjust(const Stats::Composition& query_comp, int query_len, unsigned cbs, const Sequence& target) asm("__wrap_" ADJ_SYM);
Stats::EMatrixAdjustRule wrap_adjust(const Stats::Composition& query_comp, int query_len, unsigned cbs, const Sequence& target)
{
	const Stats::EMatrixAdjustRule r = real_adjust(query_comp, query_len, cbs, target);
	if (cbs_file() && Stats::CBS::matrix_adjust(cbs) && cbs_budget.fetch_sub(1) > 0) {
		Buf b;
		b.i32(0x314a4441);
		b.bytes(query_comp.data(), 20 * sizeof(double));
		b.i32(query_len); b.i32((int32_t)cbs); b.i32((int32_t)target.length());
		b.bytes(target.data(), (size_t)target.length());
		b.i32((int32_t)r);
		cbs_write(b);
	}
	return r;
}
void real_tmx(Stats::TargetMatrix* self, const Stats::Composition& query_comp, int query_len, unsigned cbs, const Sequence& target, Statistics& stats, std::pmr::monotonic_buffer_resource& pool, Stats::EMatrixAdjustRule rule) asm("__real_" TMX_SYM);
void wrap_tmx(Stats::TargetMatrix* self, const Stats::Composition& query_comp, int query_len, unsigned cbs, const Sequence& target, Statistics& stats, std::pmr::monotonic_buffer_resource& pool, Stats::EMatrixAdjustRule rule) asm("__wrap_" TMX_SYM);
void wrap_tmx(Stats::TargetMatrix* self, const Stats::Composition& query_comp, int query_len, unsigned cbs, const Sequence& target, Statistics& stats, std::pmr::monotonic_buffer_resource& pool, Stats::EMatrixAdjustRule rule)
{
	real_tmx(self, query_comp, query_len, cbs, target, stats, pool, rule);
	if (cbs_file() && cbs_budget.fetch_sub(1) > 0) {
		Buf b;
		b.i32(0x31584d54);
		b.bytes(query_comp.data(), 20 * sizeof(double));
		b.i32(query_len); b.i32((int32_t)cbs); b.i32((int32_t)rule); b.i32((int32_t)target.length());
		b.bytes(target.data(), (size_t)target.length());
		b.bytes(self->scores.data(), 32 * AMINO_ACID_COUNT);
		b.i32(self->score_min); b.i32(self->score_max);
		cbs_write(b);
	}
}

// ---- sixth seam: the three-frame banded sweep of frameshift alignment (blastx -F), dispatch point banded_3frame_swipe
// (dp/dp.h:296, dp/swipe/banded_3frame_swipe.cpp:647; called twice per query -- forward and reverse strand -- from
// ExtensionPipeline::BandedSwipe::Pipeline::run_swipe, align/legacy/banded_swipe_pipeline.cpp:157-171).
// $DIAMOND_TAP_3F=file. Header 'F3H1' once: gap_open gap_extend frame_shift | f64 db_letters | f64 max_evalue | int8 matrix8[32*32],
// then per call 'F3S1': strand score_only dna_len | 3 x { len, letters[len] } (the strand's three frames)
//   | n_targets x { target_idx d_begin d_end cols tlen seq[tlen] }   (in the order the caller passed them: the sweep sorts them itself)
//   | n_hsps x { swipe_target score frame q_begin q_end s_begin s_end qs_begin qs_end length identities mismatches positives
//                gap_openings gaps f64 evalue f64 bit_score n_transcript transcript[] }
#define F3_SYM "_Z19banded_3frame_swipeB5cxx11RK18TranslatedSequence6StrandN9__gnu_cxx17__normal_iteratorIP8DpTargetSt6vectorIS5_SaIS5_EEEESA_R6DpStatbb"
std::list<Hsp> real_3f(const TranslatedSequence& query, Strand strand, std::vector<DpTarget>::iterator target_begin, std::vector<DpTarget>::iterator target_end, DpStat& stat, bool score_only, bool parallel) asm("__real_" F3_SYM);
std::list<Hsp> wrap_3f(const TranslatedSequence& query, Strand strand, std::vector<DpTarget>::iterator target_begin, std::vector<DpTarget>::iterator target_end, DpStat& stat, bool score_only, bool parallel) asm("__wrap_" F3_SYM);
std::list<Hsp> wrap_3f(const TranslatedSequence& query, Strand strand, std::vector<DpTarget>::iterator target_begin, std::vector<DpTarget>::iterator target_end, DpStat& stat, bool score_only, bool parallel)
{
	static FILE* f = getenv("DIAMOND_TAP_3F") ? fopen(getenv("DIAMOND_TAP_3F"), "wb") : nullptr;
	static std::mutex mtx;
	static bool header = false;
	static std::atomic<int64_t> budget(getenv("DIAMOND_TAP_3F_MAX") ? atoll(getenv("DIAMOND_TAP_3F_MAX")) : (int64_t)1 << 62);
	if (!f || target_begin == target_end || budget.fetch_sub(1) <= 0)
		return real_3f(query, strand, target_begin, target_end, stat, score_only, parallel);
	Buf b;
	b.i32(0x31533346); b.i32((int32_t)strand); b.i32(score_only ? 1 : 0); b.i32((int32_t)query.source().length());
	Sequence q[3];
	query.get_strand(strand, q);
	for (int k = 0; k < 3; ++k) { b.i32((int32_t)q[k].length()); b.bytes(q[k].data(), (size_t)q[k].length()); }
	b.i32((int32_t)(target_end - target_begin));
	for (auto t = target_begin; t != target_end; ++t) {
		b.i32((int32_t)t->target_idx); b.i32(t->d_begin); b.i32(t->d_end); b.i32(t->cols); b.i32((int32_t)t->seq.length());
		b.bytes(t->seq.data(), (size_t)t->seq.length());
	}
	std::list<Hsp> out = real_3f(query, strand, target_begin, target_end, stat, score_only, parallel);
	b.i32((int32_t)out.size());
	for (const Hsp& h : out) {
		b.i32(h.swipe_target); b.i32(h.score); b.i32(h.frame);
		b.i32(h.query_range.begin_); b.i32(h.query_range.end_); b.i32(h.subject_range.begin_); b.i32(h.subject_range.end_);
		b.i32(h.query_source_range.begin_); b.i32(h.query_source_range.end_);
		b.i32(h.length); b.i32(h.identities); b.i32(h.mismatches); b.i32(h.positives); b.i32(h.gap_openings); b.i32(h.gaps);
		b.f64(h.evalue); b.f64(h.bit_score);
		const auto& tr = h.transcript.data();
		b.i32((int32_t)tr.size());
		for (const PackedOperation& op : tr) { const uint8_t c = op.code; b.bytes(&c, 1); }
	}
	std::lock_guard<std::mutex> lock(mtx);
	if (!header) {
		header = true;
		Buf h;
		h.i32(0x31483346); h.i32(score_matrix.gap_open()); h.i32(score_matrix.gap_extend()); h.i32(score_matrix.frame_shift());
		h.f64(score_matrix.db_letters()); h.f64(config.max_evalue);
		h.bytes(score_matrix.matrix8(), 32 * 32);
		fwrite(h.d.data(), 1, h.d.size(), f);
	}
	fwrite(b.d.data(), 1, b.d.size(), f);
	fflush(f);
	return out;
}
