// oracle/ref_hip_bridge.cpp -- TEST INFRASTRUCTURE ONLY (A/B parity harness, SURVEY.md 8b last paragraph).
//
// Links the GENUINE reference (compiled in place from /root/reference by oracle/Makefile) into
// oracle/_ref/diamond_hip with GNU ld --wrap on DP::BandedSwipe::swipe (dp/dp.h:287), and answers
// every such call through OUR C ABI (include/diamond_hip.h, libdiamond_hip.so, dlopen'ed at first
// use): the reference's own CLI, seeding, chaining, culling and output code then run on top of the
// MI355X banded Smith-Waterman, so `diamond_hip blastp ...` must produce byte-identical output to
// `diamond blastp ...`. This is the cgo/JNI-style binding a maintainer would add, written as a
// link-time shim so that no reference source is modified or copied.
//
// Hsp construction mirrors what the reference's traceback() overloads fill
// (dp/swipe/banded_swipe.h:41-183); work items of the statistics-without-traceback bins (3-5) are
// routed to DMND_SWIPE_STATS when DMND_BRIDGE_STATS=1, otherwise left to the reference's own kernel.
//
// Second seam (DMND_BRIDGE_SEED=1): Search::search_shape (search/search.h:80, search/stage0.cpp:219-228), the dispatch point
// of the seed stage. The reference calls it once per shape; dmnd_seed_search runs ALL shapes and index chunks of the block
// pair in one call (the left-most filter and the seed masks couple the shapes), so the call for shape 0 uploads both blocks,
// runs the whole seed stage and pushes every Search::Hit through the reference's own HitBuffer::Writer, and the calls for the
// other shapes return at once. Everything the bridge does not cover (iterated / linclust / global-ranking / target-indexed /
// frequency-masked searches, sensitivities without a preset) goes to the reference's own implementation.
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>
#include <list>
#include "dp/dp.h"
#include "stats/score_matrix.h"
#include "basic/config.h"
#include "basic/shape_config.h"
#include "run/config.h"
#include "search/search.h"
#include "search/hit_buffer.h"
#include "data/block/block.h"
#include "data/seed_set.h"
#include "masking/masking.h"
#include "diamond_hip.h"

namespace {

struct Api {
	void* h = nullptr;
	decltype(&dmnd_create) create;
	decltype(&dmnd_default_params) default_params;
	decltype(&dmnd_banded_swipe_host) swipe_host;
	decltype(&dmnd_last_error) last_error;
	decltype(&dmnd_set_db_letters) set_db_letters;
	decltype(&dmnd_upload_block) upload_block;
	decltype(&dmnd_seed_params_preset) seed_params_preset;
	decltype(&dmnd_seed_params_set_index_chunks) set_index_chunks;
	decltype(&dmnd_seed_params_set_query_indexed) set_query_indexed;
	decltype(&dmnd_seed_search) seed_search;
	decltype(&dmnd_seed_hits) seed_hits;
	decltype(&dmnd_set_query_contexts) set_query_contexts;
	decltype(&dmnd_set_motif_table) set_motif_table;
	decltype(&dmnd_soft_mask_block) soft_mask_block;
	dmnd_params params;
	dmnd_ctx* ctx = nullptr;
	std::mutex mtx;       // one ctx, calls serialised (the reference calls swipe from many align_worker threads)
};

Api& api()
{
	static Api a;
	static std::once_flag once;
	std::call_once(once, [] {
		const char* path = getenv("DMND_HIP_LIB");
		a.h = dlopen(path ? path : "libdiamond_hip.so", RTLD_NOW | RTLD_LOCAL);
		if (!a.h) throw std::runtime_error(std::string("ref_hip_bridge: cannot load libdiamond_hip.so: ") + dlerror());
		a.create = (decltype(a.create))dlsym(a.h, "dmnd_create");
		a.default_params = (decltype(a.default_params))dlsym(a.h, "dmnd_default_params");
		a.swipe_host = (decltype(a.swipe_host))dlsym(a.h, "dmnd_banded_swipe_host");
		a.last_error = (decltype(a.last_error))dlsym(a.h, "dmnd_last_error");
		a.set_db_letters = (decltype(a.set_db_letters))dlsym(a.h, "dmnd_set_db_letters");
		a.upload_block = (decltype(a.upload_block))dlsym(a.h, "dmnd_upload_block");
		a.seed_params_preset = (decltype(a.seed_params_preset))dlsym(a.h, "dmnd_seed_params_preset");
		a.set_index_chunks = (decltype(a.set_index_chunks))dlsym(a.h, "dmnd_seed_params_set_index_chunks");
		a.set_query_indexed = (decltype(a.set_query_indexed))dlsym(a.h, "dmnd_seed_params_set_query_indexed");
		a.seed_search = (decltype(a.seed_search))dlsym(a.h, "dmnd_seed_search");
		a.seed_hits = (decltype(a.seed_hits))dlsym(a.h, "dmnd_seed_hits");
		a.set_query_contexts = (decltype(a.set_query_contexts))dlsym(a.h, "dmnd_set_query_contexts");
		a.set_motif_table = (decltype(a.set_motif_table))dlsym(a.h, "dmnd_set_motif_table");
		a.soft_mask_block = (decltype(a.soft_mask_block))dlsym(a.h, "dmnd_soft_mask_block");
		dmnd_params& p = a.params;
		a.default_params(&p);
		memcpy(p.matrix8, score_matrix.matrix8(), 32 * 32);          // the reference's globals cross the seam
		p.gap_open = score_matrix.gap_open();
		p.gap_extend = score_matrix.gap_extend();
		p.db_letters = (double)score_matrix.db_letters();
		p.max_evalue = config.max_evalue;
		a.ctx = a.create(-1, &p);
		if (!a.ctx) throw std::runtime_error(std::string("ref_hip_bridge: dmnd_create failed: ") + a.last_error());
	});
	return a;
}

struct Item { int bin; const DpTarget* t; };

}

std::list<Hsp> real_swipe(const DP::Targets& targets, DP::Params& params) asm("__real__ZN2DP11BandedSwipe5swipeB5cxx11ERKSt5arrayINS_9TargetVecELm6EERNS_6ParamsE");
std::list<Hsp> wrap_swipe(const DP::Targets& targets, DP::Params& params) asm("__wrap__ZN2DP11BandedSwipe5swipeB5cxx11ERKSt5arrayINS_9TargetVecELm6EERNS_6ParamsE");

std::list<Hsp> wrap_swipe(const DP::Targets& targets, DP::Params& p)
{
	// anything outside the path this back end covers goes to the reference's own kernel
	const bool plain = !flag_any(p.flags, DP::Flags::FULL_MATRIX | DP::Flags::SEMI_GLOBAL) && !p.reverse_targets;
	std::vector<Item> gpu;
	DP::Targets rest;
	bool have_rest = false;
	static const bool bridge_stats = getenv("DMND_BRIDGE_STATS") != nullptr;
	for (int bin = 0; bin < DP::BINS; ++bin)
		for (const DpTarget& t : targets[bin]) {
			const bool stats_bin = p.v != HspValues::NONE && bin >= DP::SCORE_BINS;
			if (plain && t.carry_over.i1 == 0 && (!stats_bin || bridge_stats))
				gpu.push_back(Item{ bin, &t });
			else {
				rest[bin].push_back(t);
				have_rest = true;
			}
		}
	std::list<Hsp> out;
	if (have_rest)
		out = real_swipe(rest, p);
	if (gpu.empty())
		return out;

	Api& a = api();
	const int qlen = p.query.length();
	// one ABI call per mode present in this swipe() call
	for (int pass = 0; pass < 2; ++pass) {
		std::vector<const Item*> sel;
		for (const Item& it : gpu) {
			const bool stats_bin = p.v != HspValues::NONE && it.bin >= DP::SCORE_BINS;
			if ((pass == 1) == stats_bin) sel.push_back(&it);
		}
		if (sel.empty()) continue;
		// HspValues::NONE runs as COORDS so that even the unused end coordinate the reference derives from
		// max_col with a DummyRowCounter (banded_swipe.h:101-103, max_band_row = 0) can be reproduced
		const int mode = p.v == HspValues::NONE ? DMND_SWIPE_COORDS : (pass == 1 ? DMND_SWIPE_STATS : DMND_SWIPE_TRACEBACK);
		std::vector<dmnd_host_target> ht(sel.size());
		int64_t cap = 16;
		for (size_t k = 0; k < sel.size(); ++k) {
			const DpTarget& t = *sel[k]->t;
			ht[k] = dmnd_host_target{ (const int8_t*)t.seq.data(), t.seq.length(), t.d_begin, t.d_end, t.adjusted_matrix() ? t.matrix->scores.data() : nullptr };
			cap += (int64_t)qlen + t.seq.length() + 2;
		}
		std::vector<dmnd_hsp> res(sel.size());
		std::vector<uint8_t> tr(mode == DMND_SWIPE_TRACEBACK ? (size_t)cap : 1);
		int64_t used = 0;
		{
			std::lock_guard<std::mutex> lock(a.mtx);
			a.set_db_letters(a.ctx, (double)score_matrix.db_letters());
			const int rc = a.swipe_host(a.ctx, (const int8_t*)p.query.data(), qlen, p.composition_bias, ht.data(), (int64_t)ht.size(),
				mode, (uint32_t)p.v, res.data(), mode == DMND_SWIPE_TRACEBACK ? tr.data() : nullptr, cap, &used);
			if (rc != DMND_OK)
				throw std::runtime_error(std::string("dmnd_banded_swipe_host: ") + a.last_error());
		}
		for (size_t k = 0; k < sel.size(); ++k) {
			const DpTarget& t = *sel[k]->t;
			const dmnd_hsp& r = res[k];
			const int score = r.score * config.cbs_matrix_scale;
			if (score <= 0) continue;
			const double evalue = score_matrix.evalue(score, qlen, t.true_target_len);
			if (!score_matrix.report_cutoff(score, evalue)) continue;                       // banded_swipe.h:334-336
			Hsp h(mode == DMND_SWIPE_TRACEBACK);
			h.swipe_target = t.target_idx;
			h.swipe_bin = sel[k]->bin;
			h.score = score;
			h.evalue = evalue;
			h.bit_score = score_matrix.bitscore(h.score);
			h.corrected_bit_score = score_matrix.bitscore_corrected(h.score, qlen, t.true_target_len);
			h.frame = p.frame.index();
			h.matrix = t.matrix;
			h.d_begin = t.d_begin;
			h.d_end = t.d_end;
			h.target_seq = t.seq;
			if (mode == DMND_SWIPE_COORDS) {
				// Matrix<Cell> traceback with DummyRowCounter: i1_ = i0 + max_col + 0 + 1, j1_ = j0 + max_col + 1
				const int i1 = std::max(t.d_end - 1, 0), band = t.d_end - t.d_begin, i0 = i1 + 1 - band, j0 = i1 - (t.d_end - 1);
				const int max_col = (r.s_end - 1) - j0;
				h.query_range.end_ = i0 + max_col + 1;
				h.subject_range.end_ = j0 + max_col + 1;
			}
			else {
				h.query_range = Interval(r.q_begin, r.q_end);
				h.subject_range = Interval(r.s_begin, r.s_end);
				h.length = r.length; h.identities = r.identities; h.mismatches = r.mismatches; h.positives = r.positives;
				h.gap_openings = r.gap_openings; h.gaps = r.gaps;
				if (mode == DMND_SWIPE_TRACEBACK) {
					const uint8_t* c = tr.data() + r.transcript_off;
					h.transcript.reserve((size_t)r.transcript_len + 1);
					for (int x = 0; x < r.transcript_len; ++x) {
						const EditOperation op = (EditOperation)(c[x] >> 6);
						if (op == op_match || op == op_insertion) h.transcript.push_back(op, (unsigned)(c[x] & 63));
						else h.transcript.push_back(op, (Letter)(c[x] & 63));
					}
					h.transcript.push_terminator();
				}
				h.approx_id = h.approx_id_percent(p.query, t.seq);
			}
			h.query_source_range = TranslatedPosition::absolute_interval(TranslatedPosition(h.query_range.begin_, p.frame),
				TranslatedPosition(h.query_range.end_, p.frame), p.query_source_len);
			h.subject_source_range = h.subject_range;
			out.push_back(std::move(h));
		}
	}
	return out;
}


// ---- seam 2: Search::search_shape -----------------------------------------------------------------------------------------------
void real_search_shape(unsigned sid, int query_block, unsigned query_iteration, char* query_buffer, char* ref_buffer, Search::Config& cfg, const HashedSeedSet* target_seeds)
	asm("__real__ZN6Search12search_shapeEjijPcS0_RNS_6ConfigEPK13HashedSeedSet");
void wrap_search_shape(unsigned sid, int query_block, unsigned query_iteration, char* query_buffer, char* ref_buffer, Search::Config& cfg, const HashedSeedSet* target_seeds)
	asm("__wrap__ZN6Search12search_shapeEjijPcS0_RNS_6ConfigEPK13HashedSeedSet");

namespace {

int preset_of(Sensitivity s)
{
	switch (s) {
	case Sensitivity::FAST: return DMND_SENS_FAST;
	case Sensitivity::DEFAULT: return DMND_SENS_DEFAULT;
	case Sensitivity::MID_SENSITIVE: return DMND_SENS_MID_SENSITIVE;
	case Sensitivity::SENSITIVE: return DMND_SENS_SENSITIVE;
	case Sensitivity::MORE_SENSITIVE: return DMND_SENS_MORE_SENSITIVE;
	case Sensitivity::VERY_SENSITIVE: return DMND_SENS_VERY_SENSITIVE;
	case Sensitivity::ULTRA_SENSITIVE: return DMND_SENS_ULTRA_SENSITIVE;
	default: return -1;
	}
}

// the seed configuration our library would run, checked against what the reference is about to run (its `shapes` and Search::Config)
bool seed_params_for(Api& a, const Search::Config& cfg, dmnd_seed_params& sp)
{
	if (cfg.sensitivity.size() != 1 || cfg.sensitivity[0].linearize) return false;
	const int preset = preset_of(cfg.sensitivity[0].sensitivity);
	if (preset < 0) return false;
	a.params.db_letters = (double)score_matrix.db_letters();
	if (a.seed_params_preset(&sp, preset, config.threads_, &a.params, nullptr) != DMND_OK) return false;
	if (a.set_index_chunks(&sp, (int)cfg.index_chunks, config.threads_) != DMND_OK) return false;
	if (cfg.seed_encoding == SeedEncoding::HASHED && a.set_query_indexed(&sp, config.threads_) != DMND_OK) return false;
	if (cfg.seed_encoding != SeedEncoding::HASHED && cfg.seed_encoding != SeedEncoding::SPACED_FACTOR) return false;
	sp.hamming_filter_id = (int32_t)cfg.hamming_filter_id;
	sp.seed_complexity_cut = cfg.seed_complexity_cut;
	sp.query_translated = align_mode.query_translated ? 1 : 0;
	if (sp.n_shapes != (int)shapes.count() || sp.seedp_bits != cfg.seedp_bits || sp.index_chunks != (int)cfg.index_chunks) return false;
	for (int i = 0; i < sp.n_shapes; ++i)
		if (sp.shape_len[i] != (int)shapes[i].length_ || sp.shape_weight[i] != (int)shapes[i].weight_ || sp.shape_mask[i] != shapes[i].mask_) return false;
	return true;
}

bool seed_bridge_on() { static const bool on = getenv("DMND_BRIDGE_SEED") != nullptr; return on; }

}

void wrap_search_shape(unsigned sid, int query_block, unsigned query_iteration, char* query_buffer, char* ref_buffer, Search::Config& cfg, const HashedSeedSet* target_seeds)
{
	static bool delivered = false;                 // the hits of all shapes of the current block pair went out with shape 0
	const bool plain = seed_bridge_on() && !target_seeds && !config.global_ranking_targets && !config.freq_masking && !config.lin_stage1_query && !cfg.lin_stage1_target
		&& !Search::keep_target_id(cfg) && cfg.minimizer_window == 0 && cfg.sketch_size == 0 && !config.swipe_all && !cfg.self && !cfg.query_skip
		&& cfg.min_length_ratio == 0.0 && !config.trace_pt_membuf;
	if (sid == 0) delivered = false;
	if (sid > 0 && delivered) return;
	dmnd_seed_params sp;
	if (!plain || sid != 0) { real_search_shape(sid, query_block, query_iteration, query_buffer, ref_buffer, cfg, target_seeds); return; }
	Api& a = api();
	std::lock_guard<std::mutex> lock(a.mtx);
	if (!seed_params_for(a, cfg, sp)) {
		static bool told = false;
		if (!told) { fprintf(stderr, "ref_hip_bridge: seed configuration outside the bridge, Search::search_shape runs in the reference\n"); told = true; }
		real_search_shape(sid, query_block, query_iteration, query_buffer, ref_buffer, cfg, target_seeds);
		return;
	}
	auto chk = [&](int rc, const char* what) { if (rc != DMND_OK) throw std::runtime_error(std::string(what) + ": " + a.last_error()); };
	a.set_db_letters(a.ctx, (double)score_matrix.db_letters());
	chk(a.set_query_contexts(a.ctx, (int)align_mode.query_contexts), "dmnd_set_query_contexts");
	// both blocks as they stand (tantan masking has run where the reference runs it before the seed stage): SequenceSet::data_ / limits_
	SequenceSet& qs = cfg.query->seqs(), &ts = cfg.target->seqs();
	std::vector<int64_t> ql((size_t)qs.size() + 1), tl((size_t)ts.size() + 1);
	for (int64_t i = 0; i <= (int64_t)qs.size(); ++i) ql[(size_t)i] = (int64_t)qs.position(i, 0);
	for (int64_t i = 0; i <= (int64_t)ts.size(); ++i) tl[(size_t)i] = (int64_t)ts.position(i, 0);
	chk(a.upload_block(a.ctx, DMND_QUERY, (const int8_t*)qs.data(0), qs.raw_len() + 256, ql.data(), (int64_t)qs.size()), "dmnd_upload_block(query)");
	chk(a.upload_block(a.ctx, DMND_TARGET, (const int8_t*)ts.data(0), ts.raw_len() + 256, tl.data(), (int64_t)ts.size()), "dmnd_upload_block(target)");
	if (cfg.soft_masking != MaskingAlgo::NONE) {       // motif soft masking during seed enumeration (enum_seeds.h:255-260, masking.cpp:110-131)
		static bool table_set = false;
		if (!table_set) {
			std::vector<uint64_t> codes;
			for (const auto& k : motif_table) codes.push_back(k.code);
			std::sort(codes.begin(), codes.end());
			chk(a.set_motif_table(codes.data(), (int64_t)codes.size()), "dmnd_set_motif_table");
			table_set = true;
		}
		chk(a.soft_mask_block(a.ctx, DMND_QUERY, nullptr), "dmnd_soft_mask_block(query)");
		if (cfg.seed_encoding != SeedEncoding::HASHED) chk(a.soft_mask_block(a.ctx, DMND_TARGET, nullptr), "dmnd_soft_mask_block(target)");
	}
	int64_t n_hits = 0;
	chk(a.seed_search(a.ctx, &sp, &n_hits), "dmnd_seed_search");
	std::vector<dmnd_seed_hit> hits((size_t)n_hits);
	chk(a.seed_hits(a.ctx, hits.data(), n_hits), "dmnd_seed_hits");
	{
		Search::HitBuffer::Writer w(*cfg.seed_hit_buf, 0);      // flushes in its destructor, as a search worker's does (stage0.cpp:84-97)
		for (const dmnd_seed_hit& h : hits) {
			w.new_query(h.query, (Loc)h.seed_offset);
			w.write(h.query, PackedLoc((uint64_t)h.subject), (uint16_t)h.score);
		}
	}
	delivered = true;
}
