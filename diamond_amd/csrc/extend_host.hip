// extend_host.hip -- the extension stage above the GPU Smith-Waterman: what Extension::extend does per query
// (/root/reference/src/align/extend.cpp:226-420), re-organised for the GPU as block-wide batches.
// Since round 6 dmnd_extend has TWO halves behind one entry. The default protein search (one query context, one HSP per target,
// -k culling by e-value, Hauser bias or none, banded extension, no transcripts) is planned (plan_kernels.hip) and extended
// (extend_kernels.hip; extend_device.hip drives it) entirely in HBM: the host reads a few counters per ranking iteration, writes
// its own e-value and bit score into the records and leaves them on the device for the join. Every other mode, and the queries the
// device half hands back, take the HOST PATH described next (extend_range) -- same records either way; the two outputs are merged by query.
// The host path: the queries with seed hits go through
//   prelude     :  Hauser bias of the whole query block                              (GPU, bias_kernels.hip; once per call)
//                  gapped filter of every seed hit, --sensitive and above             (GPU, gapped_kernels.hip; once per call)
//   all queries :  load_hits -> x-drop ungapped -> chaining -> band construction      (host; looked up from the device planner's lists when it ran)
//   ONE call    :  round-1 banded swipe over every DpTarget of the ranking chunk, in traceback mode with kept trace rows
//                  (score-only when the caller wants transcripts or the rows exceed the trace budget)      (GPU)
//   all queries :  e-value cutoff, per-target best HSP, ranking-chunk logic, top-k culling                 (host)
//   ONE call    :  round 2 = traceback walk over the kept traces of the surviving targets (or a second sweep with traceback;
//                  statistics kernels for matrices above max_swipe_dp)                                     (GPU)
//   all queries :  final culling -> match records                                                          (host)
// instead of the reference's per-query calls of DP::BandedSwipe::swipe from a thread pool (host work in fixed slices of the query
// range on the context's worker pool; tuning.h: extend_team / extend_split).
// Reference pieces restated here (chaining itself is in chain_graph.h, one source for host and device):
//   HauserCorrection                        src/stats/hauser_correction.cpp:53-109   (host copy: dmnd_extend_plan; GPU: bias_core.h)
//   load_hits                               src/align/load_hits.h:44-127
//   ranking_chunk_size, ranking loop        src/align/extend.cpp:79-119, 289-336
//   ungapped_stage                          src/align/ungapped.cpp:62-126
//   Extension::band, add_dp_targets         src/align/gapped_score.cpp:41-180
//   DP::BandedSwipe::bin                    src/dp/swipe/swipe_wrapper.cpp:75-102
//   Target::add_hit / inner_culling, culling, output_range   src/align/target.h:97-113, culling.cpp:37-113,189-203
//   round-2 add_dp_targets / align          src/align/gapped_final.cpp:66-160
//   (the join of reference blocks, the blast tab line and the six-frame translation lived here until round 6:
//    join_blocks.hip, format_tab.hip, translate.hip)
//   composition-based matrix adjustment     src/align/ungapped.cpp:44-58 -> cbs_adjust.cpp (--comp-based-stats 2 - 5)
// Scope: blastp and blastx (1 or 6 query contexts), any max_hsps, comp-based-stats 0 - 5, gapped filter, ranking chunks,
// id / coverage filters; no frameshift alignment.
#include <algorithm>
#include <array>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <vector>
#include "ctx.h"
#include "chain_graph.h"
#include "host_pool.h"
#include "bias_kernels.h"
#include "cbs_adjust.h"
#include "read_coverage.h"
#include "plan_kernels.h"
#include "extend_kernels.h"
#include "match_order.h"
#include "extend_device.h"
#include <unordered_map>

namespace dmnd {
static thread_local int tls_pool = -1;
static double g_extend_t0 = 0;           // DMND_TRACE=2: start of the current dmnd_extend (absolute timeline of the runners)
void set_thread_pool(int k) { tls_pool = k; }
int thread_pool() { return tls_pool; }
WorkerPool& pool()
{
	static WorkerPool pools[MAX_POOLS + 1];
	return pools[tls_pool >= 0 && tls_pool < MAX_POOLS ? tls_pool + 1 : 0];
}
}

using namespace dmnd;

namespace {

// NCBI BLOSUM62 background frequencies (ARNDCQEGHILKMFPSTWYV), the values the reference's statistics use
// (src/stats/matrices/blosum62.h, Stats::blosum62.background_freqs)
const double BLOSUM62_BG[20] = { 7.4216205067993410e-02, 5.1614486141284638e-02, 4.4645808512757915e-02, 5.3626000838554413e-02,
	2.4687457167944848e-02, 3.4259650591416023e-02, 5.4311925684587502e-02, 7.4146941452644999e-02, 2.6212984805266227e-02,
	6.7917367618953756e-02, 9.8907868497150955e-02, 5.8155682303079680e-02, 2.4990197579643110e-02, 4.7418459742284751e-02,
	3.8538003320306206e-02, 5.7229029476494421e-02, 5.0891364550287033e-02, 1.3029956129972148e-02, 3.2281512313758580e-02,
	7.2919098205619245e-02 };

struct HostCfg {
	ScoreTable S;
	double background_scores[20];
	int xdrop = 20;                      // config.raw_ungapped_xdrop = score_matrix.rawscore(12.3 bits), config.cpp:428,853
	alignas(32) int16_t rows[32][24];    // score matrix rows (row = letter, 20 residue columns padded to 24) for the vectorised window sums
	int cbs_window = 40;                 // config.cbs_window
	int max_target_seqs = 25;
	int max_hsps = 1;                    // config.max_hsps: HSPs reported per target; 0 = all of them
	int global_ranking = 0;              // config.global_ranking_targets > 0: the hits are one record per ranked target, no ranking chunks
	int64_t max_swipe_dp = 1000000;      // config.max_swipe_dp
	int band_mode_fast = 1;              // Extension::Mode::BANDED_FAST for every sensitivity up to --sensitive
	bool ext_full = false;               // Extension::Mode::FULL (--ext full): no chaining, one full-matrix DpTarget per target and context
	double ref_letters = 0;
	double ranking_block_letters = 2e9;
	bool use_cbs = true;                 // Stats::CBS::hauser(config.comp_based_stats): the Hauser bias of modes 1 - 3; modes 0, 4, 5 have none
	int cbs_mode = 1;                    // config.comp_based_stats
	const CbsModel* cbs_model = nullptr; // modes 2 - 5: what the per-target matrix adjustment reads of the scoring matrix (cbs_adjust.h)
	int contexts = 1;                    // align_mode.query_contexts: 1 (blastp) or 6 (blastx: the block holds 6 frames per read)
	double top = -1.0;                   // config.toppercent (--top): >= 0 = report the targets within this percentage of the best bit score
	const Evaluer* evaluer = nullptr;    // ScoreMatrix::evalue / bitscore of the context
	double max_evalue = 0.001;
	double min_bit_score = 0.0;          // config.min_bit_score (--min-score): replaces the e-value cutoff (ScoreMatrix::report_cutoff)
	double min_id = 0.0, query_cover = 0.0, subject_cover = 0.0;      // --id, --query-cover, --subject-cover (filter_hsp, culling.cpp:147-170)
	const int32_t* source_lens = nullptr; // translated queries: DNA read length per query
	bool have_filters() const { return min_id > 0 || query_cover > 0 || subject_cover > 0; }
	bool reported(int score, double evalue) const { return min_bit_score != 0.0 ? evaluer->bitscore(score) >= min_bit_score : evalue <= max_evalue; }
};

void make_cfg(const dmnd_ctx* c, HostCfg& h)
{
	for (int i = 0; i < 32 * 32; ++i) { h.S.m[i] = c->params.matrix8[i]; if ((i & 31) < 24) h.rows[i >> 5][i & 31] = (i & 31) < 20 ? c->params.matrix8[i] : 0; }
	h.S.gap_open = c->params.gap_open; h.S.gap_extend = c->params.gap_extend;
	h.xdrop = (int)std::ceil((12.3 * 0.69314718055994530941723212145818 + std::log(c->params.K)) / c->params.lambda);
	for (int i = 0; i < 20; ++i) {                     // ScoreMatrix::init_background_scores, score_matrix.cpp:241-248
		h.background_scores[i] = 0;
		for (int j = 0; j < 20; ++j) h.background_scores[i] += BLOSUM62_BG[j] * h.S.at(i, j);
	}
}

// HauserCorrection(seq): sliding-window expected-score bias per query position, rounded to int8
// (stats/hauser_correction.cpp:28-107). The window sums of all 20 residue scores are kept as one int16[24] vector and
// updated with the matrix row of the letter entering / leaving the window (vectorised by the compiler).
void hauser_int8(const HostCfg& h, const SeqRef& seq, int8_t* out)
{
	const unsigned l = (unsigned)seq.len, window = (unsigned)h.cbs_window, window_half = std::min(window / 2, l - 1);
	// 16-bit sums are exact: at most 41 window letters x |score| <= 15
	alignas(32) int16_t scores[24] = { 0 };
	auto add = [&](int letter) { const int16_t* r = h.rows[letter]; for (int i = 0; i < 24; ++i) scores[i] = (int16_t)(scores[i] + r[i]); };
	auto sub = [&](int letter) { const int16_t* r = h.rows[letter]; for (int i = 0; i < 24; ++i) scores[i] = (int16_t)(scores[i] - r[i]); };
	auto emit = [&](unsigned m, unsigned n) {
		const int r = seq[(int)m];
		float f = 0.0f;
		if (r < 20) f = (float)h.background_scores[r] - float((int)scores[r] - (int)h.rows[r][r]) / (n - 1);
		out[m] = (int8_t)(f < 0.0f ? f - 0.5f : f + 0.5f);
	};
	unsigned n = 0, hh = 0, m = 0, t = 0;
	while (n < window_half && hh < l) { ++n; add(seq[(int)hh]); ++hh; }
	while (n < window + 1 && hh < l) { ++n; add(seq[(int)hh]); emit(m, n); ++hh; ++m; }
	while (hh < l) { add(seq[(int)hh]); sub(seq[(int)t]); emit(m, n); ++hh; ++t; ++m; }
	while (m < l && n > window_half + 1) { --n; sub(seq[(int)t]); emit(m, n); ++t; ++m; }
	while (m < l) { emit(m, n); ++m; }
}

int band_for(int len, bool fast)                            // Extension::band, gapped_score.cpp:41-73
{
	if (fast) return len < 50 ? 12 : len < 100 ? 16 : len < 250 ? 30 : len < 350 ? 40 : 64;
	return len < 50 ? 15 : len < 100 ? 20 : len < 150 ? 30 : len < 200 ? 50 : len < 250 ? 60 : len < 350 ? 100 : len < 500 ? 120 : 150;
}

int64_t ranking_chunk_size(double ref_letters, int max_target_seqs, double default_letters, bool toppercent)       // extend.cpp:79-92
{
	const int64_t block_mult = std::max((int64_t)std::llround(ref_letters / default_letters), (int64_t)1);
	if (toppercent) return 128 * block_mult;             // MIN_CHUNK_SIZE
	const int64_t m32 = ((int64_t)max_target_seqs + 31) / 32 * 32;
	return std::max((int64_t)128, std::min(m32, (int64_t)400)) * block_mult;
}

DeviceCfg device_cfg(const HostCfg& h)      // what the device half reads of the configuration (extend_device.h)
{
	DeviceCfg d;
	d.gap_open = h.S.gap_open; d.gap_extend = h.S.gap_extend; d.band_mode_fast = h.band_mode_fast; d.max_target_seqs = h.max_target_seqs;
	d.ranking_chunk = ranking_chunk_size(h.ref_letters, h.max_target_seqs, h.ranking_block_letters, false);
	d.max_swipe_dp = h.max_swipe_dp; d.use_cbs = h.use_cbs; d.max_evalue = h.max_evalue;
	return d;
}

struct PlanTarget { uint32_t query, target; int32_t d_begin, d_end, ungapped_score; };      // query = block sequence id = query id * contexts + frame

// One query's seed hits grouped by target (SeedHitList of load_hits) + its ranking state (extend.cpp:226-344)
struct TargetGroup { uint32_t target; size_t begin, end; int score; bool pass; };      // pass: survives the gapped filter (any hit flagged)

struct QueryWork {
	uint32_t query = 0;
	std::vector<HostSeedHit> sh;            // seed hits, grouped by target
	std::vector<TargetGroup> groups;        // in load order (ascending target)
	std::vector<uint32_t> order;            // ranking order: indices into groups (l.target_scores)
	int64_t chunk_size = 0;
	size_t i0 = 0, i1 = 0;                  // current ranking chunk [i0, i1) of `order`
	// planned on the device (plan_kernels.hip): the query's group records (parallel to `groups`) and the call's band list; sh stays
	// empty until a group that the device left to the host (PLAN_ON_HOST) is planned
	const PlanGroup* dev = nullptr;
	const PlanBand* dev_bands = nullptr;
	const dmnd_seed_hit* hb = nullptr;      // the query's hits and the index of the first one in the call's hit list
	int64_t first_index = -1;
};

// the target ranking of extend() (extend.cpp:403-414) over the loaded groups: the order in which ranking chunks take them
void rank_groups(const HostCfg& h, QueryWork& w, int query_len)
{
	w.order.resize(w.groups.size());
	for (size_t i = 0; i < w.order.size(); ++i) w.order[i] = (uint32_t)i;
	w.chunk_size = ranking_chunk_size(h.ref_letters, h.max_target_seqs, h.ranking_block_letters, h.top >= 0.0);
	if (h.global_ranking > 0) w.chunk_size = std::max<int64_t>((int64_t)w.groups.size(), 1);      // extend.cpp:80: one chunk = all ranked targets
	if (w.chunk_size < (int64_t)w.groups.size())      // TargetScore::operator<: score desc, target index asc (target.h:146-158)
		std::sort(w.order.begin(), w.order.end(), [&](uint32_t a, uint32_t b) {
			return w.groups[a].score > w.groups[b].score || (w.groups[a].score == w.groups[b].score && a < b);
		});
	w.i0 = 0;
	w.i1 = std::min((size_t)w.chunk_size, w.order.size());
	// a first chunk smaller than -k (k above MAX_CHUNK_SIZE) grows by 16 targets at a time while their seed-hit score would pass
	// the e-value cutoff against a 50-letter target (extend.cpp:262-268, UNIFIED_TARGET_LEN)
	if (h.top < 0.0 && h.min_bit_score == 0.0 && (int64_t)(w.i1 - w.i0) < (int64_t)h.max_target_seqs && h.evaluer)
		while (w.i1 < w.order.size() && h.evaluer->evalue(w.groups[w.order[w.i1]].score, (unsigned)query_len, 50u) <= h.max_evalue)
			w.i1 += std::min<size_t>(16, w.order.size() - w.i1);
}

// load_hits as the device did it (plan_kernels.hip): the query's groups are copied from its PlanGroup records, no hit is touched
void load_query_planned(const HostCfg& h, QueryWork& w, uint32_t query, const PlanGroup* g, size_t n_groups, const PlanBand* bands,
	const dmnd_seed_hit* hb, int64_t first_index, int query_len)
{
	w.query = query;
	w.sh.clear();
	w.dev = g; w.dev_bands = bands; w.hb = hb; w.first_index = first_index;
	w.groups.resize(n_groups);
	for (size_t k = 0; k < n_groups; ++k)
		w.groups[k] = TargetGroup{ g[k].target, (size_t)(g[k].hit_begin - (uint32_t)first_index), (size_t)(g[k].hit_begin - (uint32_t)first_index) + g[k].n_hits, (int)g[k].score, g[k].pass != 0 };
	rank_groups(h, w, query_len);
}

// load_hits (load_hits.h:44-127) + the target ranking of extend() (extend.cpp:403-414)
void load_query(const HostCfg& h, QueryWork& w, uint32_t query, const dmnd_seed_hit* hb, const dmnd_seed_hit* he, const uint8_t* gf_flags,
	const int64_t* tl, int64_t nt, const uint32_t* coarse = nullptr, int query_len = 0, int64_t first_index = -1)
{
	auto by_subject = [](const dmnd_seed_hit& a, const dmnd_seed_hit& b) {       // Hit::CmpSubject
		return a.subject < b.subject || (a.subject == b.subject && (a.query < b.query || (a.query == b.query && a.seed_offset < b.seed_offset)));
	};
	// dmnd_seed_search hands the hits over sorted by (query, subject, seed offset): for an untranslated query that IS the order wanted
	// here, and the copy + sort (a third of this function's time, itself a third of the host time of a --sensitive step) is skipped.
	// A translated query's hits come frame after frame and are merged by the sort.
	const size_t n_hits = (size_t)(he - hb);
	const bool in_order = std::is_sorted(hb, he, by_subject);
	std::vector<dmnd_seed_hit> sorted_copy;
	const dmnd_seed_hit* hits = hb;
	if (!in_order) {
		sorted_copy.assign(hb, he);
		// carried through the sort: bit 0 = the hit's gapped-filter flag, the rest = its position in [hb, he)
		for (size_t x = 0; x < n_hits; ++x) sorted_copy[x].pad = (int32_t)((gf_flags ? (gf_flags[x] ? 1u : 0u) : 1u) | ((uint32_t)x << 1));
		std::sort(sorted_copy.begin(), sorted_copy.end(), by_subject);
		hits = sorted_copy.data();
	}
	auto carried = [&](size_t x) -> uint32_t { return in_order ? (gf_flags ? (gf_flags[x] ? 1u : 0u) : 1u) | ((uint32_t)x << 1) : (uint32_t)hits[x].pad; };
	w.query = query;
	w.sh.resize(n_hits);
	w.groups.clear();
	const int64_t* it = tl;
	for (size_t x = 0; x < n_hits; ++x) {
		const int64_t s = hits[x].subject;
		if (coarse) {                                        // sequences starting inside the hit's 4 KiB stretch, from the table's entry on
			const int64_t* lo = tl + coarse[s >> dmnd_ctx::COARSE_SHIFT];
			if (lo > it) it = lo;
			while (it + 1 <= tl + nt && it[1] <= s) ++it;
			++it;
		}
		else
			it = std::upper_bound(it, tl + nt + 1, s);
		const uint32_t t = (uint32_t)(it - tl) - 1;
		--it;
		if (w.groups.empty() || w.groups.back().target != t) w.groups.push_back(TargetGroup{ t, x, x, 0, false });
		w.sh[x] = HostSeedHit{ hits[x].seed_offset, (int)(s - tl[t]), hits[x].score, (int)(hits[x].query % (uint32_t)h.contexts),
			first_index >= 0 ? (int)(first_index + (int64_t)(carried(x) >> 1)) : -1 };
		w.groups.back().end = x + 1;
		w.groups.back().score = std::max(w.groups.back().score, (int)(uint16_t)hits[x].score);
		w.groups.back().pass |= (carried(x) & 1u) != 0;  // gapped-filter flag of the hit (1 everywhere when the filter is off)
	}
	rank_groups(h, w, query_len);
}

// ungapped_stage + chaining + add_dp_targets for the targets order[g0, g1) of one query (all its contexts)
void plan_groups(const HostCfg& h, ChainWorkspace& ws, QueryWork& w, size_t g0, size_t g1,
	const int8_t* qdata, const int64_t* ql, const int8_t* tdata, const int64_t* tl, const int8_t* cbs_all, std::vector<PlanTarget>& out,
	const XdropSeg* xd = nullptr)      // xd: the x-drop extension of every hit as the device computed it (xdrop_seg_kernel); NULL: walked here
{
	const int C = h.contexts;
	const uint32_t q0 = w.query * (uint32_t)C;                  // block id of context 0
	SeqRef q[6];
	const int8_t* cbs[6];
	for (int f = 0; f < C; ++f) {
		q[f] = SeqRef{ qdata + ql[q0 + f], (int)(ql[q0 + f + 1] - ql[q0 + f] - 1) };
		cbs[f] = cbs_all ? cbs_all + ql[q0 + f] : nullptr;
	}
	std::vector<Seg> segs[6];
	std::vector<Chain> chains;
	std::vector<HostSeedHit>& sh = w.sh;
	const int base_band = band_for(q[0].len, h.band_mode_fast != 0);       // Extension::band(query_seq->length(), mode): context 0
	for (size_t gi = g0; gi < g1; ++gi) {
		const TargetGroup& g = w.groups[w.order[gi]];
		if (!g.pass) continue;                               // gapped filter (extend.cpp:205-213): dropped before chaining
		if (w.dev) {
			// planned on the device: the bands are looked up. A group the device left to the host goes through the code below, on
			// the query's seed hits (built here, once, when the first such group turns up)
			const PlanGroup& pg = w.dev[w.order[gi]];
			if (pg.n_bands != PLAN_ON_HOST) {
				for (uint32_t k = 0; k < pg.n_bands; ++k)
					out.push_back(PlanTarget{ q0, g.target, w.dev_bands[pg.band_begin + k].d_begin, w.dev_bands[pg.band_begin + k].d_end, g.score });
				continue;
			}
			if (sh.empty()) {
				sh.resize(w.groups.back().end);
				for (const TargetGroup& tg : w.groups)
					for (size_t x = tg.begin; x < tg.end; ++x)
						sh[x] = HostSeedHit{ w.hb[x].seed_offset, (int)(w.hb[x].subject - tl[tg.target]), w.hb[x].score, 0, (int)(w.first_index + (int64_t)x) };
			}
		}
		const SeqRef t{ tdata + tl[g.target], (int)(tl[g.target + 1] - tl[g.target] - 1) };
		int ungapped[6] = { 0, 0, 0, 0, 0, 0 };
		for (int f = 0; f < C; ++f) segs[f].clear();
		if (h.ext_full) {
			// Mode::FULL (ungapped.cpp:70-75, gapped_score.cpp:123-130): the contexts that have a seed hit, whole matrix each
			for (size_t x = g.begin; x < g.end; ++x) ungapped[sh[x].frame] = std::max(ungapped[sh[x].frame], sh[x].score);
			for (int f = 0; f < C; ++f)
				if (ungapped[f] != 0) out.push_back(PlanTarget{ q0 + (uint32_t)f, g.target, -(t.len - 1), q[f].len, ungapped[0] });
			continue;
		}
		const bool single_translated = C > 1 && g.end - g.begin == 1;      // ungapped.cpp:76-80: one seed hit of a translated query
		if (single_translated) {                                           // becomes a one-diagonal ApproxHsp without extension
			const HostSeedHit& x = sh[g.begin];
			ungapped[x.frame] = x.score;
		}
		else {
			std::sort(sh.begin() + (ptrdiff_t)g.begin, sh.begin() + (ptrdiff_t)g.end, [](const HostSeedHit& a, const HostSeedHit& b) {
				const int d1 = a.i - a.j, d2 = b.i - b.j;
				return d1 < d2 || (d1 == d2 && a.j < b.j);
			});
			for (size_t x = g.begin; x < g.end; ++x) {
				const int f = sh[x].frame;
				ungapped[f] = std::max(ungapped[f], sh[x].score);
				if (!segs[f].empty() && segs[f].back().diag() == sh[x].i - sh[x].j && segs[f].back().j_end() >= sh[x].j) continue;
				const Seg d = xd && sh[x].src >= 0
					? Seg{ sh[x].i - xd[sh[x].src].left, sh[x].j - xd[sh[x].src].left, xd[sh[x].src].left + xd[sh[x].src].right, xd[sh[x].src].score }
					: xdrop_ungapped(h.S, q[f], cbs[f], t, sh[x].i, sh[x].j, h.xdrop);
				if (d.score > 0) segs[f].push_back(d);
			}
		}
		for (int f = 0; f < C; ++f) {
			if (single_translated) {
				const HostSeedHit& x = sh[g.begin];
				if (x.frame != f) continue;
				chains.clear();
				chains.push_back(Chain{ x.i - x.j, x.i - x.j, x.score, x.i, x.i + 1, x.j, x.j + 1 });
			}
			else {
				if (segs[f].empty()) continue;
				// (std::stable_sort takes a temporary buffer from the heap even for one element: most targets have one segment and one chain)
				if (segs[f].size() > 1) std::stable_sort(segs[f].begin(), segs[f].end(), [](const Seg& a, const Seg& b) { return a.diag() < b.diag() || (a.diag() == b.diag() && a.j < b.j); });
				ws.run(h.S, q[f], t, segs[f], chains);
				if (chains.size() > 1) std::stable_sort(chains.begin(), chains.end(), [](const Chain& a, const Chain& b) { return a.d_min < b.d_min; });
			}
			// add_dp_targets (gapped_score.cpp:107-180): merge overlapping bands of the context's chains
			int d0 = INT_MAX, d1 = INT_MIN;
			for (const Chain& c : chains) {
				const int b0 = std::max(c.d_min - base_band, -(t.len - 1)), b1 = std::min(c.d_max + 1 + base_band, q[f].len);
				const int lo = std::max(d0, b0), hi = std::min(d1, b1);
				const double overlap = hi > lo ? hi - lo : 0;
				// (d1 - d0) wraps for the initial (INT_MAX, INT_MIN) pair exactly as in the reference: the first chain never merges
				const double wd = (double)(int)((unsigned)d1 - (unsigned)d0);
				if (overlap / wd > 0.0 || overlap / (b1 - b0) > 0.0) { d0 = std::min(d0, b0); d1 = std::max(d1, b1); }
				else {
					if (d0 != INT_MAX) out.push_back(PlanTarget{ q0 + (uint32_t)f, g.target, d0, d1, ungapped[0] });
					d0 = b0; d1 = b1;
				}
			}
			if (d0 != INT_MAX) out.push_back(PlanTarget{ q0 + (uint32_t)f, g.target, d0, d1, ungapped[0] });
		}
	}
}

struct Range { size_t b, e; };

std::vector<Range> split_by_query(const dmnd_seed_hit* hits, int64_t n, int contexts)
{
	std::vector<Range> r;
	const uint32_t C = (uint32_t)contexts;
	for (int64_t i = 0; i < n;) {
		int64_t j = i;
		while (j < n && hits[j].query / C == hits[i].query / C) ++j;
		r.push_back(Range{ (size_t)i, (size_t)j });
		i = j;
	}
	return r;
}

// all queries: load + plan every target (used by the CPU-checkable dmnd_extend_plan)
int plan_all(const HostCfg& h, int threads, const dmnd_seed_hit* hits, int64_t n_hits,
	const int8_t* qdata, const std::vector<int64_t>& ql, const int8_t* tdata, const std::vector<int64_t>& tl,
	const int8_t* cbs_all, std::vector<PlanTarget>& out)
{
	const std::vector<Range> qr = split_by_query(hits, n_hits, h.contexts);
	std::vector<std::vector<PlanTarget>> per(qr.size());
	threads = std::max(1, threads);
	std::vector<ChainWorkspace> ws((size_t)threads);
	// the table dmnd_upload_block keeps for load_query (ctx.h coarse[]), so that this entry runs the code the extension stage runs
	const int64_t n_seqs = (int64_t)tl.size() - 1;
	std::vector<uint32_t> coarse((size_t)(tl[(size_t)n_seqs] >> dmnd_ctx::COARSE_SHIFT) + 2, 0);
	{
		int64_t sidx = 0;
		for (size_t b = 0; b < coarse.size(); ++b) {
			const int64_t pos = (int64_t)b << dmnd_ctx::COARSE_SHIFT;
			while (sidx + 1 <= n_seqs && tl[(size_t)sidx + 1] <= pos) ++sidx;
			coarse[b] = (uint32_t)std::min<int64_t>(sidx, n_seqs - 1);
		}
	}
	parallel_for(qr.size(), threads, [&](size_t i, int t) {
		QueryWork w;
		load_query(h, w, hits[qr[i].b].query / (uint32_t)h.contexts, hits + qr[i].b, hits + qr[i].e, nullptr, tl.data(), n_seqs, n_seqs > 0 ? coarse.data() : nullptr);
		plan_groups(h, ws[(size_t)t], w, 0, w.order.size(), qdata, ql.data(), tdata, tl.data(), cbs_all, per[i]);
	});
	size_t total = 0;
	for (auto& v : per) total += v.size();
	out.clear(); out.reserve(total);
	for (auto& v : per) out.insert(out.end(), v.begin(), v.end());
	return DMND_OK;
}

void all_hauser(const HostCfg& h, int threads, const int8_t* qdata, const std::vector<int64_t>& ql, std::vector<int8_t>& cbs)
{
	cbs.assign((size_t)ql.back() + 64, 0);
	parallel_for(ql.size() - 1, threads, [&](size_t i, int) {
		const SeqRef q{ qdata + ql[i], (int)(ql[i + 1] - ql[i] - 1) };
		if (q.len > 0) hauser_int8(h, q, cbs.data() + ql[i]);
	});
}

}  // namespace

// ---- C ABI ----------------------------------------------------------------------------------------------------------

// Pure host part (no device needed): the Hauser bias of every query and the round-1 DpTargets the extension stage
// would send to the swipe. Exposed so that the band construction can be checked on a CPU-only box.
extern "C" int dmnd_extend_plan(const dmnd_params* params, const int8_t* qdata, const int64_t* qlimits, int64_t nq,
	const int8_t* tdata, const int64_t* tlimits, int64_t nt, const dmnd_seed_hit* hits, int64_t n_hits, int threads, int query_contexts,
	int8_t* cbs_out, dmnd_plan_target* out, int64_t cap, int64_t* n_out)
{
	if (!params || !qdata || !qlimits || !tdata || !tlimits || (!hits && n_hits) || !n_out) return fail(DMND_E_ARG, "dmnd_extend_plan: NULL argument");
	if ((query_contexts != 1 && query_contexts != 6) || nq % query_contexts != 0) return fail(DMND_E_ARG, "dmnd_extend_plan: query_contexts must be 1 or 6 and divide the block size");
	dmnd_ctx tmp;
	tmp.params = *params;
	HostCfg h;
	make_cfg(&tmp, h);
	h.ref_letters = (double)(tlimits[nt] - tlimits[0] - nt);
	h.contexts = query_contexts;
	const std::vector<int64_t> ql(qlimits, qlimits + nq + 1), tl(tlimits, tlimits + nt + 1);
	std::vector<int8_t> cbs;
	all_hauser(h, threads, qdata, ql, cbs);
	if (cbs_out) std::memcpy(cbs_out, cbs.data(), (size_t)ql.back());
	std::vector<PlanTarget> plan;
	if (int rc = plan_all(h, threads, hits, n_hits, qdata, ql, tdata, tl, cbs.data(), plan)) return rc;
	*n_out = (int64_t)plan.size();
	if ((int64_t)plan.size() > cap) return fail(DMND_E_CAP, "dmnd_extend_plan: output buffer too small");
	for (size_t i = 0; i < plan.size(); ++i)
		out[i] = dmnd_plan_target{ plan[i].query, plan[i].target, plan[i].d_begin, plan[i].d_end, plan[i].ungapped_score };
	return DMND_OK;
}

namespace {

struct Cand {              // one target of one query after round 1 (Extension::Target; with max_hsps = 1 its one HSP)
	uint32_t target;
	int score, d_begin, d_end, ungapped, frame;
	double evalue;
	int arena = -1;            // >= 0: the round-1 sweep kept this DpTarget's trace (KeptTrace kts[arena], entry `item`)
	int64_t item = -1;
	int band_off = -1, band_n = 0;      // max_hsps != 1: ALL reported round-1 DpTargets of the target, QueryState::bands[band_off, band_off + band_n)
};

// max_hsps != 1: a round-1 DpTarget whose score passed the report cutoff. Round 1 runs without coordinates, so Target::inner_culling
// is not applied to them (gapped_score.cpp:240) and every one of them is a DpTarget of round 2 (add_dp_targets, gapped_final.cpp:62-76).
struct Band { int d_begin, d_end, frame, arena; int64_t item; };

// A reported target of a query = Extension::Match: its first HSP in m (m.evalue / m.hsp.score double as Match::filter_evalue /
// filter_score) and, with max_hsps != 1, the further HSPs of its list in QueryState::extra[extra] (Hsp::operator< order)
struct MatchG { dmnd_match m; int extra = -1; };

bool cand_less(const Cand& a, const Cand& b)                 // Target::comp_evalue, target.h:123-129
{
	return a.evalue < b.evalue || (a.evalue == b.evalue && (a.score > b.score || (a.score == b.score && a.target < b.target)));
}

bool cand_less_score(const Cand& a, const Cand& b)           // Target::comp_score
{
	return a.score > b.score || (a.score == b.score && a.target < b.target);
}

// what is reported of a sorted list (output_range, culling.cpp:97-113): the first -k entries, or with --top the entries whose bit
// score is within the percentage of the best one
struct CullCfg { int k; double top; const Evaluer* ev; };

template<typename Score>
size_t output_range(size_t n, const CullCfg& cc, Score score_at)
{
	if (n == 0) return 0;
	if (cc.top < 0.0) return std::min(n, (size_t)cc.k);
	const double cutoff = std::max((1.0 - cc.top / 100.0) * cc.ev->bitscore(score_at(0)), 1.0);      // top_cutoff_score<double>
	size_t i = 0;
	while (i < n && cc.ev->bitscore(score_at(i)) >= cutoff) ++i;
	return i;
}

// culling(targets, sort_only, cfg) for first-round targets (culling.cpp:189-193)
void cull(std::vector<Cand>& t, bool sort_only, const CullCfg& cc)
{
	std::sort(t.begin(), t.end(), cc.top >= 0.0 ? cand_less_score : cand_less);
	if (!sort_only) t.resize(output_range(t.size(), cc, [&](size_t i) { return t[i].score; }));
}

// culling(matches, cfg), culling.cpp:199-202. A match whose HSP was removed by a filter stays in the list as a placeholder with
// filter_evalue = DBL_MAX, filter_score = 0 (Match::apply_filters): it sorts last, and output_range drops the placeholders at the
// end of the reported range (and everything, if the best entry is one).
void cull(std::vector<MatchG>& t, const CullCfg& cc)
{
	if (cc.top >= 0.0) std::sort(t.begin(), t.end(), [](const MatchG& a, const MatchG& b) { return match_less_score(a.m, b.m); });
	else std::sort(t.begin(), t.end(), [](const MatchG& a, const MatchG& b) { return match_less(a.m, b.m); });
	if (t.empty() || t[0].m.evalue == DBL_MAX) { t.clear(); return; }
	size_t n = output_range(t.size(), cc, [&](size_t i) { return t[i].m.hsp.score; });
	if (cc.top < 0.0) while (n > 1 && t[n - 1].m.evalue == DBL_MAX) --n;
	t.resize(n);
}

// Hsp::query_source_range (TranslatedPosition::absolute_interval, basic/translated_position.h:130-136): the query interval of an HSP
// in the coordinates of the source sequence -- the DNA read for a translated query
void source_interval(const dmnd_match& m, int contexts, int dna_len, int& b, int& e)
{
	if (contexts == 1) { b = m.hsp.q_begin; e = m.hsp.q_end; return; }
	const int offset = m.frame % 3;
	if (m.frame < 3) { b = offset + 3 * m.hsp.q_begin; e = offset + 3 * m.hsp.q_end; }
	else { e = dna_len - offset - 3 * m.hsp.q_begin; b = dna_len - offset - 3 * m.hsp.q_end; }
}

// Match::inner_culling for max_hsps != 1 (culling.cpp:40-57): the HSPs of a target in Hsp::operator< order (score descending, band
// start, query start: basic/match.h:199-202; list::sort is stable), an HSP dropped when half of its query or subject range lies
// inside a better one that was kept (Hsp::is_enveloped_by, hssp.cpp:233-244; --culling-overlap 50), the list cut at max_hsps
void inner_culling(std::vector<dmnd_match>& l, int contexts, int dna_len, int max_hsps)
{
	if (l.size() <= 1) return;
	struct Key { int qb, qe; };
	std::vector<Key> k(l.size());
	std::vector<size_t> ord(l.size());
	for (size_t i = 0; i < l.size(); ++i) { source_interval(l[i], contexts, dna_len, k[i].qb, k[i].qe); ord[i] = i; }
	std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) {
		return l[a].hsp.score > l[b].hsp.score || (l[a].hsp.score == l[b].hsp.score && (l[a].d_begin < l[b].d_begin || (l[a].d_begin == l[b].d_begin && k[a].qb < k[b].qb)));
	});
	auto overlap_factor = [](int b0, int e0, int b1, int e1) {      // Interval::overlap_factor: overlap / own length
		const int o = std::max(0, std::min(e0, e1) - std::max(b0, b1));
		return (double)o / (double)(e0 - b0);
	};
	std::vector<size_t> kept;
	for (size_t i : ord) {
		bool enveloped = false;
		for (size_t j : kept)
			if (overlap_factor(k[i].qb, k[i].qe, k[j].qb, k[j].qe) >= 0.5
				|| overlap_factor(l[i].hsp.s_begin, l[i].hsp.s_end, l[j].hsp.s_begin, l[j].hsp.s_end) >= 0.5) { enveloped = true; break; }
		if (!enveloped) kept.push_back(i);
	}
	if (max_hsps > 0 && kept.size() > (size_t)max_hsps) kept.resize((size_t)max_hsps);
	std::vector<dmnd_match> out;
	out.reserve(kept.size());
	for (size_t i : kept) out.push_back(l[i]);
	l.swap(out);
}

// append_hits(targets, begin, end, with_culling, cfg), culling.cpp:115-145
bool append_hits(std::vector<Cand>& targets, const std::vector<Cand>& v, const CullCfg& cc, bool with_culling)
{
	if (v.empty()) return false;
	bool new_hits = cc.top < 0.0 && (int)targets.size() < cc.k;
	bool append = !with_culling || new_hits;
	cull(targets, append, cc);
	double min_evalue = DBL_MAX;
	int max_score = 0;
	for (const Cand& c : v) { min_evalue = std::min(min_evalue, c.evalue); max_score = std::max(max_score, c.score); }
	const size_t range_end = output_range(targets.size(), cc, [&](size_t i) { return targets[i].score; });
	if (targets.empty()
		|| (cc.top < 0.0 && min_evalue <= targets[range_end - 1].evalue)
		|| (cc.top >= 0.0 && max_score >= (int)((1.0 - cc.top / 100.0) * targets[range_end - 1].score))) {      // top_cutoff_score<int>
		append = true; new_hits = true;
	}
	if (append) targets.insert(targets.end(), v.begin(), v.end());
	return new_hits;
}

// Per-query ranking state of Extension::extend (extend.cpp:226-344)
struct QueryState {
	QueryWork w;
	std::vector<Cand> aligned;          // aligned_targets of the current outer iteration
	std::vector<Band> bands;            // max_hsps != 1: the reported DpTargets of the aligned targets (Cand::band_off)
	std::vector<MatchG> matches;
	std::vector<std::vector<dmnd_match>> extra;      // max_hsps != 1: HSP lists beyond the first (MatchG::extra)
	std::vector<size_t> r2_slot;        // max_hsps != 1: first result slot of every target of the current round-2 step
	bool alt_pending = false;           // round 2 done, alternative HSPs (recompute_alt_hsps) still to be searched
	int tail_score = 0, previous_tail_score = 0;
	bool new_hits_ev = false;
	bool in_inner = true, done = false;
	std::vector<PlanTarget> plan;       // DpTargets of the current chunk
	size_t item_begin = 0, item_end = 0;
	// round 2 (align(), gapped_final.cpp:80-160): the aligned targets are extended [r2_pos, r2_end) at a time
	bool in_round2 = false;
	size_t r2_pos = 0, r2_end = 0;
	std::vector<MatchG> round;          // the round's matches so far
	// --comp-based-stats 2 - 5 (WorkTarget::WorkTarget, ungapped.cpp:44-58): the query's composition, and per planned target the
	// number of its adjusted matrix among the work context's (dmnd_append_matrices), -1 = the target keeps the standard matrix;
	// values <= -2 between the two halves of a planning step: -2 - (number among the slice's new matrices)
	double comp[20];
	int true_aa = -1;                   // -1: composition not computed yet
	std::unordered_map<uint32_t, int32_t> mat_of;
	int32_t matrix_of(uint32_t target) const { const auto it = mat_of.find(target); return it == mat_of.end() ? -1 : it->second; }
};

}

// Steps 2.. of the extension stage for the queries qr[qr_begin, qr_end): c = the context that owns the blocks, limits, bias
// and statistics parameters (read only), w = the context whose stream, device work buffers and counters this range uses
// (w == c, or one of c's auxiliary contexts when the block is processed as concurrent sub-batches).
// Host work is cut into `threads` fixed slices of the query range, slice t always handled by participant t of the calling
// thread's pool (parallel_each): a query's state is built, updated and released by one thread, so it stays in that core's
// caches across the phases and its memory goes back to the allocator arena it came from. The GPU phases in between are ONE
// launch each over the items of all slices.
static int extend_range(const dmnd_ctx* c, dmnd_ctx* w, const HostCfg& h, const std::vector<Range>& qr, size_t qr_begin, size_t qr_end,
	const dmnd_seed_hit* hits, const std::vector<uint8_t>& gf, const int8_t* qdata, const int8_t* tdata, const int8_t* cbs,
	int threads, uint32_t hsp_values, std::vector<dmnd_match>& out_matches,
	uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used, hipStream_t bias_stream = nullptr, const XdropSeg* xd = nullptr,
	const DevPlan* dp = nullptr,      // dp: the call's groups and bands as the device planned them (entry k of dp->queries = query range qr[k] ...
	const uint32_t* pq_index = nullptr)      // ... or, when qr is a subset of the call's queries, entry pq_index[k])
{
	const std::vector<int64_t>& ql = c->limits[DMND_QUERY];
	const std::vector<int64_t>& tl = c->limits[DMND_TARGET];
	const uint32_t C = (uint32_t)h.contexts;
	const int K = h.max_target_seqs;
	const CullCfg cc{ K, h.top, &c->evaluer };
	const bool first_round_culling = !h.have_filters() || h.top >= 0.0;      // extend.cpp:272
	const bool multi = h.max_hsps != 1;                                       // several HSPs per target: every reported band goes through round 2
	for (int i = 0; i < 12; ++i) if (i != 4 || w != c) w->ext_stats[i] = 0;      // [4] (bias + upload) of the caller's prelude is kept
	for (double& x : w->host_ms) x = 0;
	auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	auto cells_of = [](const dmnd_dp_target& d) { return (double)dmnd_banded_cols(d.query_len, d.target_len, d.d_begin, d.d_end) * (double)(d.d_end - d.d_begin); };
	double t_mark = now();
	const double t_enter = t_mark;
	double fine[16] = { 0 }, at[16] = { 0 };          // DMND_TRACE=1: finer host timeline on stderr (at[]: end of each phase since dmnd_extend began, first pass)
	// DMND_TRACE: CPU time of the whole process per phase beside the wall time (only meaningful when nothing else runs, as in the
	// serial steps bench.py appends to its timed region)
	static const bool trace_cpu = std::getenv("DMND_TRACE") != nullptr;
	auto cpu_now = [] { timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
	double cpu[16] = { 0 }, cpu_mark = trace_cpu ? cpu_now() : 0;
	auto lap = [&](int slot, int f = -1) {
		const double t = now(); w->ext_stats[slot] += t - t_mark;
		if (f >= 0) { fine[f] += t - t_mark; if (at[f] == 0) at[f] = t - g_extend_t0; if (trace_cpu) { const double cn = cpu_now(); cpu[f] += cn - cpu_mark; cpu_mark = cn; } }
		t_mark = t;
	};
	lap(4, 1);
	const size_t nq = qr_end - qr_begin;
	const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(threads, 1), (nq + 63) / 64));
	auto slice_begin = [&](int t) { return nq * (size_t)t / (size_t)T; };
	struct Ref { size_t q, k; };
	struct KeptGroup { std::vector<dmnd_dp_target> items; std::vector<int64_t> src; std::vector<Ref> ref; };
	struct NewMatrix { size_t q; uint32_t target; };
	struct Slice {                       // what slice t contributes to the phase at hand
		std::vector<int8_t> new_mats;    // adjusted matrices made in this planning step (32 x 32 each) and whose they are
		std::vector<NewMatrix> new_refs;
		size_t n_items = 0, item_off = 0;
		double cells = 0;
		bool keepable = true, any = false;
		std::vector<KeptGroup> kept;
		std::vector<dmnd_dp_target> it_tb, it_st;
		std::vector<Ref> ref_tb, ref_st;
		size_t n_matches = 0, match_off = 0;
	};
	std::vector<Slice> sl((size_t)T);
	// 2. load_hits for every query
	std::vector<QueryState> qs(nq);
	std::vector<ChainWorkspace> ws((size_t)T);
	const uint32_t* coarse = c->coarse[DMND_TARGET].empty() ? nullptr : c->coarse[DMND_TARGET].data();
	parallel_each(T, [&](int t) {
		for (size_t i = slice_begin(t); i < slice_begin(t + 1); ++i) {
			const Range& r = qr[qr_begin + i];
			const uint32_t q0 = hits[r.b].query / (uint32_t)h.contexts * (uint32_t)h.contexts;       // first context of the query
			if (dp) {
				const size_t pqi = pq_index ? pq_index[qr_begin + i] : qr_begin + i;
				const PlanQuery& pq = dp->queries[pqi];
				load_query_planned(h, qs[i].w, hits[r.b].query, dp->groups + pq.group_begin, (size_t)(dp->queries[pqi + 1].group_begin - pq.group_begin), dp->bands,
					hits + r.b, (int64_t)r.b, (int)(ql[q0 + 1] - ql[q0] - 1));
			}
			else
			load_query(h, qs[i].w, hits[r.b].query / (uint32_t)h.contexts, hits + r.b, hits + r.e, gf.empty() ? nullptr : gf.data() + r.b, tl.data(), (int64_t)tl.size() - 1, coarse,
				(int)(ql[q0 + 1] - ql[q0] - 1), (int64_t)r.b);
			if (qs[i].w.order.empty()) qs[i].done = true;
		}
	});
	if (bias_stream) HIP_TRY(sync_stream(bias_stream));      // the Hauser bias (kernel + copy to the host) ran beside load_hits
	lap(5, 3);
	// mat: number of the target's adjusted matrix (it is then swept with that matrix and without the bias), -1 = standard matrix
	// (the reference's full-matrix sweep -- --ext full, alternative HSPs -- keeps the bias for adjusted targets: full_swipe.h:164)
	auto matrix_code = [&](int32_t mat) { return (int64_t)-2 - mat - (h.use_cbs ? DMND_CBS_MATRIX_WITH_BIAS : (int64_t)0); };
	auto item_of = [&](uint32_t q, uint32_t t, int d0, int d1, int32_t mat) {
		return dmnd_dp_target{ ql[q], tl[t], mat >= 0 ? (h.ext_full ? matrix_code(mat) : (int64_t)-2 - mat) : h.use_cbs ? ql[q] : (int64_t)-1, (int32_t)(ql[q + 1] - ql[q] - 1), (int32_t)(tl[t + 1] - tl[t] - 1), d0, d1 };
	};
	const bool adjust = cbs_matrix_adjust(h.cbs_mode);
	// The reference's full-matrix sweep cannot trace an alignment back on an adjusted matrix: Hsp traceback(...) of
	// dp/swipe/full_swipe.h:97-102 throws as soon as such a target has an HSP to report (--ext full, and the alternative HSPs of
	// --max-hsps != 1). Same refusal here, with its message.
	std::atomic<bool> adjusted_full_traceback(false);
	std::vector<int8_t> upload_mats;
	w->n_adj_matrices = 0;
	const bool full_matrix = h.ext_full;               // DP::Flags::FULL_MATRIX: DpTarget::cells = query length x target length, dp/dp.h:121-124
	auto dp_size = [full_matrix](const dmnd_dp_target& d) {
		return full_matrix ? (int64_t)d.query_len * (int64_t)d.target_len
			: (int64_t)dmnd_banded_cols(d.query_len, d.target_len, d.d_begin, d.d_end) * (int64_t)(d.d_end - d.d_begin);
	};
	double sw1 = 0, sw2 = 0, tb2 = 0;
	int64_t used = 0;
	std::vector<dmnd_dp_target> items;
	std::vector<dmnd_hsp> res;
	// Round 2 re-runs the DpTargets that survive culling with traceback -- the same targets, bands and bias as round 1. So
	// round 1 already sweeps in traceback mode and keeps the trace rows (one arena per ranking-chunk iteration of a pass), and
	// round 2 only walks the kept traces. Not used when the caller wants transcripts, when a chunk holds a matrix too large for
	// the traceback path (those go through the statistics kernels), or past the context's trace budget: then round 2 sweeps again.
	static const bool keep_traces = [] { const char* e = std::getenv("DMND_EXTEND_KEEP_TRACE"); return !e || e[0] != '0'; }();
	if (!w->kts) w->kts = new std::vector<KeptTrace>(8);
	std::vector<KeptTrace>& kts = *w->kts;
	for (KeptTrace& k : kts) { k.kept = false; k.arena = -1; }
	std::vector<std::vector<dmnd_hsp>> r2(nq);
	for (;;) {
		int arena_iter = 0;
		// ---- inner loop: ranking chunks, round 1 (score only, or traceback mode with kept traces) ----
		for (;;) {
			// plan the current chunk of every query that is still ranking
			parallel_each(T, [&](int t) {
				Slice& me = sl[(size_t)t];
				me.n_items = 0; me.any = false;
				for (size_t i = slice_begin(t); i < slice_begin(t + 1); ++i) {
					QueryState& s = qs[i];
					if (s.done || !s.in_inner) continue;
					me.any = true;
					s.plan.clear();
					plan_groups(h, ws[(size_t)t], s.w, s.w.i0, s.w.i1, qdata, ql.data(), tdata, tl.data(), h.use_cbs ? cbs : nullptr, s.plan, xd);
					me.n_items += s.plan.size();
					if (!adjust) continue;
					// the adjusted matrix of every target that got a DpTarget (the reference makes one for every target of the chunk,
					// WorkTarget::WorkTarget; only the sweeps read it)
					const uint32_t q0 = s.w.query * C;
					if (s.true_aa < 0) cbs_composition(qdata + ql[q0], (int)(ql[q0 + 1] - ql[q0] - 1), s.comp, &s.true_aa);
					for (const PlanTarget& p : s.plan) {
						if (s.mat_of.count(p.target)) continue;
						const int8_t* tseq = tdata + tl[p.target];
						const int tlen = (int)(tl[p.target + 1] - tl[p.target] - 1);
						const int rule = cbs_rule(*h.cbs_model, h.cbs_mode, s.comp, s.true_aa, tseq, tlen);
						if (rule == CBS_RULE_NONE) { s.mat_of[p.target] = -1; continue; }
						s.mat_of[p.target] = -2 - (int32_t)me.new_refs.size();
						me.new_refs.push_back(NewMatrix{ i, p.target });
						me.new_mats.resize(me.new_mats.size() + 32 * 32);
						cbs_target_matrix(*h.cbs_model, rule, s.comp, s.true_aa, tseq, tlen, me.new_mats.data() + me.new_mats.size() - 32 * 32);
					}
				}
			});
			bool any = false;
			size_t total = 0;
			for (Slice& x : sl) { any |= x.any; x.item_off = total; total += x.n_items; }
			if (!any) break;
			if (adjust) {
				// the step's new matrices get their numbers (slice order) and go to the device behind the ones already there
				const int64_t before = w->n_adj_matrices;
				int64_t next = before;
				upload_mats.clear();
				for (Slice& x : sl) {
					for (const NewMatrix& r : x.new_refs) qs[r.q].mat_of[r.target] = (int32_t)next++;
					upload_mats.insert(upload_mats.end(), x.new_mats.begin(), x.new_mats.end());
					x.new_refs.clear(); x.new_mats.clear();
				}
				if (next > 0x7fffffff) return fail(DMND_E_CAP, "dmnd_extend: more than 2^31 adjusted matrices in one call");
				if (next > before) { if (int rc = dmnd_append_matrices(w, before, upload_mats.data(), next - before)) return rc; }
			}
			items.resize(total);
			parallel_each(T, [&](int t) {
				Slice& me = sl[(size_t)t];
				size_t o = me.item_off;
				me.cells = 0; me.keepable = true;
				for (size_t i = slice_begin(t); i < slice_begin(t + 1); ++i) {
					QueryState& s = qs[i];
					if (s.done || !s.in_inner) continue;
					s.item_begin = o;
					for (const PlanTarget& p : s.plan) {
						const dmnd_dp_target d = item_of(p.query, p.target, p.d_begin, p.d_end, adjust ? s.matrix_of(p.target) : -1);
						me.cells += cells_of(d);
						me.keepable &= dp_size(d) <= h.max_swipe_dp;
						items[o++] = d;
					}
					s.item_end = o;
				}
			});
			lap(5, 4);
			res.assign(items.size(), dmnd_hsp());
			int arena = -1;
			if (!items.empty()) {
				bool keep = keep_traces && !transcript && arena_iter < (int)kts.size();
				for (const Slice& x : sl) keep &= x.keepable;
				if (keep) {
					if (int rc = dmnd_swipe_keep(w, c, items.data(), (int64_t)items.size(), arena_iter, res.data(), kts[(size_t)arena_iter])) return rc;
					if (kts[(size_t)arena_iter].kept) arena = arena_iter;
					++arena_iter;
				}
				else if (int rc = dmnd_swipe_shared(w, c, items.data(), (int64_t)items.size(), DMND_SWIPE_SCORE, 0, res.data(), nullptr, 0, nullptr)) return rc;
				sw1 += w->swipe_ms;
				w->ext_stats[0] += (double)items.size();
				for (const Slice& x : sl) w->ext_stats[2] += x.cells;
			}
			lap(6, 5);
			parallel_each(T, [&](int t) {
				std::vector<Cand> v;
				for (size_t qi = slice_begin(t); qi < slice_begin(t + 1); ++qi) {
					QueryState& s = qs[qi];
					if (s.done || !s.in_inner) continue;
					// extend_chunk -> align (gapped_score.cpp:182-268): report cutoff, best HSP per target
					v.clear();
					for (size_t x = s.item_begin; x < s.item_end; ++x) {
						const PlanTarget& p = s.plan[x - s.item_begin];
						const int score = res[x].score;
						if (score <= 0) continue;
						const double ev = c->evaluer.evalue(score, (unsigned)items[x].query_len, (unsigned)items[x].target_len);
						if (!h.reported(score, ev)) continue;
						const int frame = (int)(p.query % C);
						if (!v.empty() && v.back().target == p.target) {
							// Target::add_hit(list, it) (target.h:105-113): the best context is the first one (contexts ascending) that
							// reaches the maximum score; inner_culling then keeps that context's best HSP by (score desc, d_begin asc)
							// (Hsp::operator<, match.h:199)
							Cand& k = v.back();
							if (score > k.score || (score == k.score && frame == k.frame && p.d_begin < k.d_begin)) {
								k.score = score; k.evalue = ev; k.d_begin = p.d_begin; k.d_end = p.d_end; k.frame = frame; k.arena = arena; k.item = (int64_t)x;
							}
							if (multi) { s.bands.push_back(Band{ p.d_begin, p.d_end, frame, arena, (int64_t)x }); ++k.band_n; }
						}
						else {
							v.push_back(Cand{ p.target, score, p.d_begin, p.d_end, p.ungapped_score, frame, ev, arena, (int64_t)x });
							if (multi) { v.back().band_off = (int)s.bands.size(); v.back().band_n = 1; s.bands.push_back(Band{ p.d_begin, p.d_end, frame, arena, (int64_t)x }); }
						}
					}
					const bool multi_chunk = (s.w.i1 - s.w.i0) < s.w.order.size();
					bool new_hits = s.new_hits_ev = !v.empty();
					if (multi_chunk) new_hits = append_hits(s.aligned, v, cc, first_round_culling);
					else s.aligned = v;
					// advance the chunk window (extend.cpp:325-329)
					s.w.i0 = s.w.i1;
					s.w.i1 += std::min((size_t)s.w.chunk_size, s.w.order.size() - s.w.i1);
					s.previous_tail_score = s.tail_score;
					const int next_tail = s.w.groups[s.w.order[s.w.i1 - 1]].score;
					if (new_hits) s.tail_score = next_tail;
					// ranking_terminate (extend.cpp:111-119) with default options
					const bool terminate = !new_hits && (s.previous_tail_score == 0
						|| (double)next_tail / (double)s.previous_tail_score <= 0.95
						|| c->evaluer.bitscore(next_tail) < 25.0);
					if (!(s.w.i0 < s.w.order.size() && !terminate)) s.in_inner = false;
				}
			});
			lap(7, 6);
		}
		// ---- round 2 for every query that just left the inner loop: all its aligned targets at once, or -- with --id / --query-cover /
		// --subject-cover -- a step at a time until enough matches pass the filters (gapped_final.cpp:105-152) ----
		bool any_round2 = false;
		for (;;) {
		parallel_each(T, [&](int t) {
			Slice& me = sl[(size_t)t];
			me.any = false;
			me.kept.assign(kts.size(), KeptGroup());
			me.it_tb.clear(); me.it_st.clear(); me.ref_tb.clear(); me.ref_st.clear();
			for (size_t i = slice_begin(t); i < slice_begin(t + 1); ++i) {
				QueryState& s = qs[i];
				if (s.done || s.in_inner) continue;
				me.any = true;
				if (!s.in_round2) {
					cull(s.aligned, !first_round_culling, cc);                       // extend.cpp:331
					s.in_round2 = true; s.r2_pos = 0; s.round.clear();
				}
				const size_t left = s.aligned.size() - s.r2_pos;
				size_t step = left;
				if (!first_round_culling && h.top < 0.0)
					step = std::min<size_t>(((size_t)std::max<int64_t>((int64_t)K - (int64_t)s.round.size(), 16) + 15) / 16 * 16, left);
				s.r2_end = s.r2_pos + step;
				// one DpTarget per aligned target (its best band), or -- max_hsps != 1 -- one per reported band of the target
				// (add_dp_targets, gapped_final.cpp:62-76): result slot k, or r2_slot[k - r2_pos] + band
				auto add_item = [&](uint32_t target, int frame, int d0, int d1, int arena, int64_t item, size_t slot) {
					const dmnd_dp_target d = item_of(s.w.query * C + (uint32_t)frame, target, d0, d1, adjust ? s.matrix_of(target) : -1);
					// DP::BandedSwipe::bin (swipe_wrapper.cpp:75-102): above max_swipe_dp cells the statistics cells replace the traceback,
					// unless the output needs the transcript -- then the matrix is traced whatever its size
					if (dp_size(d) > h.max_swipe_dp && !transcript) { me.it_st.push_back(d); me.ref_st.push_back(Ref{ i, slot }); }
					else if (arena >= 0) { KeptGroup& g = me.kept[(size_t)arena]; g.items.push_back(d); g.src.push_back(item); g.ref.push_back(Ref{ i, slot }); }
					else { me.it_tb.push_back(d); me.ref_tb.push_back(Ref{ i, slot }); }
				};
				if (!multi) {
					r2[i].assign(s.aligned.size(), dmnd_hsp());
					for (size_t k = s.r2_pos; k < s.r2_end; ++k) {
						const Cand& cd = s.aligned[k];
						add_item(cd.target, cd.frame, cd.d_begin, cd.d_end, cd.arena, cd.item, k);
					}
				}
				else {
					s.r2_slot.clear();
					size_t slot = 0;
					for (size_t k = s.r2_pos; k < s.r2_end; ++k) { s.r2_slot.push_back(slot); slot += (size_t)s.aligned[k].band_n; }
					s.r2_slot.push_back(slot);
					r2[i].assign(slot, dmnd_hsp());
					for (size_t k = s.r2_pos; k < s.r2_end; ++k) {
						const Cand& cd = s.aligned[k];
						for (int j = 0; j < cd.band_n; ++j) {
							const Band& b = s.bands[(size_t)(cd.band_off + j)];
							add_item(cd.target, b.frame, b.d_begin, b.d_end, b.arena, b.item, s.r2_slot[k - s.r2_pos] + (size_t)j);
						}
					}
				}
			}
		});
		bool any_batch = false;
		for (const Slice& x : sl) any_batch |= x.any;
		if (!any_batch) break;
		any_round2 = true;
		// the slices' lists, concatenated in slice order (= query order)
		auto gather = [&](auto pick_items, auto pick_refs, std::vector<dmnd_dp_target>& its, std::vector<Ref>& refs) {
			its.clear(); refs.clear();
			for (Slice& x : sl) { its.insert(its.end(), pick_items(x).begin(), pick_items(x).end()); refs.insert(refs.end(), pick_refs(x).begin(), pick_refs(x).end()); }
		};
		std::vector<dmnd_dp_target> it_tb, it_st, it_k;
		std::vector<Ref> ref_tb, ref_st, ref_k;
		std::vector<int64_t> src_k;
		gather([](Slice& x) -> std::vector<dmnd_dp_target>& { return x.it_tb; }, [](Slice& x) -> std::vector<Ref>& { return x.ref_tb; }, it_tb, ref_tb);
		gather([](Slice& x) -> std::vector<dmnd_dp_target>& { return x.it_st; }, [](Slice& x) -> std::vector<Ref>& { return x.ref_st; }, it_st, ref_st);
		lap(7, 7);
		for (size_t a = 0; a < kts.size(); ++a) {                               // walks over the traces kept by round 1
			gather([a](Slice& x) -> std::vector<dmnd_dp_target>& { return x.kept[a].items; }, [a](Slice& x) -> std::vector<Ref>& { return x.kept[a].ref; }, it_k, ref_k);
			if (it_k.empty()) continue;
			src_k.clear();
			for (Slice& x : sl) src_k.insert(src_k.end(), x.kept[a].src.begin(), x.kept[a].src.end());
			res.assign(it_k.size(), dmnd_hsp());
			if (int rc = dmnd_traceback_kept(w, c, it_k.data(), kts[a], src_k.data(), (int64_t)it_k.size(), res.data())) return rc;
			for (size_t x = 0; x < ref_k.size(); ++x) r2[ref_k[x].q][ref_k[x].k] = res[x];
			tb2 += w->traceback_ms;
			w->ext_stats[1] += (double)it_k.size();                              // the reference's round-2 targets and cells
			for (const dmnd_dp_target& d : it_k) w->ext_stats[3] += cells_of(d);
		}
		if (!it_tb.empty()) {
			uint8_t* arena = transcript ? transcript + used : nullptr;           // NULL: statistics only, no transcripts copied back
			int64_t arena_cap = transcript ? transcript_cap - used : 0;
			res.assign(it_tb.size(), dmnd_hsp());
			int64_t u = 0;
			if (int rc = dmnd_swipe_shared(w, c, it_tb.data(), (int64_t)it_tb.size(), DMND_SWIPE_TRACEBACK, hsp_values, res.data(), arena, arena_cap, &u)) return rc;
			for (size_t x = 0; x < ref_tb.size(); ++x) {
				r2[ref_tb[x].q][ref_tb[x].k] = res[x];
				if (transcript) r2[ref_tb[x].q][ref_tb[x].k].transcript_off += used; else r2[ref_tb[x].q][ref_tb[x].k].transcript_off = -1;
			}
			if (transcript) used += u;
			sw2 += w->swipe_ms; tb2 += w->traceback_ms;
		}
		if (!it_st.empty()) {
			res.assign(it_st.size(), dmnd_hsp());
			if (int rc = dmnd_swipe_shared(w, c, it_st.data(), (int64_t)it_st.size(), DMND_SWIPE_STATS, hsp_values, res.data(), nullptr, 0, nullptr)) return rc;
			for (size_t x = 0; x < ref_st.size(); ++x) { r2[ref_st[x].q][ref_st[x].k] = res[x]; r2[ref_st[x].q][ref_st[x].k].transcript_len = 0; r2[ref_st[x].q][ref_st[x].k].transcript_off = -1; }
			sw2 += w->swipe_ms;
		}
		w->ext_stats[1] += (double)(it_tb.size() + it_st.size());
		for (const dmnd_dp_target& d : it_tb) w->ext_stats[3] += cells_of(d);
		for (const dmnd_dp_target& d : it_st) w->ext_stats[3] += cells_of(d);
		lap(8, 8);
		parallel_each(T, [&](int t) {
			for (size_t i = slice_begin(t); i < slice_begin(t + 1); ++i) {
				QueryState& s = qs[i];
				if (s.done || s.in_inner) continue;
				// align() round 2 (gapped_final.cpp:80-160): report cutoff again, filters, culling of this round's matches
				std::vector<MatchG>& round = s.round;
				const uint32_t q = s.w.query;
				// one HSP of target cd in context `frame` as a match record; false: no score, or below the report cutoff
				auto make = [&](const Cand& cd, int frame, int d0, int d1, const dmnd_hsp& hsp, dmnd_match& m) {
					if (hsp.score <= 0) return false;
					const int tlen = (int)(tl[cd.target + 1] - tl[cd.target] - 1);
					const uint32_t qc = q * C + (uint32_t)frame;
					const int qlen = (int)(ql[qc + 1] - ql[qc] - 1);
					const double ev = c->evaluer.evalue(hsp.score, (unsigned)qlen, (unsigned)tlen);
					if (!h.reported(hsp.score, ev)) return false;
					if (adjust && h.ext_full && s.matrix_of(cd.target) >= 0 && ((int64_t)qlen * (int64_t)tlen <= h.max_swipe_dp || transcript)) adjusted_full_traceback = true;
					m.query = q; m.target = cd.target; m.evalue = ev; m.bit_score = c->evaluer.bitscore(hsp.score); m.read_begin = m.read_end = 0;
					m.ungapped_score = cd.ungapped; m.d_begin = d0; m.d_end = d1; m.frame = frame; m.hsp = hsp;
					if (multi && h.ext_full) m.d_begin = m.d_end = 0;       // Hsp::d_begin of a full-matrix sweep (tie-break of Hsp::operator<)
					return true;
				};
				// filter_hsp (culling.cpp:147-170): --id, --query-cover, --subject-cover, --no-self-hits
				auto filtered = [&](const dmnd_match& m) {
					const dmnd_hsp& hsp = m.hsp;
					const int tlen = (int)(tl[m.target + 1] - tl[m.target] - 1);
					const uint32_t qc = q * C + (uint32_t)m.frame;
					const int qlen = (int)(ql[qc + 1] - ql[qc] - 1);
					if (h.have_filters() && ((double)hsp.identities * 100.0 / (double)hsp.length < h.min_id
						|| (C == 1 ? (double)(hsp.q_end - hsp.q_begin) * 100 / qlen : (double)(3 * (hsp.q_end - hsp.q_begin)) * 100 / (h.source_lens ? h.source_lens[q] : 1)) < h.query_cover
						|| (double)(hsp.s_end - hsp.s_begin) * 100 / tlen < h.subject_cover)) return true;
					// --no-self-hits: same letters (Sequence::operator==, the query's first context) and same title
					if (c->same_title && C == 1 && qlen == tlen) {
						const int8_t* a = qdata + ql[qc];
						const int8_t* b = tdata + tl[m.target];
						bool same = true;
						for (int x = 0; x < qlen && same; ++x) same = ((a[x] ^ b[x]) & 31) == 0;
						if (same && c->same_title(c->same_title_user, q, m.target)) return true;
					}
					return false;
				};
				std::vector<dmnd_match> L;
				for (size_t k = s.r2_pos; k < s.r2_end; ++k) {
					const Cand& cd = s.aligned[k];
					dmnd_match m;
					if (!multi) {
						if (!make(cd, cd.frame, cd.d_begin, cd.d_end, r2[i][k], m)) continue;
						// the HSP is removed, the match stays as a placeholder for the culling below
						if (filtered(m)) { m.evalue = DBL_MAX; m.hsp.score = 0; }
						round.push_back(MatchG{ m, -1 });
						continue;
					}
					// max_hsps != 1: the HSPs of all the target's bands; Match::inner_culling, then the filters HSP by HSP
					// (Match::apply_filters: filter values = those of the first HSP left)
					L.clear();
					for (int j = 0; j < cd.band_n; ++j) {
						const Band& b = s.bands[(size_t)(cd.band_off + j)];
						if (make(cd, b.frame, b.d_begin, b.d_end, r2[i][s.r2_slot[k - s.r2_pos] + (size_t)j], m)) L.push_back(m);
					}
					if (L.empty()) continue;
					inner_culling(L, (int)C, h.source_lens ? h.source_lens[q] : 0, h.max_hsps);
					const dmnd_match first = L[0];
					L.erase(std::remove_if(L.begin(), L.end(), filtered), L.end());
					if (L.empty()) { m = first; m.evalue = DBL_MAX; m.hsp.score = 0; round.push_back(MatchG{ m, -1 }); continue; }
					MatchG g{ L[0], -1 };
					if (L.size() > 1) { g.extra = (int)s.extra.size(); s.extra.emplace_back(L.begin() + 1, L.end()); }
					round.push_back(g);
				}
				cull(round, cc);
				s.r2_pos = s.r2_end;
				if (s.r2_pos < s.aligned.size() && (h.top >= 0.0 || (int)(round.size() + s.matches.size()) < K)) continue;      // next step (goon)
				if (multi) { s.alt_pending = true; continue; }                 // recompute_alt_hsps (gapped_final.cpp:156) comes first
				s.in_round2 = false;
				s.matches.insert(s.matches.end(), round.begin(), round.end());
				round.clear();
				s.aligned.clear();
				// outer loop condition (extend.cpp:336)
				if (h.top < 0.0 && (int)s.matches.size() < K && s.w.i0 < s.w.order.size() && s.new_hits_ev) s.in_inner = true;
				else s.done = true;
			}
		});
		if (adjusted_full_traceback) return fail(DMND_E_ARG, "Traceback with adjusted matrix not supported");
		// recompute_alt_hsps (alt_hsp.cpp:86-142) for the queries whose round 2 is complete: every reported target is copied per
		// context with the subject ranges of its HSPs overwritten by SUPER_HARD_MASK (letter 25, scored like the matrix minimum)
		// and swept over the WHOLE matrix; an HSP found is added, masked, and the target goes round again until a sweep finds
		// nothing, its copy is masked through, or it has max_hsps HSPs. One launch per round for all queries of the batch.
		if (multi) {
			struct Alt { size_t qi, mi; uint32_t target; int tlen; int64_t off[6]; uint32_t have, sweep; std::vector<dmnd_match> hs; };
			std::vector<Alt> alts;
			std::vector<int8_t> scratch(256, (int8_t)31);
			auto mask_range = [&](Alt& a, const dmnd_match& m) {
				std::fill(scratch.begin() + (ptrdiff_t)(a.off[m.frame] + m.hsp.s_begin), scratch.begin() + (ptrdiff_t)(a.off[m.frame] + m.hsp.s_end), (int8_t)25);
			};
			for (size_t i = 0; i < nq; ++i) {
				QueryState& s = qs[i];
				if (!s.alt_pending) continue;
				for (size_t mi = 0; mi < s.round.size(); ++mi) {
					const MatchG& g = s.round[mi];
					if (g.m.evalue == DBL_MAX) continue;                  // every HSP filtered: nothing to mask, no context
					Alt a;
					a.qi = i; a.mi = mi; a.target = g.m.target; a.tlen = (int)(tl[a.target + 1] - tl[a.target] - 1); a.have = 0; a.sweep = 0;
					a.hs.push_back(g.m);
					if (g.extra >= 0) a.hs.insert(a.hs.end(), s.extra[(size_t)g.extra].begin(), s.extra[(size_t)g.extra].end());
					for (const dmnd_match& m : a.hs) {                   // ActiveTarget::copy_seq: one copy per context that has an HSP
						if (!((a.have >> m.frame) & 1u)) {
							a.off[m.frame] = (int64_t)scratch.size();
							scratch.insert(scratch.end(), tdata + tl[a.target], tdata + tl[a.target] + a.tlen);
							scratch.push_back((int8_t)31);
							a.have |= 1u << m.frame;
						}
						mask_range(a, m);
					}
					a.sweep = a.have;
					alts.push_back(std::move(a));
				}
			}
			std::vector<dmnd_dp_target> it_alt[2];                      // [0] traceback, [1] statistics cells (matrix above max_swipe_dp)
			struct ARef { size_t a; int frame; };
			std::vector<ARef> ref_alt[2];
			while (!alts.empty()) {
				for (int k = 0; k < 2; ++k) { it_alt[k].clear(); ref_alt[k].clear(); }
				for (size_t x = 0; x < alts.size(); ++x) {
					const Alt& a = alts[x];
					const uint32_t q = qs[a.qi].w.query;
					for (int f = 0; f < (int)C; ++f) {
						if (!((a.sweep >> f) & 1u)) continue;
						const uint32_t qc = q * C + (uint32_t)f;
						const int qlen = (int)(ql[qc + 1] - ql[qc] - 1);
						// (the alternative HSPs are searched with the match's own matrix: alt_hsp.cpp:91)
						const int32_t mat = adjust ? qs[a.qi].matrix_of(a.target) : -1;
						const dmnd_dp_target d{ ql[qc], a.off[f], mat >= 0 ? matrix_code(mat) : h.use_cbs ? ql[qc] : (int64_t)-1, qlen, a.tlen, -(a.tlen - 1), qlen };
						const int k = ((int64_t)qlen * (int64_t)a.tlen > h.max_swipe_dp && !transcript) ? 1 : 0;
						it_alt[k].push_back(d); ref_alt[k].push_back(ARef{ x, f });
					}
				}
				const size_t bytes = scratch.size() + 256;
				if (int rc = w->alt_targets.ensure(bytes)) return rc;
				HIP_TRY(hipMemsetAsync(w->alt_targets.as<int8_t>() + scratch.size(), 31, 256, w->stream));
				HIP_TRY(hipMemcpyAsync(w->alt_targets.p, scratch.data(), scratch.size(), hipMemcpyHostToDevice, w->stream));
				std::vector<uint32_t> found(alts.size(), 0);
				for (int k = 0; k < 2; ++k) {
					if (it_alt[k].empty()) continue;
					res.assign(it_alt[k].size(), dmnd_hsp());
					uint8_t* arena = transcript && k == 0 ? transcript + used : nullptr;
					const int64_t arena_cap = transcript && k == 0 ? transcript_cap - used : 0;
					int64_t u = 0;
					if (int rc = dmnd_swipe_targets(w, c, w->alt_targets.as<int8_t>(), (int64_t)bytes, it_alt[k].data(), (int64_t)it_alt[k].size(),
						k == 0 ? DMND_SWIPE_TRACEBACK : DMND_SWIPE_STATS, hsp_values, res.data(), arena, arena_cap, &u)) return rc;
					sw2 += w->swipe_ms; if (k == 0) tb2 += w->traceback_ms;
					w->ext_stats[1] += (double)it_alt[k].size();
					for (const dmnd_dp_target& d : it_alt[k]) w->ext_stats[3] += (double)d.query_len * (double)d.target_len;
					for (size_t x = 0; x < ref_alt[k].size(); ++x) {
						Alt& a = alts[ref_alt[k][x].a];
						const int f = ref_alt[k][x].frame;
						dmnd_hsp hsp = res[x];
						if (hsp.score <= 0) continue;
						const double ev = c->evaluer.evalue(hsp.score, (unsigned)it_alt[k][x].query_len, (unsigned)a.tlen);
						if (!h.reported(hsp.score, ev)) continue;
						if (k == 0 && it_alt[k][x].cbs_off <= -2) return fail(DMND_E_ARG, "Traceback with adjusted matrix not supported");
						if (k == 0 && transcript) hsp.transcript_off += used; else { hsp.transcript_off = -1; if (k == 1) hsp.transcript_len = 0; }
						dmnd_match m;
						m.query = a.hs[0].query; m.target = a.target; m.evalue = ev; m.bit_score = c->evaluer.bitscore(hsp.score); m.read_begin = m.read_end = 0;
						m.ungapped_score = a.hs[0].ungapped_score; m.d_begin = 0; m.d_end = 0; m.frame = f; m.hsp = hsp;      // (the full-matrix sweep leaves Hsp::d_begin / d_end at 0)
						a.hs.push_back(m);
						mask_range(a, m);
						found[ref_alt[k][x].a] |= 1u << f;
					}
					if (transcript && k == 0) used += u;
				}
				std::vector<Alt> next;
				for (size_t x = 0; x < alts.size(); ++x) {
					Alt& a = alts[x];
					if (!found[x]) continue;                              // nothing new: the target is finished (its record is up to date)
					QueryState& s = qs[a.qi];
					inner_culling(a.hs, (int)C, h.source_lens ? h.source_lens[s.w.query] : 0, h.max_hsps);
					MatchG& g = s.round[a.mi];
					g.m = a.hs[0];
					if (a.hs.size() > 1) {
						if (g.extra < 0) { g.extra = (int)s.extra.size(); s.extra.emplace_back(); }
						s.extra[(size_t)g.extra].assign(a.hs.begin() + 1, a.hs.end());
					}
					else if (g.extra >= 0) s.extra[(size_t)g.extra].clear();
					// check_fully_masked: a context goes on while its copy still has a residue (Util::Seq::is_fully_masked: all letters >= 20)
					uint32_t active = found[x];
					for (int f = 0; f < (int)C; ++f) {
						if (!((active >> f) & 1u)) continue;
						bool residue = false;
						for (int p = 0; p < a.tlen && !residue; ++p) residue = (scratch[(size_t)(a.off[f] + p)] & 31) < 20;
						if (!residue) active &= ~(1u << f);
					}
					if (active && (h.max_hsps == 0 || (int)a.hs.size() < h.max_hsps)) { a.sweep = active; next.push_back(std::move(a)); }
				}
				// Which of them really go round again. The reference collects them with `out.emplace_back(t)` into a growing
				// std::vector<ActiveTarget> (alt_hsp.cpp:116-121), and ActiveTarget's copy constructor -- which is what a reallocation
				// of that vector relocates its elements with -- drops the masked copy of every context that is not in `active`, of an
				// element already in `out` (active = 0) all of them (alt_hsp.cpp:47-56). With the doubling growth of the vector only the
				// elements appended since its last reallocation keep their copies: of a query's n continuing targets those from index
				// 2^floor(log2(n - 1)) on. The others are swept no more. Restated as it stands: the outputs are compared byte for byte.
				{
					std::vector<Alt> go;
					for (size_t b = 0; b < next.size();) {
						size_t e = b;
						while (e < next.size() && next[e].qi == next[b].qi) ++e;
						const size_t n = e - b;
						size_t first = 0;
						if (n >= 2) { first = 1; while (first * 2 <= n - 1) first *= 2; }
						for (size_t x = b + first; x < e; ++x) go.push_back(std::move(next[x]));
						b = e;
					}
					next.swap(go);
				}
				alts.swap(next);
			}
			for (size_t i = 0; i < nq; ++i) {
				QueryState& s = qs[i];
				if (!s.alt_pending) continue;
				s.alt_pending = false;
				s.in_round2 = false;
				s.matches.insert(s.matches.end(), s.round.begin(), s.round.end());
				s.round.clear();
				s.aligned.clear();
				s.bands.clear();
				if (h.top < 0.0 && (int)s.matches.size() < K && s.w.i0 < s.w.order.size() && s.new_hits_ev) s.in_inner = true;
				else s.done = true;
			}
		}
		}
		if (!any_round2) break;
		lap(7, 9);
	}
	w->swipe_ms = sw1 + sw2; w->traceback_ms = tb2;
	w->ext_stats[9] = sw1; w->ext_stats[10] = sw2; w->ext_stats[11] = tb2;
	if (transcript_used) *transcript_used = transcript ? used : 0;
	// final culling(matches, cfg) per query (extend.cpp:341) -> records in query order; every slice sorts, copies out and releases
	// the state of its own queries
	parallel_each(T, [&](int t) {
		Slice& me = sl[(size_t)t];
		me.n_matches = 0;
		for (size_t i = slice_begin(t); i < slice_begin(t + 1); ++i) {
			QueryState& s = qs[i];
			cull(s.matches, cc);
			for (const MatchG& g : s.matches) me.n_matches += 1 + (g.extra >= 0 ? s.extra[(size_t)g.extra].size() : 0);
		}
	});
	size_t total_matches = 0;
	for (Slice& x : sl) { x.match_off = total_matches; total_matches += x.n_matches; }
	out_matches.resize(total_matches);
	parallel_each(T, [&](int t) {
		size_t o = sl[(size_t)t].match_off;
		for (size_t i = slice_begin(t); i < slice_begin(t + 1); ++i) {
			for (const MatchG& g : qs[i].matches) {            // a target's HSPs follow each other (Hsp::operator< order)
				out_matches[o++] = g.m;
				if (g.extra >= 0) for (const dmnd_match& m : qs[i].extra[(size_t)g.extra]) out_matches[o++] = m;
			}
			QueryState empty;
			std::swap(qs[i], empty);
			std::vector<dmnd_hsp>().swap(r2[i]);
		}
		Slice fresh;
		std::swap(sl[(size_t)t], fresh);
	});
	lap(7, 10);
	if (const char* tr = std::getenv("DMND_TRACE")) if (tr[0] == '2')
		std::fprintf(stderr, "  timeline[%zu queries] enter %.2f | bias %.2f load %.2f plan %.2f swipe1 %.2f post1 %.2f build2 %.2f swipe2 %.2f post2 %.2f final %.2f\n",
			qs.size(), t_enter - g_extend_t0, at[1], at[3], at[4], at[5], at[6], at[7], at[8], at[9], at[10]);
	if (std::getenv("DMND_TRACE"))
		std::fprintf(stderr, "dmnd_extend[%zu queries, %d slices] ms: bias %.2f load %.2f plan %.2f swipe1 %.2f post1 %.2f build2 %.2f swipe2 %.2f post2 %.2f final %.2f | swipe host: prep %.2f run %.2f post %.2f\n",
			qs.size(), T, fine[1], fine[3], fine[4], fine[5], fine[6], fine[7], fine[8], fine[9], fine[10], w->host_ms[0], w->host_ms[1], w->host_ms[2]);
	if (trace_cpu)
		std::fprintf(stderr, "dmnd_extend process CPU ms: bias %.2f load %.2f plan %.2f swipe1 %.2f post1 %.2f build2 %.2f swipe2 %.2f post2 %.2f final %.2f\n",
			cpu[1], cpu[3], cpu[4], cpu[5], cpu[6], cpu[7], cpu[8], cpu[9], cpu[10]);
	return DMND_OK;
}

// The whole extension stage for one (query block, reference block) pair on the uploaded blocks. The reference's per-query
// loop over ranking chunks (extend.cpp:289-336) becomes a batch-synchronous state machine: every pass plans the current
// chunk of all still-active queries on the host threads and scores all of them in ONE GPU launch.
extern "C" int dmnd_extend(dmnd_ctx* c, const int8_t* qdata, const int8_t* tdata, const dmnd_seed_hit* hits, int64_t n_hits,
	int threads, uint32_t hsp_values, dmnd_match* out, int64_t cap, int64_t* n_out,
	uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used)
{
	if (!c || !qdata || !tdata || (!hits && n_hits) || !n_out) return fail(DMND_E_ARG, "dmnd_extend: NULL argument");
	struct Total {                     // declared first = destroyed last: covers the release of all per-call state
		std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
		~Total() { if (std::getenv("DMND_TRACE")) std::fprintf(stderr, "dmnd_extend total %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
	} total_clock;
	*n_out = 0;
	if (transcript_used) *transcript_used = 0;
	// the calling thread works with the context's own worker pool for the length of the call (unless it already selected one)
	struct PoolScope {
		int old;
		explicit PoolScope(int k) : old(thread_pool()) { if (old < 0) set_thread_pool(k); }
		~PoolScope() { set_thread_pool(old); }
	} pool_scope(c->pool_id);
	const std::vector<int64_t>& ql = c->limits[DMND_QUERY];
	const std::vector<int64_t>& tl = c->limits[DMND_TARGET];
	if (ql.size() < 2 || tl.size() < 2) return fail(DMND_E_ARG, "dmnd_extend: blocks must be uploaded with limits");
	if (c->frame_shift > 0) {
		// frameshift alignment: the reference's legacy pipeline instead of Extension::extend (align/align.cpp:168-172)
		if (c->global_ranking > 0) return fail(DMND_E_ARG, "Global ranking is not supported in this mode.");
		if (cbs_matrix_adjust(c->comp_based_stats)) return fail(DMND_E_ARG, "This mode of composition based stats is not supported for translated searches.");
		std::vector<dmnd_match> v;
		if (int rc = dmnd_extend_frameshift(c, qdata, tdata, hits, n_hits, threads, v, transcript, transcript_cap, transcript_used)) return rc;
		*n_out = (int64_t)v.size();
		if ((int64_t)v.size() > cap) return fail(DMND_E_CAP, "dmnd_extend: output buffer too small");
		if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(dmnd_match));
		return DMND_OK;
	}
	HostCfg h;
	make_cfg(c, h);
	h.max_target_seqs = c->max_target_seqs;
	h.max_hsps = c->max_hsps;
	h.global_ranking = c->global_ranking;
	if (h.global_ranking > 0 && c->ext_mode != DMND_EXT_FULL) return fail(DMND_E_ARG, "dmnd_extend: globally ranked targets are extended over the full matrix (dmnd_set_extension_mode(DMND_EXT_FULL))");
	h.top = c->top_percent;
	h.evaluer = &c->evaluer;
	h.max_evalue = c->params.max_evalue;
	h.min_bit_score = c->min_bit_score; h.min_id = c->min_id; h.query_cover = c->query_cover; h.subject_cover = c->subject_cover;
	if (h.query_cover > 0 && c->query_contexts != 1 && c->source_lens.size() != (ql.size() - 1) / (size_t)c->query_contexts)
		return fail(DMND_E_ARG, "dmnd_extend: the query cover of translated queries needs the read lengths (dmnd_set_query_source_lengths)");
	if (h.max_hsps != 1 && c->query_contexts != 1 && c->source_lens.size() != (ql.size() - 1) / (size_t)c->query_contexts)
		return fail(DMND_E_ARG, "dmnd_extend: several HSPs per target of translated queries need the read lengths (dmnd_set_query_source_lengths)");
	h.source_lens = c->source_lens.empty() ? nullptr : c->source_lens.data();
	h.ranking_block_letters = c->ranking_block_letters;
	h.band_mode_fast = c->band_mode_fast;
	h.ext_full = c->ext_mode == DMND_EXT_FULL;
	if (c->ext_mode == DMND_EXT_BANDED_FAST) h.band_mode_fast = 1; else if (c->ext_mode == DMND_EXT_BANDED_SLOW) h.band_mode_fast = 0;
	h.contexts = c->query_contexts;
	h.cbs_mode = c->comp_based_stats;
	h.use_cbs = cbs_hauser(h.cbs_mode);
	CbsModel cbs_model;
	if (cbs_matrix_adjust(h.cbs_mode)) {
		// basic/config.cpp:700, :837, :688
		if (c->query_contexts != 1) return fail(DMND_E_ARG, "This mode of composition based stats is not supported for translated searches.");
		if (h.global_ranking > 0) return fail(DMND_E_ARG, "Global ranking is not supported in this mode.");
		cbs_model_init(cbs_model, c->params);
		if (!cbs_model.valid) return fail(DMND_E_ARG, "This value for --comp-based-stats is not supported when using a custom scoring matrix.");
		h.cbs_model = &cbs_model;
	}
	const uint32_t C = (uint32_t)h.contexts;
	if ((ql.size() - 1) % C != 0) return fail(DMND_E_ARG, "dmnd_extend: query block size is not a multiple of the query contexts");
	h.ref_letters = (double)(tl.back() - tl.front() - ((int64_t)tl.size() - 1));
	if (hsp_values == 0) hsp_values = 510;
	threads = std::max(1, threads);
	for (double& x : c->ext_stats) x = 0;
	for (double& x : c->host_ms) x = 0;
	auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double t_mark = now();
	g_extend_t0 = t_mark;
	double fine[16] = { 0 };          // DMND_TRACE=1: finer host timeline on stderr
	auto lap = [&](int slot, int f = -1) { const double t = now(); c->ext_stats[slot] += t - t_mark; if (f >= 0) fine[f] += t - t_mark; t_mark = t; };
	TraceLaps trp("dmnd_extend (prelude)");
	// 1. Hauser bias for every query, resident next to the query block
	const std::vector<Range> qr = split_by_query(hits, n_hits, h.contexts);
	trp.lap("queries split");
	// One launch over the whole query block (bias_kernels.hip: closed-form window per position), result kept in HBM next to
	// the block for the swipe kernels and the gapped filter, and copied into a pinned host buffer for the host's x-drop stage.
	const int8_t* cbs = nullptr;
	bool bias_pending = false;
	static const bool xdrop_gpu = [] { const char* e = std::getenv("DMND_EXTEND_XDROP_GPU"); return !e || e[0] != '0'; }();
	if (!h.use_cbs) {
		c->cbs_len = 0;                                     // --comp-based-stats 0: no bias anywhere on the path
	}
	else {
		HIP_TRY(hipSetDevice(c->device));
		const bool host_walks = !(xdrop_gpu && !h.ext_full && n_hits > 0);
		if (c->cbs.cap < (size_t)ql.back() + 256) c->cbs_generation = ~(uint64_t)0;      // (the buffer is about to be replaced)
		if (int rc = c->cbs.ensure((size_t)ql.back() + 256)) return rc;
		if (host_walks && c->pinned_cbs_cap < (size_t)ql.back() + 64) {
			if (c->pinned_cbs) (void)hipHostFree(c->pinned_cbs);
			c->pinned_cbs = nullptr; c->pinned_cbs_cap = 0;
			HIP_TRY(hipHostMalloc((void**)&c->pinned_cbs, (size_t)ql.back() + 64, hipHostMallocDefault));
			c->pinned_cbs_cap = (size_t)ql.back() + 64;
		}
		// Computed once per query block (dmnd_ctx::query_generation changes whenever the block's letters do) for ALL its sequences and
		// kept in HBM next to the block: every later call against another database block finds it there. The host copy is only made
		// when the host is going to walk letters itself (DMND_EXTEND_XDROP_GPU=0).
		if (c->cbs_generation != c->query_generation || c->cbs_len != ql.back()) {
			BiasArgs ba;
			ba.block = c->block[DMND_QUERY].as<int8_t>(); ba.limits = c->d_limits[DMND_QUERY].as<int64_t>(); ba.n_seqs = (int64_t)ql.size() - 1;
			ba.ids = nullptr;
			ba.matrix = c->matrix.as<int8_t>(); ba.window = h.cbs_window; ba.out = c->cbs.as<int8_t>();
			for (int i = 0; i < 20; ++i) ba.bg[i] = (float)h.background_scores[i];
			HIP_TRY(launch_hauser_bias(ba, c->stream));
			c->cbs_generation = c->query_generation;
			bias_pending = true;                            // awaited after load_hits (extend_range) / before the runners start
		}
		c->cbs_len = ql.back();
		if (host_walks) {
			HIP_TRY(hipMemcpyAsync(c->pinned_cbs, c->cbs.p, (size_t)ql.back(), hipMemcpyDeviceToHost, c->stream));
			bias_pending = true;
			cbs = c->pinned_cbs;
		}
	}
	trp.lap("bias");
	// 1a. x-drop ungapped extension of every seed hit on the device (xdrop_seg_kernel), behind the bias kernel on the same stream;
	// the host's chaining stage picks the segments up instead of walking the letters itself (DMND_EXTEND_XDROP_GPU=0: host walks)
	const XdropSeg* xd = nullptr;
	if (xdrop_gpu && !h.ext_full && n_hits > 0) {
		HIP_TRY(hipSetDevice(c->device));
		if (int rc = c->xd_hits.ensure((size_t)n_hits * sizeof(dmnd_seed_hit))) return rc;
		if (int rc = c->xd_out.ensure((size_t)n_hits * sizeof(XdropSeg))) return rc;
		HIP_TRY(hipMemcpyAsync(c->xd_hits.p, hits, (size_t)n_hits * sizeof(dmnd_seed_hit), hipMemcpyHostToDevice, c->stream));
		XdropArgs xa;
		xa.qblock = c->block[DMND_QUERY].as<int8_t>(); xa.tblock = c->block[DMND_TARGET].as<int8_t>();
		xa.cbs = h.use_cbs ? c->cbs.as<int8_t>() : nullptr;
		xa.qlimits = c->d_limits[DMND_QUERY].as<int64_t>(); xa.matrix = c->matrix.as<int8_t>();
		xa.hits = c->xd_hits.as<dmnd_seed_hit>(); xa.n_hits = n_hits; xa.xdrop = h.xdrop; xa.out = c->xd_out.as<XdropSeg>();
		HIP_TRY(launch_xdrop_segs(xa, c->stream));
		bias_pending = true;                                // the same wait covers it
		xd = reinterpret_cast<const XdropSeg*>(1);          // (set below, once it is known whether the host needs the copy)
	}
	lap(4, 1);
	// The groups, segments, chains and bands of every (query, target) pair on the device (plan_kernels.hip; one query context, banded
	// extension). DMND_EXTEND_PLAN_GPU=0: the host plans, as up to round 5.
	static const bool plan_gpu = [] { const char* e = std::getenv("DMND_EXTEND_PLAN_GPU"); return !e || e[0] != '0'; }();
	const bool try_plan = plan_gpu && xd && h.contexts == 1 && n_hits < ((int64_t)1 << 31);
	// 1b. gapped filter of every seed hit in one launch (only --sensitive and above; extend.cpp:205-213)
	std::vector<uint8_t> gf;
	c->gf_ms = 0;
	const bool gf_on = c->gapped_filter_evalue > 0.0 && n_hits > 0 && h.global_ranking == 0;      // (extend.cpp:206: not for globally ranked targets)
	if (gf_on) {
		if (bias_pending) { HIP_TRY(sync_stream(c->stream)); bias_pending = false; }      // the filter's profile reads the bias
		if (!try_plan) gf.resize((size_t)n_hits);
		// (with the device planner the hits are in HBM already and the flags stay there)
		if (int rc = dmnd_gapped_filter_on(c, hits, try_plan ? c->xd_hits.as<dmnd_seed_hit>() : nullptr, n_hits, h.use_cbs ? 1 : 0, try_plan ? nullptr : gf.data(), nullptr)) return rc;
	}
	lap(4, 2);
	trp.lap("x-drop enqueued, gapped filter");
	DevPlan plan;
	bool planned = false;
	if (try_plan) {
		if (int rc = plan_on_device(c, device_cfg(h), n_hits, gf_on, plan, planned)) return rc;
		trp.lap("planned");
		if (planned && plan.n_queries != qr.size()) planned = false;
		if (!planned && gf_on) {                            // hits out of order (a caller's own list): the host plans, and needs the flags
			gf.resize((size_t)n_hits);
			HIP_TRY(copy_now(c->stream, gf.data(), c->gf_flags.p, (size_t)n_hits, hipMemcpyDeviceToHost));
		}
		bias_pending = false;                               // plan_on_device has waited for the stream
	}
	if (xd && (!planned || plan.n_on_host > 0)) {
		if (int rc = c->xd_host.ensure((size_t)n_hits * sizeof(XdropSeg))) return rc;      // (page-locked: allocated when first needed)
		HIP_TRY(hipMemcpyAsync(c->xd_host.p, c->xd_out.p, (size_t)n_hits * sizeof(XdropSeg), hipMemcpyDeviceToHost, c->stream));
		bias_pending = true;
	}
	if (xd) xd = c->xd_host.as<XdropSeg>();              // (NULL while no group was left to the host: nothing reads it then)
	const DevPlan* dp = planned ? &plan : nullptr;
	// The queries whose targets fit one ranking chunk are extended in HBM from here on (extend_kernels.hip): the default search of a
	// protein query block -- one HSP per target, -k culling by e-value, Hauser bias or none, no --id / cover filters, no transcripts
	// (the caller formats from the statistics). The others, and every query of any other mode, take the host path below.
	// DMND_EXTEND_DEVICE=0: all queries on the host path, as up to round 5.
	static const bool ext_gpu = [] { const char* e = std::getenv("DMND_EXTEND_DEVICE"); return !e || e[0] != '0'; }();
	c->ext_records_dev = nullptr; c->ext_records_n = -1;
	std::vector<dmnd_match> dev_records;
	std::vector<Range> qr_host;
	std::vector<uint32_t> pq_host;
	bool on_device = false;
	for (double& x : c->ext_dev_stats) x = 0;
	if (ext_gpu && planned && h.max_hsps == 1 && !h.have_filters() && h.top < 0.0 && h.min_bit_score == 0.0 && !cbs_matrix_adjust(h.cbs_mode) && !h.ext_full && !transcript
		&& h.global_ranking == 0 && !c->same_title && h.max_target_seqs > 0) {
		std::vector<uint8_t> qstate;
		double kept[12];
		for (int i = 0; i < 12; ++i) kept[i] = c->ext_stats[i];
		if (int rc = extend_on_device(c, device_cfg(h), plan, threads, dev_records, qstate, on_device)) return rc;
		bias_pending = false;                               // (it has waited for the stream)
		if (on_device) {
			for (size_t k = 0; k < qr.size(); ++k)
				if (qstate[k] != EXT_Q_DEVICE) { qr_host.push_back(qr[k]); pq_host.push_back((uint32_t)k); }
		}
		else for (int i = 0; i < 12; ++i) c->ext_stats[i] = kept[i];
	}
	const std::vector<Range>& qr_all = qr;
	const std::vector<Range>& qr_run = on_device ? qr_host : qr_all;
	if (planned && !qr_run.empty()) { if (int rc = plan_fetch_lists(c, plan)) return rc; bias_pending = false; }
	const uint32_t* pq_index = on_device ? pq_host.data() : nullptr;
	double dev_stats[12];
	for (int i = 0; i < 12; ++i) dev_stats[i] = on_device ? c->ext_stats[i] : 0.0;
	c->ext_plan_stats[0] = planned ? (double)plan.n_groups : 0; c->ext_plan_stats[1] = planned ? (double)plan.n_on_host : 0; c->ext_plan_stats[2] = planned ? (double)plan.n_bands : 0;
	lap(4, 2);
	if (const char* tr = std::getenv("DMND_TRACE")) if (tr[0] == '3') {
		const double t0 = now();
		for (int i = 0; i < 20; ++i) parallel_for((size_t)threads, threads, [](size_t, int) {});
		std::fprintf(stderr, "dmnd_extend ms: hauser+upload %.2f gapped_filter %.2f | empty parallel loop over %d threads: %.3f ms\n", fine[1], fine[2], threads, (now() - t0) / 20);
	}
	// 2.. : the queries are cut into `split` sub-batches that `runners` host threads pull from a queue, each runner with its own
	// HIP stream, device work buffers (auxiliary context) and share of the worker threads. While one runner waits for its swipe
	// kernels the others chain, cull and pack on the host; the GPU serialises the runners' launches, which staggers them after
	// the first round, so with more sub-batches than runners host and device phases overlap like a pipeline. Results are
	// concatenated in query order, so the output does not depend on the split.
	// How many: measured on C2 (20k seed hits, 6.7k queries with hits) eight single-threaded runners beat every combination with
	// worker threads under them -- the host phases of a sub-batch are a few hundred microseconds, less than it costs to wake a
	// pool -- and use a quarter of the CPU time, which matters on CPU-quota'd hosts. Worker threads are added per runner only
	// when a runner's share of the seed hits is large enough to pay for them (--sensitive: 1.2e6 hits per block).
	// Default: one range, host work in min(threads, 8) fixed slices, ONE launch per GPU phase (extend_range). DMND_EXTEND_SPLIT /
	// DMND_EXTEND_RUNNERS > 1 select the older layout of independent runners with their own streams and small launches.
	int split = 1, runners = 1;
	split = tuning().extend_split; runners = tuning().extend_runners;
	if (transcript || qr_run.size() < (size_t)split * 2) split = 1;
	runners = std::min(std::min(runners, split), (int)MAX_POOLS);
	std::vector<std::vector<dmnd_match>> parts((size_t)split);
	std::vector<int> rcs((size_t)split, DMND_OK);
	std::vector<std::string> errs((size_t)split);
	const int team = tuning().extend_team;
	if (on_device && qr_run.empty()) {
		if (bias_pending) HIP_TRY(sync_stream(c->stream));      // every query was extended on the device
	}
	else if (split == 1 && cbs_matrix_adjust(h.cbs_mode)) {
		// --comp-based-stats 2-5: every planned (query, target) pair owns a 1 KB adjusted matrix for the whole extend_range call
		// (QueryState::mat_of, dmnd_ctx::adj_matrices), where the reference holds a TargetMatrix for the queries of one chunk only.
		// A block pair of 1e5-1e6 queries would ask for tens of GB of matrices in one call. So the queries go through in passes
		// of at most DMND_CBS_PASS_HITS seed hits (default 2 M: a pair needs a seed hit, so <= 2 GB of matrices, in practice a tenth);
		// a pass frees its matrices (extend_range starts from none). The result does not depend on the cut: queries are independent.
		static const int64_t pass_hits = [] { const char* e = std::getenv("DMND_CBS_PASS_HITS"); return e ? std::max<int64_t>(1, std::atoll(e)) : (int64_t)2 << 20; }();
		int64_t t_used = 0;
		double stats[12] = { 0 };
		for (size_t b = 0; b < qr_run.size() && rcs[0] == DMND_OK;) {
			size_t e = b + 1;
			while (e < qr_run.size() && (int64_t)(qr_run[e].e - qr_run[b].b) <= pass_hits) ++e;
			std::vector<dmnd_match> part;
			int64_t used = 0;
			rcs[0] = extend_range(c, c, h, qr_run, b, e, hits, gf, qdata, tdata, cbs, std::min(threads, team), hsp_values, part, transcript ? transcript + t_used : nullptr,
				transcript ? transcript_cap - t_used : 0, &used, b == 0 && bias_pending ? c->stream : nullptr, xd, dp, pq_index);
			if (rcs[0] != DMND_OK) break;
			if (t_used > 0) for (dmnd_match& m : part) if (m.hsp.transcript_off >= 0) m.hsp.transcript_off += t_used;
			t_used += used;
			parts[0].insert(parts[0].end(), part.begin(), part.end());
			for (int i = 0; i < 12; ++i) stats[i] += c->ext_stats[i];
			c->ext_stats[4] = 0;                             // (extend_range keeps slot 4 of its caller: the prelude's time, counted with the first pass)
			b = e;
		}
		for (int i = 0; i < 12; ++i) c->ext_stats[i] = stats[i];
		if (transcript_used) *transcript_used = t_used;
	}
	else if (split == 1) {
		rcs[0] = extend_range(c, c, h, qr_run, 0, qr_run.size(), hits, gf, qdata, tdata, cbs, std::min(threads, team), hsp_values, parts[0], transcript, transcript_cap, transcript_used,
			bias_pending ? c->stream : nullptr, xd, dp, pq_index);
	}
	else {
		if (bias_pending) HIP_TRY(sync_stream(c->stream));
		const double stats4 = c->ext_stats[4];
		std::vector<dmnd_ctx*> work((size_t)runners);
		for (int r = 0; r < runners; ++r) {
			work[(size_t)r] = aux_context(c, r, runners);
			if (!work[(size_t)r]) return fail(DMND_E_DEVICE, "dmnd_extend: cannot create an auxiliary context");
		}
		std::vector<std::thread> th;
		int sub_threads = std::max(1, std::min(threads / runners, (int)(n_hits / runners / 8192)));
		if (tuning().extend_sub_threads > 0) sub_threads = tuning().extend_sub_threads;
		std::atomic<int> next_sub(0);
		std::vector<std::array<double, 12>> acc((size_t)runners);
		for (auto& x : acc) x.fill(0.0);
		for (int r = 0; r < runners; ++r)
			th.emplace_back([&, r] {
				set_thread_pool(r);
				(void)hipSetDevice(c->device);
				dmnd_ctx* w = work[(size_t)r];
				for (int k; (k = next_sub.fetch_add(1)) < split;) {
					const size_t b = qr_run.size() * (size_t)k / (size_t)split, e = qr_run.size() * (size_t)(k + 1) / (size_t)split;
					rcs[(size_t)k] = extend_range(c, w, h, qr_run, b, e, hits, gf, qdata, tdata, cbs, sub_threads, hsp_values, parts[(size_t)k], nullptr, 0, nullptr, nullptr, xd, dp, pq_index);
					if (rcs[(size_t)k] != DMND_OK) { errs[(size_t)k] = dmnd_last_error(); break; }
					for (int i = 0; i < 12; ++i) acc[(size_t)r][(size_t)i] += w->ext_stats[i];
				}
				set_thread_pool(-1);
			});
		for (auto& t : th) t.join();
		// statistics of the call: counts and device times add up, host wall times are those of the busiest runner
		for (double& x : c->ext_stats) x = 0;
		c->ext_stats[4] = stats4;
		for (int r = 0; r < runners; ++r) {
			const std::array<double, 12>& w = acc[(size_t)r];
			for (int i : { 0, 1, 2, 3, 9, 10, 11 }) c->ext_stats[i] += w[(size_t)i];
			for (int i : { 5, 6, 7, 8 }) c->ext_stats[i] = std::max(c->ext_stats[i], w[(size_t)i]);
			c->ext_stats[4] = std::max(c->ext_stats[4], stats4 + w[4]);
		}
		c->swipe_ms = c->ext_stats[9] + c->ext_stats[10]; c->traceback_ms = c->ext_stats[11];
	}
	for (int k = 0; k < split; ++k)
		if (rcs[(size_t)k] != DMND_OK) return split == 1 ? rcs[0] : fail(rcs[(size_t)k], errs[(size_t)k]);
	int64_t n = (int64_t)dev_records.size();
	for (const auto& v : parts) n += (int64_t)v.size();
	if (!on_device || n != (int64_t)dev_records.size()) { c->ext_records_dev = nullptr; c->ext_records_n = -1; }      // (records of the host path exist on the host only)
	*n_out = n;
	if (n > cap) return fail(DMND_E_CAP, "dmnd_extend: match buffer too small");
	if (on_device) {
		// counts, cells and device times of the two halves add up; the host times are those of the host path
		if (!qr_run.empty()) for (int i : { 0, 1, 2, 3, 9, 11 }) c->ext_stats[i] += dev_stats[i];      // (no host path: the context's counters are the device's)
		c->swipe_ms = c->ext_stats[9] + c->ext_stats[10]; c->traceback_ms = c->ext_stats[11];
	}
	if (out && !on_device) {
		int64_t off = 0;
		for (const auto& v : parts) { std::copy(v.begin(), v.end(), out + off); off += (int64_t)v.size(); }
	}
	else if (out) {
		// the records of the device's queries and of the host's, both in query order, merged (a query is in one of them)
		std::vector<dmnd_match> host_records;
		if (split > 1) for (const auto& v : parts) host_records.insert(host_records.end(), v.begin(), v.end());
		const std::vector<dmnd_match>& hr = split > 1 ? host_records : parts[0];
		size_t x = 0, y = 0;
		int64_t off = 0;
		while (x < dev_records.size() || y < hr.size()) {
			const bool take_dev = y == hr.size() || (x < dev_records.size() && dev_records[x].query < hr[y].query);
			out[off++] = take_dev ? dev_records[x++] : hr[y++];
		}
	}
	return DMND_OK;
}

// xdrop_ungapped for every seed hit of a block pair at once (SURVEY 8(b)'s optional entry; /root/reference/src/dp/ungapped_align.cpp:
// 151-199: from the seed position (qa, sa) the diagonal is walked to the left and to the right, each walk ending at a sequence end
// or once the running score has fallen `xdrop` below the best; DiagonalSegment(qa - delta, sa - delta, len + delta, score)). The kernel
// is dmnd_extend's own first device stage (xdrop_seg_kernel, bias_kernels.hip; arithmetic shared with the host in xdrop_core.h).
// use_bias: the Hauser composition bias of the queries is added to every letter score (Extension::extend's call shape); 0 = none
// (the reference's legacy mapper and global ranking call it that way). xdrop <= 0: config.raw_ungapped_xdrop of the context's matrix.
extern "C" int dmnd_xdrop_ungapped(dmnd_ctx* c, const dmnd_seed_hit* hits, int64_t n_hits, int use_bias, int xdrop, dmnd_diagonal_segment* out)
{
	if (!c || n_hits < 0 || (n_hits > 0 && (!hits || !out))) return fail(DMND_E_ARG, "dmnd_xdrop_ungapped: bad argument");
	const std::vector<int64_t>& ql = c->limits[DMND_QUERY];
	const std::vector<int64_t>& tl = c->limits[DMND_TARGET];
	if (ql.size() < 2 || tl.size() < 2) return fail(DMND_E_ARG, "dmnd_xdrop_ungapped: both blocks must be uploaded with limits");
	if (n_hits == 0) return DMND_OK;
	for (int64_t k = 0; k < n_hits; ++k) {
		const dmnd_seed_hit& h = hits[k];
		if ((size_t)h.query + 1 >= ql.size() || h.seed_offset < 0 || ql[h.query] + h.seed_offset >= ql[h.query + 1] - 1 || h.subject < tl.front() || h.subject >= tl.back())
			return fail(DMND_E_ARG, "dmnd_xdrop_ungapped: seed hit " + std::to_string(k) + " outside the blocks");
	}
	HIP_TRY(hipSetDevice(c->device));
	HostCfg h;
	make_cfg(c, h);
	if (use_bias) {
		c->cbs_generation = ~(uint64_t)0;
		if (int rc = c->cbs.ensure((size_t)ql.back() + 256)) return rc;
		BiasArgs ba;
		ba.block = c->block[DMND_QUERY].as<int8_t>(); ba.limits = c->d_limits[DMND_QUERY].as<int64_t>(); ba.n_seqs = (int64_t)ql.size() - 1; ba.ids = nullptr;
		ba.matrix = c->matrix.as<int8_t>(); ba.window = h.cbs_window; ba.out = c->cbs.as<int8_t>();
		for (int i = 0; i < 20; ++i) ba.bg[i] = (float)h.background_scores[i];
		HIP_TRY(launch_hauser_bias(ba, c->stream));
	}
	if (int rc = c->xd_hits.ensure((size_t)n_hits * sizeof(dmnd_seed_hit))) return rc;
	if (int rc = c->xd_out.ensure((size_t)n_hits * sizeof(XdropSeg))) return rc;
	HIP_TRY(hipMemcpyAsync(c->xd_hits.p, hits, (size_t)n_hits * sizeof(dmnd_seed_hit), hipMemcpyHostToDevice, c->stream));
	XdropArgs xa;
	xa.qblock = c->block[DMND_QUERY].as<int8_t>(); xa.tblock = c->block[DMND_TARGET].as<int8_t>();
	xa.cbs = use_bias ? c->cbs.as<int8_t>() : nullptr;
	xa.qlimits = c->d_limits[DMND_QUERY].as<int64_t>(); xa.matrix = c->matrix.as<int8_t>();
	xa.hits = c->xd_hits.as<dmnd_seed_hit>(); xa.n_hits = n_hits; xa.xdrop = xdrop > 0 ? xdrop : h.xdrop; xa.out = c->xd_out.as<XdropSeg>();
	HIP_TRY(launch_xdrop_segs(xa, c->stream));
	std::vector<XdropSeg> seg((size_t)n_hits);
	if (int rc = download_bytes(c, seg.data(), c->xd_out.p, (size_t)n_hits * sizeof(XdropSeg))) return rc;
	for (int64_t k = 0; k < n_hits; ++k)
		out[k] = dmnd_diagonal_segment{ hits[k].seed_offset - seg[(size_t)k].left, hits[k].subject - (int64_t)seg[(size_t)k].left, seg[(size_t)k].left + seg[(size_t)k].right, seg[(size_t)k].score };
	return DMND_OK;
}

extern "C" int dmnd_set_frameshift(dmnd_ctx* c, int penalty, int range_culling, double range_cover, int channels)
{
	if (!c || penalty < 0 || channels < 1 || range_cover < 0) return fail(DMND_E_ARG, "dmnd_set_frameshift: bad argument");
	if (range_culling && penalty == 0) return fail(DMND_E_ARG, "Query range culling is only supported in frameshift alignment mode (option -F).");
	c->frame_shift = penalty; c->range_culling = range_culling != 0; c->range_cover = range_cover; c->fs_channels = channels;
	return DMND_OK;
}

extern "C" int dmnd_set_comp_based_stats(dmnd_ctx* c, int mode)
{
	if (!c || mode < 0 || mode > 5) return fail(DMND_E_ARG, "Invalid value for --comp-based-stats. Permitted values: 0, 1, 2, 3, 4, 5.");
	c->comp_based_stats = mode;
	return DMND_OK;
}

extern "C" int dmnd_set_sensitivity(dmnd_ctx* c, int sensitivity)
{
	if (!c || sensitivity < DMND_SENS_FAST || sensitivity > DMND_SENS_ULTRA_SENSITIVE) return fail(DMND_E_ARG, "dmnd_set_sensitivity: bad argument");
	c->ranking_block_letters = sensitivity >= DMND_SENS_VERY_SENSITIVE ? 800e6 : 2e9;      // extend.cpp:87
	c->band_mode_fast = sensitivity <= DMND_SENS_SENSITIVE ? 1 : 0;                          // default_ext_mode, extend.cpp:62-75
	return DMND_OK;
}

extern "C" int dmnd_set_extension_mode(dmnd_ctx* c, int mode)
{
	if (!c || mode < DMND_EXT_DEFAULT || mode > DMND_EXT_FULL) return fail(DMND_E_ARG, "dmnd_set_extension_mode: bad argument");
	c->ext_mode = mode;
	return DMND_OK;
}

// ---- global ranking (align/global_ranking/table.cpp) ------------------------------------------------------------------
// get_query_hits_reextend + target_score (table.cpp:88-133): per (query, target) of a block pair's seed hits the best x-drop
// ungapped score over the target's hits -- no composition bias; hits in (diagonal, column) order, a hit skipped when the segment
// computed LAST on its diagonal already reaches it -- and the context it was found in.
extern "C" int dmnd_rank_targets(dmnd_ctx* c, const int8_t* qdata, const int8_t* tdata, const dmnd_seed_hit* hits, int64_t n_hits, int threads,
	dmnd_ranked_target* out, int64_t cap, int64_t* n_out)
{
	if (!c || !qdata || !tdata || (!hits && n_hits) || !n_out || (!out && cap)) return fail(DMND_E_ARG, "dmnd_rank_targets: NULL argument");
	*n_out = 0;
	const std::vector<int64_t>& ql = c->limits[DMND_QUERY];
	const std::vector<int64_t>& tl = c->limits[DMND_TARGET];
	if (ql.size() < 2 || tl.size() < 2) return fail(DMND_E_ARG, "dmnd_rank_targets: blocks must be uploaded with limits");
	HostCfg h;
	make_cfg(c, h);
	h.contexts = c->query_contexts;
	h.evaluer = nullptr;
	const uint32_t C = (uint32_t)h.contexts;
	const std::vector<Range> qr = split_by_query(hits, n_hits, h.contexts);
	std::vector<std::vector<dmnd_ranked_target>> per(qr.size());
	threads = std::max(1, threads);
	const uint32_t* coarse = c->coarse[DMND_TARGET].empty() ? nullptr : c->coarse[DMND_TARGET].data();
	parallel_for(qr.size(), threads, [&](size_t i, int) {
		QueryWork w;
		const uint32_t query = hits[qr[i].b].query / C;
		load_query(h, w, query, hits + qr[i].b, hits + qr[i].e, nullptr, tl.data(), (int64_t)tl.size() - 1, coarse);
		std::vector<HostSeedHit>& sh = w.sh;
		for (const TargetGroup& g : w.groups) {
			std::sort(sh.begin() + (ptrdiff_t)g.begin, sh.begin() + (ptrdiff_t)g.end, [](const HostSeedHit& a, const HostSeedHit& b) {
				const int d1 = a.i - a.j, d2 = b.i - b.j;
				return d1 < d2 || (d1 == d2 && a.j < b.j);
			});
			const SeqRef t{ tdata + tl[g.target], (int)(tl[g.target + 1] - tl[g.target] - 1) };
			auto walk = [&](const HostSeedHit& x) {
				const uint32_t qc = query * C + (uint32_t)x.frame;
				return xdrop_ungapped(h.S, SeqRef{ qdata + ql[qc], (int)(ql[qc + 1] - ql[qc] - 1) }, nullptr, t, x.i, x.j, h.xdrop);
			};
			Seg d = walk(sh[g.begin]);
			int score = d.score, context = sh[g.begin].frame;
			for (size_t x = g.begin + 1; x < g.end; ++x) {
				if (d.diag() == sh[x].i - sh[x].j && d.j_end() >= sh[x].j) continue;
				d = walk(sh[x]);
				if (d.score > score) { score = d.score; context = sh[x].frame; }
			}
			per[i].push_back(dmnd_ranked_target{ query, g.target, (uint16_t)score, (uint8_t)context, 0 });
		}
	});
	int64_t n = 0;
	for (const auto& v : per) n += (int64_t)v.size();
	*n_out = n;
	if (n > cap) return fail(DMND_E_CAP, "dmnd_rank_targets: record buffer too small");
	int64_t o = 0;
	for (const auto& v : per) { std::copy(v.begin(), v.end(), out + o); o += (int64_t)v.size(); }
	return DMND_OK;
}

// merge_hits (table.cpp:135-151): the records of one block pair into the table of the best n targets of every query -- per target its
// best score, the table in (score descending, target ascending) order (Hit::operator<), empty entries have score 0
extern "C" int dmnd_rank_update(dmnd_ranked_target* table, int64_t n_queries, int n, const dmnd_ranked_target* recs, int64_t n_recs)
{
	if (!table || n_queries < 0 || n < 1 || (!recs && n_recs) || n_recs < 0) return fail(DMND_E_ARG, "dmnd_rank_update: bad argument");
	std::vector<dmnd_ranked_target> hits, merged;
	for (int64_t b = 0; b < n_recs;) {
		int64_t e = b;
		while (e < n_recs && recs[e].query == recs[b].query) ++e;
		const uint32_t q = recs[b].query;
		if ((int64_t)q >= n_queries) return fail(DMND_E_ARG, "dmnd_rank_update: query id outside the table");
		dmnd_ranked_target* row = table + (size_t)q * (size_t)n;
		int count = n;
		while (count > 0 && row[count - 1].score == 0) --count;
		hits.assign(recs + b, recs + e);
		hits.insert(hits.end(), row, row + count);
		std::sort(hits.begin(), hits.end(), [](const dmnd_ranked_target& x, const dmnd_ranked_target& y) { return x.target < y.target || (x.target == y.target && x.score > y.score); });
		merged.clear();
		for (const dmnd_ranked_target& x : hits) if (merged.empty() || merged.back().target != x.target) merged.push_back(x);
		std::sort(merged.begin(), merged.end(), [](const dmnd_ranked_target& x, const dmnd_ranked_target& y) { return x.score > y.score || (x.score == y.score && x.target < y.target); });
		const size_t keep = std::min((size_t)n, merged.size());
		for (size_t k = 0; k < keep; ++k) { row[k] = merged[k]; row[k].query = q; }
		b = e;
	}
	return DMND_OK;
}

extern "C" int dmnd_set_global_ranking(dmnd_ctx* c, int n)
{
	if (!c || n < 0) return fail(DMND_E_ARG, "dmnd_set_global_ranking: bad argument");
	c->global_ranking = n;
	return DMND_OK;
}

extern "C" int dmnd_set_max_hsps(dmnd_ctx* c, int n)
{
	if (!c || n < 0) return fail(DMND_E_ARG, "dmnd_set_max_hsps: bad argument");
	c->max_hsps = n;
	return DMND_OK;
}

extern "C" int dmnd_set_max_target_seqs(dmnd_ctx* c, int k)
{
	if (!c || k < 1) return fail(DMND_E_ARG, "dmnd_set_max_target_seqs: bad argument");
	c->max_target_seqs = k;
	return DMND_OK;
}

extern "C" int dmnd_set_filters(dmnd_ctx* c, double min_id, double query_cover, double subject_cover, double min_bit_score)
{
	if (!c || min_id < 0 || min_id > 100 || query_cover < 0 || query_cover > 100 || subject_cover < 0 || subject_cover > 100 || min_bit_score < 0)
		return fail(DMND_E_ARG, "dmnd_set_filters: bad argument");
	c->min_id = min_id; c->query_cover = query_cover; c->subject_cover = subject_cover; c->min_bit_score = min_bit_score;
	return DMND_OK;
}

extern "C" int dmnd_set_query_source_lengths(dmnd_ctx* c, const int32_t* lengths, int64_t n_queries)
{
	if (!c || n_queries < 0 || (n_queries > 0 && !lengths)) return fail(DMND_E_ARG, "dmnd_set_query_source_lengths: bad argument");
	for (int64_t i = 0; i < n_queries; ++i) if (lengths[i] < 1) return fail(DMND_E_ARG, "dmnd_set_query_source_lengths: a read length below 1");
	c->source_lens.assign(lengths, lengths + n_queries);
	return DMND_OK;
}

extern "C" int dmnd_set_no_self_hits(dmnd_ctx* c, dmnd_same_title_fn same_title, void* user)
{
	if (!c) return fail(DMND_E_ARG, "dmnd_set_no_self_hits: ctx is NULL");
	c->same_title = same_title; c->same_title_user = user;
	return DMND_OK;
}

extern "C" int dmnd_set_top_percent(dmnd_ctx* c, double percent)
{
	if (!c || percent > 100.0) return fail(DMND_E_ARG, "dmnd_set_top_percent: bad argument");
	c->top_percent = percent < 0.0 ? -1.0 : percent;
	return DMND_OK;
}

// [0] round-1 DpTargets [1] round-2 DpTargets [2] round-1 cells [3] round-2 cells (DpTarget::cells, dp/dp.h:121-124)
// host wall ms: [4] Hauser+upload [5] chaining/plan [6] round-1 call [7] culling [8] round-2 call
// device ms: [9] round-1 swipe kernels [10] round-2 swipe kernels [11] traceback kernel
extern "C" int dmnd_extend_stats(const dmnd_ctx* c, double out[12])
{
	if (!c || !out) return fail(DMND_E_ARG, "dmnd_extend_stats: NULL argument");
	for (int i = 0; i < 12; ++i) out[i] = c->ext_stats[i];
	return DMND_OK;
}

// First-call allocations of dmnd_extend made ahead of it (a driver calls this beside its upload / masking phase): the device work
// arrays of the x-drop stage, the planner and the device half for about n_hits_hint seed hits, and the page-locked result buffer.
// A hint that turns out too small only means the call itself grows the buffers, as it would have without this.
extern "C" int dmnd_extend_reserve(dmnd_ctx* c, int64_t n_hits_hint)
{
	if (!c || n_hits_hint < 0) return fail(DMND_E_ARG, "dmnd_extend_reserve: bad argument");
	if (n_hits_hint == 0) return DMND_OK;
	HIP_TRY(hipSetDevice(c->device));
	const size_t n = (size_t)n_hits_hint, nq = c->limits[DMND_QUERY].size() > 1 ? c->limits[DMND_QUERY].size() - 1 : n;
	const size_t n_rec = std::min(n, nq * (size_t)std::max(c->max_target_seqs, 1));
	if (int rc = c->xd_hits.ensure(n * sizeof(dmnd_seed_hit))) return rc;
	if (int rc = c->xd_out.ensure(n * sizeof(XdropSeg))) return rc;
	if (int rc = c->plan_dev.ensure(n * 112 + 8192)) return rc;            // (the planner's arrays: ~100 bytes per hit, plan_on_device)
	if (int rc = c->plan_host.ensure(sizeof(PlanCounters))) return rc;
	if (int rc = c->ext_dev.ensure(n * 440 + nq + n_rec * sizeof(dmnd_match) + 16384)) return rc;      // (~420 bytes per band, extend_on_device)
	if (int rc = c->ext_host.ensure(sizeof(ExtCounters) + nq + n_rec * sizeof(dmnd_match) + 256)) return rc;
	return DMND_OK;
}

extern "C" int dmnd_extend_records_device(const dmnd_ctx* c, const dmnd_match** records_dev, int64_t* n)
{
	if (!c || !records_dev || !n) return fail(DMND_E_ARG, "dmnd_extend_records_device: NULL argument");
	*records_dev = c->ext_records_n >= 0 ? c->ext_records_dev : nullptr;
	*n = c->ext_records_n;
	return DMND_OK;
}

extern "C" int dmnd_extend_device_stats(const dmnd_ctx* c, double out[10])
{
	if (!c || !out) return fail(DMND_E_ARG, "dmnd_extend_device_stats: NULL argument");
	for (int i = 0; i < 10; ++i) out[i] = c->ext_dev_stats[i];
	return DMND_OK;
}

extern "C" int dmnd_extend_plan_stats(const dmnd_ctx* c, double out[3])
{
	if (!c || !out) return fail(DMND_E_ARG, "dmnd_extend_plan_stats: NULL argument");
	for (int i = 0; i < 3; ++i) out[i] = c->ext_plan_stats[i];
	return DMND_OK;
}

extern "C" int dmnd_set_query_contexts(dmnd_ctx* c, int contexts)
{
	if (!c || (contexts != 1 && contexts != 6)) return fail(DMND_E_ARG, "dmnd_set_query_contexts: contexts must be 1 (blastp) or 6 (blastx)");
	c->query_contexts = contexts;
	return DMND_OK;
}

