// extend_host.hip -- the extension stage above the GPU Smith-Waterman: what Extension::extend does per query
// (/root/reference/src/align/extend.cpp:226-420), re-organised for the GPU as a block-wide batch:
//   all queries:  load_hits -> x-drop ungapped -> chaining -> band construction        (host threads)
//   ONE launch :  round-1 score-only banded swipe over every DpTarget of the block     (GPU)
//   all queries:  e-value cutoff, per-target best HSP, top-k culling                   (host)
//   ONE launch :  round-2 banded swipe with traceback / statistics                     (GPU)
//   all queries:  final culling -> match records -> BLAST tabular text                 (host)
// instead of the reference's per-query calls of DP::BandedSwipe::swipe from a thread pool.
// Reference pieces restated here (chaining itself is in chain_host.h):
//   HauserCorrection                        src/stats/hauser_correction.cpp:53-109
//   load_hits                               src/align/load_hits.h:44-127
//   ranking_chunk_size                      src/align/extend.cpp:79-92
//   ungapped_stage                          src/align/ungapped.cpp:62-126
//   Extension::band, add_dp_targets         src/align/gapped_score.cpp:41-180
//   DP::BandedSwipe::bin                    src/dp/swipe/swipe_wrapper.cpp:75-102
//   Target::add_hit / inner_culling, culling, output_range   src/align/target.h:97-113, culling.cpp:37-113,189-203
//   round-2 add_dp_targets / align          src/align/gapped_final.cpp:66-160
//   blast tab fields                        src/output/blast_tab_format.cpp
// Scope: blastp, one query context, max_hsps = 1, Hauser composition bias (comp-based-stats 1), no gapped filter,
// no id/coverage filters, single ranking chunk (a query with more targets than the ranking chunk fails loudly).
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <atomic>
#include <thread>
#include <vector>
#include "ctx.h"
#include "chain_host.h"

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
	int cbs_window = 40;                 // config.cbs_window
	int max_target_seqs = 25;
	int64_t max_swipe_dp = 1000000;      // config.max_swipe_dp
	int band_mode_fast = 1;              // Extension::Mode::BANDED_FAST for every sensitivity up to --sensitive
	double ref_letters = 0;
};

void make_cfg(const dmnd_ctx* c, HostCfg& h)
{
	for (int i = 0; i < 32 * 32; ++i) h.S.m[i] = c->params.matrix8[i];
	h.S.gap_open = c->params.gap_open; h.S.gap_extend = c->params.gap_extend;
	for (int i = 0; i < 20; ++i) {                     // ScoreMatrix::init_background_scores, score_matrix.cpp:241-248
		h.background_scores[i] = 0;
		for (int j = 0; j < 20; ++j) h.background_scores[i] += BLOSUM62_BG[j] * h.S.at(i, j);
	}
}

// HauserCorrection(seq): sliding-window expected-score bias per query position, rounded to int8
void hauser_int8(const HostCfg& h, const SeqRef& seq, int8_t* out)
{
	const unsigned l = (unsigned)seq.len, window = (unsigned)h.cbs_window, window_half = std::min(window / 2, l - 1);
	std::vector<float> f(l, 0.0f);
	int scores[20] = { 0 };
	auto add = [&](int letter, int sign) { for (int i = 0; i < 20; ++i) scores[i] += sign * h.S.at(letter, i); };
	auto emit = [&](unsigned m, unsigned n) {
		const int r = seq[(int)m];
		if (r < 20) f[m] = (float)h.background_scores[r] - float(scores[r] - h.S.at(r, r)) / (n - 1);
	};
	unsigned n = 0, hh = 0, m = 0, t = 0;
	while (n < window_half && hh < l) { ++n; add(seq[(int)hh], 1); ++hh; }
	while (n < window + 1 && hh < l) { ++n; add(seq[(int)hh], 1); emit(m, n); ++hh; ++m; }
	while (hh < l) { add(seq[(int)hh], 1); add(seq[(int)t], -1); emit(m, n); ++hh; ++t; ++m; }
	while (m < l && n > window_half + 1) { --n; add(seq[(int)t], -1); emit(m, n); ++t; ++m; }
	while (m < l) { emit(m, n); ++m; }
	for (unsigned i = 0; i < l; ++i) out[i] = (int8_t)(f[i] < 0.0f ? f[i] - 0.5f : f[i] + 0.5f);
}

int band_for(int len, bool fast)                            // Extension::band, gapped_score.cpp:41-73
{
	if (fast) return len < 50 ? 12 : len < 100 ? 16 : len < 250 ? 30 : len < 350 ? 40 : 64;
	return len < 50 ? 15 : len < 100 ? 20 : len < 150 ? 30 : len < 200 ? 50 : len < 250 ? 60 : len < 350 ? 100 : len < 500 ? 120 : 150;
}

int64_t ranking_chunk_size(double ref_letters, int max_target_seqs)       // extend.cpp:79-92, default options
{
	const int64_t block_mult = std::max((int64_t)std::llround(ref_letters / 2e9), (int64_t)1);
	const int64_t m32 = ((int64_t)max_target_seqs + 31) / 32 * 32;
	return std::max((int64_t)128, std::min(m32, (int64_t)400)) * block_mult;
}

struct PlanTarget { uint32_t query, target; int32_t d_begin, d_end, ungapped_score; };

struct QueryPlan {
	std::vector<PlanTarget> dp;          // round-1 DpTargets of one query, reference order
};

// hits of ONE query -> its round-1 DpTargets
int plan_query(const HostCfg& h, ChainWorkspace& ws, uint32_t query, const dmnd_seed_hit* hb, const dmnd_seed_hit* he,
	const int8_t* qdata, const int64_t* ql, const int8_t* tdata, const int64_t* tl, int64_t nt, const int8_t* cbs_all,
	std::vector<PlanTarget>& out, std::string& err)
{
	std::vector<dmnd_seed_hit> hits(hb, he);
	std::sort(hits.begin(), hits.end(), [](const dmnd_seed_hit& a, const dmnd_seed_hit& b) {       // Hit::CmpSubject
		return a.subject < b.subject || (a.subject == b.subject && (a.query < b.query || (a.query == b.query && a.seed_offset < b.seed_offset)));
	});
	const SeqRef q{ qdata + ql[query], (int)(ql[query + 1] - ql[query] - 1) };
	const int8_t* cbs = cbs_all ? cbs_all + ql[query] : nullptr;
	// group by target (load_hits)
	struct TG { uint32_t target; size_t begin, end; int score; };
	std::vector<TG> groups;
	std::vector<HostSeedHit> sh(hits.size());
	const int64_t* it = tl;
	for (size_t x = 0; x < hits.size(); ++x) {
		const int64_t s = hits[x].subject;
		it = std::upper_bound(it, tl + nt + 1, s);
		const uint32_t t = (uint32_t)(it - tl) - 1;
		--it;
		if (groups.empty() || groups.back().target != t) groups.push_back(TG{ t, x, x, 0 });
		sh[x] = HostSeedHit{ hits[x].seed_offset, (int)(s - tl[t]), hits[x].score };
		groups.back().end = x + 1;
		groups.back().score = std::max(groups.back().score, (int)(uint16_t)hits[x].score);
	}
	if ((int64_t)groups.size() > ranking_chunk_size(h.ref_letters, h.max_target_seqs)) {
		err = "query " + std::to_string(query) + " has " + std::to_string(groups.size()) + " seed-hit targets: ranking chunks are not implemented";
		return DMND_E_ARG;
	}
	std::vector<Seg> segs;
	std::vector<Chain> chains;
	const int base_band = band_for(q.len, h.band_mode_fast != 0);
	for (const TG& g : groups) {
		const SeqRef t{ tdata + tl[g.target], (int)(tl[g.target + 1] - tl[g.target] - 1) };
		std::sort(sh.begin() + (ptrdiff_t)g.begin, sh.begin() + (ptrdiff_t)g.end, [](const HostSeedHit& a, const HostSeedHit& b) {
			const int d1 = a.i - a.j, d2 = b.i - b.j;
			return d1 < d2 || (d1 == d2 && a.j < b.j);
		});
		segs.clear();
		int ungapped = 0;
		for (size_t x = g.begin; x < g.end; ++x) {
			ungapped = std::max(ungapped, sh[x].score);
			if (!segs.empty() && segs.back().diag() == sh[x].i - sh[x].j && segs.back().j_end() >= sh[x].j) continue;
			const Seg d = xdrop_ungapped(h.S, q, cbs, t, sh[x].i, sh[x].j, ws.cfg.xdrop);
			if (d.score > 0) segs.push_back(d);
		}
		if (segs.empty()) continue;
		std::stable_sort(segs.begin(), segs.end(), [](const Seg& a, const Seg& b) { return a.diag() < b.diag() || (a.diag() == b.diag() && a.j < b.j); });
		ws.run(h.S, q, t, segs, chains);
		std::stable_sort(chains.begin(), chains.end(), [](const Chain& a, const Chain& b) { return a.d_min < b.d_min; });
		// add_dp_targets: merge overlapping bands of the target's chains
		int d0 = INT_MAX, d1 = INT_MIN;
		for (const Chain& c : chains) {
			const int b0 = std::max(c.d_min - base_band, -(t.len - 1)), b1 = std::min(c.d_max + 1 + base_band, q.len);
			const int lo = std::max(d0, b0), hi = std::min(d1, b1);
			const double overlap = hi > lo ? hi - lo : 0;
			// (d1 - d0) wraps for the initial (INT_MAX, INT_MIN) pair exactly as in the reference: the first chain never merges
			const double w = (double)(int)((unsigned)d1 - (unsigned)d0);
			if (overlap / w > 0.0 || overlap / (b1 - b0) > 0.0) { d0 = std::min(d0, b0); d1 = std::max(d1, b1); }
			else {
				if (d0 != INT_MAX) out.push_back(PlanTarget{ query, g.target, d0, d1, ungapped });
				d0 = b0; d1 = b1;
			}
		}
		if (d0 != INT_MAX) out.push_back(PlanTarget{ query, g.target, d0, d1, ungapped });
	}
	return DMND_OK;
}

struct Range { size_t b, e; };

std::vector<Range> split_by_query(const dmnd_seed_hit* hits, int64_t n)
{
	std::vector<Range> r;
	for (int64_t i = 0; i < n;) {
		int64_t j = i;
		while (j < n && hits[j].query == hits[i].query) ++j;
		r.push_back(Range{ (size_t)i, (size_t)j });
		i = j;
	}
	return r;
}

template<typename F>
void parallel_for(size_t n, int threads, F f)
{
	threads = std::max(1, std::min<int>(threads, (int)n));
	if (threads == 1) { for (size_t i = 0; i < n; ++i) f(i, 0); return; }
	std::vector<std::thread> th;
	std::atomic<size_t> next(0);
	for (int t = 0; t < threads; ++t)
		th.emplace_back([&, t] { size_t i; while ((i = next.fetch_add(1)) < n) f(i, t); });
	for (auto& x : th) x.join();
}

int plan_all(const HostCfg& h, int threads, const dmnd_seed_hit* hits, int64_t n_hits,
	const int8_t* qdata, const std::vector<int64_t>& ql, const int8_t* tdata, const std::vector<int64_t>& tl,
	const int8_t* cbs_all, std::vector<PlanTarget>& out)
{
	const std::vector<Range> qr = split_by_query(hits, n_hits);
	std::vector<std::vector<PlanTarget>> per(qr.size());
	threads = std::max(1, threads);
	std::vector<ChainWorkspace> ws((size_t)threads);
	std::vector<std::string> errs((size_t)threads);
	std::vector<int> rcs((size_t)threads, 0);
	parallel_for(qr.size(), threads, [&](size_t i, int t) {
		const int rc = plan_query(h, ws[(size_t)t], hits[qr[i].b].query, hits + qr[i].b, hits + qr[i].e, qdata, ql.data(), tdata, tl.data(),
			(int64_t)tl.size() - 1, cbs_all, per[i], errs[(size_t)t]);
		if (rc) rcs[(size_t)t] = rc;
	});
	for (size_t t = 0; t < rcs.size(); ++t) if (rcs[t]) return fail(rcs[t], errs[t]);
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
	const int8_t* tdata, const int64_t* tlimits, int64_t nt, const dmnd_seed_hit* hits, int64_t n_hits, int threads,
	int8_t* cbs_out, dmnd_plan_target* out, int64_t cap, int64_t* n_out)
{
	if (!params || !qdata || !qlimits || !tdata || !tlimits || (!hits && n_hits) || !n_out) return fail(DMND_E_ARG, "dmnd_extend_plan: NULL argument");
	dmnd_ctx tmp;
	tmp.params = *params;
	HostCfg h;
	make_cfg(&tmp, h);
	h.ref_letters = (double)(tlimits[nt] - tlimits[0] - nt);
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

struct Cand {              // one target of one query after round 1 (Extension::Target with max_hsps = 1)
	uint32_t target;
	int score, d_begin, d_end, ungapped;
	double evalue;
};

}

// The whole extension stage for one (query block, reference block) pair on the uploaded blocks.
extern "C" int dmnd_extend(dmnd_ctx* c, const int8_t* qdata, const int8_t* tdata, const dmnd_seed_hit* hits, int64_t n_hits,
	int threads, uint32_t hsp_values, dmnd_match* out, int64_t cap, int64_t* n_out,
	uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used)
{
	if (!c || !qdata || !tdata || (!hits && n_hits) || !n_out) return fail(DMND_E_ARG, "dmnd_extend: NULL argument");
	*n_out = 0;
	if (transcript_used) *transcript_used = 0;
	const std::vector<int64_t>& ql = c->limits[DMND_QUERY];
	const std::vector<int64_t>& tl = c->limits[DMND_TARGET];
	if (ql.size() < 2 || tl.size() < 2) return fail(DMND_E_ARG, "dmnd_extend: blocks must be uploaded with limits");
	HostCfg h;
	make_cfg(c, h);
	h.max_target_seqs = c->max_target_seqs;
	h.ref_letters = (double)(tl.back() - tl.front() - ((int64_t)tl.size() - 1));
	if (hsp_values == 0) hsp_values = 510;
	for (double& x : c->ext_stats) x = 0;
	auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	auto cells_of = [](const std::vector<dmnd_dp_target>& v) {
		double s = 0;
		for (const auto& d : v) s += (double)dmnd_banded_cols(d.query_len, d.target_len, d.d_begin, d.d_end) * (double)(d.d_end - d.d_begin);
		return s;
	};
	const double t0 = now();
	// 1. Hauser bias for every query, resident next to the query block
	std::vector<int8_t> cbs;
	all_hauser(h, threads, qdata, ql, cbs);
	if (int rc = dmnd_upload_cbs(c, cbs.data(), ql.back())) return rc;
	const double t1 = now();
	// 2. plan
	std::vector<PlanTarget> plan;
	if (int rc = plan_all(h, threads, hits, n_hits, qdata, ql, tdata, tl, cbs.data(), plan)) return rc;
	const double t2 = now();
	c->ext_stats[4] = t1 - t0; c->ext_stats[5] = t2 - t1;
	if (plan.empty()) return DMND_OK;
	// 3. round 1: score only
	std::vector<dmnd_dp_target> items(plan.size());
	for (size_t i = 0; i < plan.size(); ++i) {
		const PlanTarget& p = plan[i];
		items[i] = dmnd_dp_target{ ql[p.query], tl[p.target], ql[p.query], (int32_t)(ql[p.query + 1] - ql[p.query] - 1),
			(int32_t)(tl[p.target + 1] - tl[p.target] - 1), p.d_begin, p.d_end };
	}
	std::vector<dmnd_hsp> r1(plan.size());
	if (int rc = dmnd_banded_swipe(c, items.data(), (int64_t)items.size(), DMND_SWIPE_SCORE, 0, r1.data(), nullptr, 0, nullptr)) return rc;
	const double sw1 = c->swipe_ms;
	const double t3 = now();
	c->ext_stats[0] = (double)items.size(); c->ext_stats[2] = cells_of(items); c->ext_stats[6] = t3 - t2; c->ext_stats[9] = sw1;
	// 4. per query: report cutoff, best HSP per target (Target::add_hit + inner_culling with max_hsps = 1), top-k culling
	std::vector<Cand> survivors;
	std::vector<uint32_t> surv_query;
	for (size_t i = 0; i < plan.size();) {
		size_t j = i;
		std::vector<Cand> cands;
		while (j < plan.size() && plan[j].query == plan[i].query) {
			const PlanTarget& p = plan[j];
			const int score = r1[j].score;
			const int qlen = items[j].query_len, tlen = items[j].target_len;
			if (score > 0) {
				const double ev = c->evaluer.evalue(score, (unsigned)qlen, (unsigned)tlen);
				if (ev <= c->params.max_evalue) {
					// add_hit: a later HSP of the same target replaces the filter values only with a strictly higher score;
					// inner_culling keeps the best HSP by (score desc, d_begin asc) (Hsp::operator<, basic/match.h:199)
					if (!cands.empty() && cands.back().target == p.target) {
						Cand& k = cands.back();
						if (score > k.score || (score == k.score && p.d_begin < k.d_begin)) { k.score = score; k.evalue = ev; k.d_begin = p.d_begin; k.d_end = p.d_end; }
					}
					else cands.push_back(Cand{ p.target, score, p.d_begin, p.d_end, p.ungapped_score, ev });
				}
			}
			++j;
		}
		std::sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) {        // Target::comp_evalue
			return a.evalue < b.evalue || (a.evalue == b.evalue && (a.score > b.score || (a.score == b.score && a.target < b.target)));
		});
		if ((int)cands.size() > h.max_target_seqs) cands.resize((size_t)h.max_target_seqs);
		for (const Cand& k : cands) { survivors.push_back(k); surv_query.push_back(plan[i].query); }
		i = j;
	}
	const double t4 = now();
	c->ext_stats[7] = t4 - t3;
	if (survivors.empty()) return DMND_OK;
	// 5. round 2: traceback for DP sizes <= max_swipe_dp, statistics passes above (DP::BandedSwipe::bin)
	std::vector<dmnd_dp_target> it_tb, it_st;
	std::vector<size_t> idx_tb, idx_st;
	for (size_t i = 0; i < survivors.size(); ++i) {
		const Cand& k = survivors[i];
		const uint32_t q = surv_query[i];
		dmnd_dp_target d{ ql[q], tl[k.target], ql[q], (int32_t)(ql[q + 1] - ql[q] - 1), (int32_t)(tl[k.target + 1] - tl[k.target] - 1), k.d_begin, k.d_end };
		const int64_t dp_size = (int64_t)dmnd_banded_cols(d.query_len, d.target_len, d.d_begin, d.d_end) * (int64_t)(d.d_end - d.d_begin);
		if (dp_size > h.max_swipe_dp) { it_st.push_back(d); idx_st.push_back(i); }
		else { it_tb.push_back(d); idx_tb.push_back(i); }
	}
	std::vector<dmnd_hsp> r2(survivors.size());
	std::vector<dmnd_hsp> tmp;
	int64_t used = 0;
	std::vector<uint8_t> own_arena;
	uint8_t* arena = transcript;
	int64_t arena_cap = transcript_cap;
	if (!arena) {
		int64_t need = 16;
		for (const auto& d : it_tb) need += (int64_t)d.query_len + d.target_len + 2;
		own_arena.resize((size_t)need);
		arena = own_arena.data(); arena_cap = need;
	}
	double sw2 = 0, tb2 = 0;
	if (!it_tb.empty()) {
		tmp.resize(it_tb.size());
		if (int rc = dmnd_banded_swipe(c, it_tb.data(), (int64_t)it_tb.size(), DMND_SWIPE_TRACEBACK, hsp_values, tmp.data(), arena, arena_cap, &used)) return rc;
		for (size_t x = 0; x < idx_tb.size(); ++x) r2[idx_tb[x]] = tmp[x];
		sw2 += c->swipe_ms; tb2 += c->traceback_ms;
	}
	if (!it_st.empty()) {
		tmp.resize(it_st.size());
		if (int rc = dmnd_banded_swipe(c, it_st.data(), (int64_t)it_st.size(), DMND_SWIPE_STATS, hsp_values, tmp.data(), nullptr, 0, nullptr)) return rc;
		for (size_t x = 0; x < idx_st.size(); ++x) { r2[idx_st[x]] = tmp[x]; r2[idx_st[x]].transcript_len = 0; r2[idx_st[x]].transcript_off = -1; }
		sw2 += c->swipe_ms;
	}
	c->swipe_ms = sw1 + sw2; c->traceback_ms = tb2;
	const double t5 = now();
	c->ext_stats[1] = (double)survivors.size(); c->ext_stats[3] = cells_of(it_tb) + cells_of(it_st); c->ext_stats[8] = t5 - t4;
	c->ext_stats[10] = sw2; c->ext_stats[11] = tb2;
	if (transcript_used) *transcript_used = transcript ? used : 0;
	// 6. final per-query culling (Match::cmp_evalue + output_range) -> records
	int64_t n = 0;
	for (size_t i = 0; i < survivors.size();) {
		size_t j = i;
		std::vector<dmnd_match> ms;
		while (j < survivors.size() && surv_query[j] == surv_query[i]) {
			const Cand& k = survivors[j];
			const dmnd_hsp& hsp = r2[j];
			const uint32_t q = surv_query[j];
			const int qlen = (int)(ql[q + 1] - ql[q] - 1), tlen = (int)(tl[k.target + 1] - tl[k.target] - 1);
			if (hsp.score > 0) {
				const double ev = c->evaluer.evalue(hsp.score, (unsigned)qlen, (unsigned)tlen);
				if (ev <= c->params.max_evalue) {
					dmnd_match m;
					m.query = q; m.target = k.target; m.evalue = ev; m.bit_score = c->evaluer.bitscore(hsp.score);
					m.ungapped_score = k.ungapped; m.d_begin = k.d_begin; m.d_end = k.d_end; m.hsp = hsp;
					if (!transcript) m.hsp.transcript_off = -1;
					ms.push_back(m);
				}
			}
			++j;
		}
		std::sort(ms.begin(), ms.end(), [](const dmnd_match& a, const dmnd_match& b) {     // Match::cmp_evalue
			return a.evalue < b.evalue || (a.evalue == b.evalue && (a.hsp.score > b.hsp.score || (a.hsp.score == b.hsp.score && a.target < b.target)));
		});
		if ((int)ms.size() > h.max_target_seqs) ms.resize((size_t)h.max_target_seqs);
		for (const dmnd_match& m : ms) {
			if (n < cap && out) out[n] = m;
			++n;
		}
		i = j;
	}
	*n_out = n;
	if (n > cap) return fail(DMND_E_CAP, "dmnd_extend: match buffer too small");
	return DMND_OK;
}

extern "C" int dmnd_set_max_target_seqs(dmnd_ctx* c, int k)
{
	if (!c || k < 1) return fail(DMND_E_ARG, "dmnd_set_max_target_seqs: bad argument");
	c->max_target_seqs = k;
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

// BLAST tabular line of one match (qseqid sseqid pident length mismatch gapopen qstart qend sstart send evalue bitscore),
// formatted as the reference prints it (src/output/blast_tab_format.cpp; util/text_buffer.h:238-260).
extern "C" int dmnd_format_tab(const dmnd_match* m, const char* qseqid, const char* sseqid, char* buf, int64_t cap)
{
	if (!m || !qseqid || !sseqid || !buf) return fail(DMND_E_ARG, "dmnd_format_tab: NULL argument");
	const dmnd_hsp& h = m->hsp;
	// Util::String::format_double (util/string/string.h:87-92): >= 100 -> floor, else one rounded decimal
	auto fd = [](double x, char* p, size_t n) {
		if (x >= 100.0) std::snprintf(p, n, "%lli", (long long)std::floor(x));
		else { const long long i = std::llround(x * 10.0); std::snprintf(p, n, "%lli.%lli", i / 10, i % 10); }
	};
	char pid[64], ev[64], bs[64];
	fd((double)h.identities * 100.0 / (double)h.length, pid, sizeof pid);     // Hsp::id_percent
	if (m->evalue == 0.0) std::snprintf(ev, sizeof ev, "0.0");                 // TextBuffer::print_e
	else std::snprintf(ev, sizeof ev, "%.2e", m->evalue);
	fd(m->bit_score, bs, sizeof bs);
	const int w = std::snprintf(buf, (size_t)cap, "%s\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%s\t%s\n", qseqid, sseqid, pid, h.length,
		h.mismatches, h.gap_openings, h.q_begin + 1, h.q_end, h.s_begin + 1, h.s_end, ev, bs);
	return w < cap ? w : DMND_E_CAP;
}
