// gapped_api.hip -- C ABI of the gapped filter: dmnd_set_gapped_filter, dmnd_gapped_filter.
// Replaces Extension::gapped_filter (src/align/gapped_filter.cpp:80-109) for a whole batch of seed hits.
#include <cmath>
#include <cstdlib>
#include <vector>
#include "ctx.h"
#include "gapped_kernels.h"

using namespace dmnd;

namespace {

// CutoffTable2D(evalue) (util/scores/cutoff_table.h:50-83) over ScoreMatrix::evalue_norm (score_matrix.cpp:222-225):
// smallest raw score in [10, 1000) whose e-value, normalised to a database of 1e9 letters, is <= evalue
void cutoff_table2d(const dmnd_ctx* c, double evalue, int32_t* table)
{
	Evaluer e = c->evaluer;
	e.db_letters = 1e9;
	for (int i = 0; i < 32 * 32; ++i) table[i] = 0;
	for (int b1 = 1; b1 <= 31; ++b1)
		for (int b2 = 1; b2 <= 31; ++b2) {
			int r = 1000;
			for (int i = 10; i < 1000; ++i)
				if (e.evalue(i, 1u << (b1 - 1), 1u << (b2 - 1)) <= evalue) { r = i; break; }
			table[b1 * 32 + b2] = r;
		}
}

}

extern "C" int dmnd_set_gapped_filter(dmnd_ctx* c, double evalue)
{
	if (!c || !(evalue >= 0.0)) return fail(DMND_E_ARG, "dmnd_set_gapped_filter: bad argument");
	HIP_TRY(hipSetDevice(c->device));
	c->gapped_filter_evalue = evalue;
	if (evalue == 0.0) return DMND_OK;
	std::vector<int32_t> t(2 * 32 * 32);
	cutoff_table2d(c, 2000.0, t.data());                    // config.gapped_filter_evalue1, basic/config.cpp:567
	cutoff_table2d(c, evalue, t.data() + 32 * 32);          // Search::Config::gapped_filter_evalue, run/double_indexed.cpp:301-305
	if (int rc = c->gf_tables.ensure(t.size() * sizeof(int32_t))) return rc;
	HIP_TRY(copy_now(c->stream, c->gf_tables.p, t.data(), t.size() * sizeof(int32_t), hipMemcpyHostToDevice));
	return DMND_OK;
}

extern "C" int dmnd_gapped_filter(dmnd_ctx* c, const dmnd_seed_hit* hits, int64_t n_hits, int use_cbs_flag, uint8_t* flags, int32_t* scores)
{
	if (!flags && n_hits) return fail(DMND_E_ARG, "dmnd_gapped_filter: NULL argument");
	return dmnd_gapped_filter_on(c, hits, nullptr, n_hits, use_cbs_flag, flags, scores);
}

// hits_dev != NULL: the same hits are in HBM already (dmnd_extend's x-drop stage uploaded them) -- no second upload;
// flags == NULL: the flags stay in ctx->gf_flags for the device planner, nothing is copied back
int dmnd_gapped_filter_on(dmnd_ctx* c, const dmnd_seed_hit* hits, const dmnd_seed_hit* hits_dev, int64_t n_hits, int use_cbs_flag, uint8_t* flags, int32_t* scores)
{
	if (!c || (!hits && n_hits)) return fail(DMND_E_ARG, "dmnd_gapped_filter: NULL argument");
	if (n_hits < 0) return fail(DMND_E_ARG, "dmnd_gapped_filter: negative count");
	if (c->gapped_filter_evalue <= 0.0) return fail(DMND_E_ARG, "dmnd_gapped_filter: filter is off (dmnd_set_gapped_filter)");
	if (!c->block[DMND_QUERY].p || !c->block[DMND_TARGET].p || c->limits[DMND_QUERY].size() < 2 || c->limits[DMND_TARGET].size() < 2)
		return fail(DMND_E_ARG, "dmnd_gapped_filter: blocks must be uploaded with limits");
	const bool use_cbs = use_cbs_flag != 0;
	if (use_cbs && c->cbs_len < c->limits[DMND_QUERY].back()) return fail(DMND_E_ARG, "dmnd_gapped_filter: query bias not uploaded (dmnd_upload_cbs)");
	c->gf_ms = 0;
	if (n_hits == 0) return DMND_OK;
	HIP_TRY(hipSetDevice(c->device));
	hipStream_t st = c->stream;
	if (!hits_dev) if (int rc = c->gf_hits.ensure((size_t)n_hits * sizeof(dmnd_seed_hit))) return rc;
	if (int rc = c->gf_flags.ensure((size_t)n_hits)) return rc;
	if (scores) if (int rc = c->gf_scores.ensure((size_t)n_hits * 2 * sizeof(int32_t))) return rc;
	if (!hits_dev) HIP_TRY(hipMemcpyAsync(c->gf_hits.p, hits, (size_t)n_hits * sizeof(dmnd_seed_hit), hipMemcpyHostToDevice, st));
	GfArgs a;
	const double LN2 = 0.69314718055994530941723212145818;
	a.p.diag_score = (int32_t)std::ceil((12.0 * LN2 + std::log(c->params.K)) / c->params.lambda);    // rawscore(gapped_filter_diag_bit_score), setup.cpp:368
	a.p.gap_open = c->params.gap_open; a.p.gap_extend = c->params.gap_extend;
	a.p.window2 = 200;                                                                               // config.gapped_filter_window
	a.p.use_cbs = use_cbs ? 1 : 0;
	a.p.contexts = c->query_contexts;
	a.qdata = c->block[DMND_QUERY].as<int8_t>(); a.tdata = c->block[DMND_TARGET].as<int8_t>(); a.cbs = c->cbs.as<int8_t>();
	a.qlimits = c->d_limits[DMND_QUERY].as<int64_t>(); a.tlimits = c->d_limits[DMND_TARGET].as<int64_t>();
	a.n_targets = (int64_t)c->limits[DMND_TARGET].size() - 1;
	a.matrix = c->matrix.as<int8_t>();
	a.cutoff1 = c->gf_tables.as<int32_t>(); a.cutoff2 = a.cutoff1 + 32 * 32;
	a.hits = hits_dev ? hits_dev : c->gf_hits.as<dmnd_seed_hit>(); a.n_hits = n_hits;
	a.flags = c->gf_flags.as<uint8_t>();
	a.scores = scores ? c->gf_scores.as<int32_t>() : nullptr;
	// Units: runs of consecutive hits of one query (the seed stage hands its hits over sorted by query), at most GF_UNIT_HITS each,
	// sorted into classes by the LDS their query's profile needs -- 32 rows of (query length + 2 * GF_PAD) bytes: 24 / 40 / 64 KB --
	// and one class of queries too long for that (matrix path). DMND_GF_PROFILE=0: round 2's one-wavefront-per-hit launch.
	static const bool profile_env = [] { const char* e = getenv("DMND_GF_PROFILE"); return !e || atoi(e) != 0; }();
	const char* off = getenv("DMND_GF_PROFILE_TEST");                // test hook, read per call
	const bool use_profile = profile_env && !(off && off[0] == '0');
	const int widths[4] = { 768, 1280, 2048, 0 };
	std::vector<int2> units[4];
	if (use_profile) {
		const std::vector<int64_t>& ql = c->limits[DMND_QUERY];
		for (int64_t b = 0; b < n_hits;) {
			const uint32_t q = hits[b].query;
			if ((size_t)q + 1 >= ql.size()) return fail(DMND_E_ARG, "dmnd_gapped_filter: hit with a query id outside the block");
			int64_t e = b + 1;
			while (e < n_hits && e - b < GF_UNIT_HITS && hits[e].query == q) ++e;
			const int64_t w = ql[q + 1] - ql[q] - 1 + 2 * GF_PAD;
			const int cls = w <= widths[0] ? 0 : w <= widths[1] ? 1 : w <= widths[2] ? 2 : 3;
			units[cls].push_back(make_int2((int)b, (int)e));
			b = e;
		}
		if (n_hits > 0x7fffffff) return fail(DMND_E_ARG, "dmnd_gapped_filter: more than 2^31 hits in one call");
		const size_t total = units[0].size() + units[1].size() + units[2].size() + units[3].size();
		if (int rc = c->gf_units.ensure(total * sizeof(int2))) return rc;
		size_t off = 0;
		for (int k = 0; k < 4; ++k) {
			if (!units[k].empty()) HIP_TRY(hipMemcpyAsync(c->gf_units.as<int2>() + off, units[k].data(), units[k].size() * sizeof(int2), hipMemcpyHostToDevice, st));
			off += units[k].size();
		}
	}
	HIP_TRY(hipEventRecord(c->ev0, st));
	if (use_profile) {
		size_t off = 0;
		for (int k = 0; k < 4; ++k) {
			HIP_TRY(launch_gapped_filter_units(a, c->gf_units.as<int2>() + off, (int)units[k].size(), widths[k], st));
			off += units[k].size();
		}
	}
	else HIP_TRY(launch_gapped_filter(a, st));
	HIP_TRY(hipEventRecord(c->ev1, st));
	if (flags) HIP_TRY(hipMemcpyAsync(flags, c->gf_flags.p, (size_t)n_hits, hipMemcpyDeviceToHost, st));
	if (scores) HIP_TRY(hipMemcpyAsync(scores, c->gf_scores.p, (size_t)n_hits * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
	HIP_TRY(sync_stream(st));
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
	c->gf_ms = ms;
	return DMND_OK;
}

extern "C" double dmnd_gapped_filter_ms(const dmnd_ctx* c) { return c ? c->gf_ms : 0.0; }
