// join_blocks.hip -- the join of the match records of several reference blocks (/root/reference/src/output/join_blocks.cpp:129-256):
// on the host for the records a caller holds there (dmnd_join_blocks, _top, _range: -k, --top, --range-culling), and the gather of
// several contexts' device-resident records in front of the device join (join_device.hip). Split out of extend_host.hip in round 6.
#include <algorithm>
#include <string>
#include <vector>
#include "ctx.h"
#include "read_coverage.h"
#include "extend_kernels.h"
#include "match_order.h"

using namespace dmnd;

// join_query with --top: the heap merge runs on JoinRecord::cmp_score (score descending, target ordinal ascending) and GlobalCulling
// keeps a target while (1 - bit score / best bit score) * 100 <= toppercent (output/target_culling.h:62-63)
namespace {

// The records of a join: consecutive records of one (query, target) pair are the HSPs of one match (dmnd_set_max_hsps) and move
// together, ranked by the first one. Returns the matches as (first record, count), ordered by query and `less` of the first records.
// Every block's list arrives in (query, rank) order and query ids are dense, so the order is made by a counting sort of the matches
// by query (stable: block order) and a small sort per query -- not one comparison sort over everything (C5: 116 k records of 8
// blocks, 20 ms of every 62 ms step went into std::stable_sort here).
template<typename Less>
std::vector<std::pair<int64_t, int64_t>> join_groups(const dmnd_match* r, int64_t n, Less less)
{
	std::vector<std::pair<int64_t, int64_t>> g;
	g.reserve((size_t)n);
	uint32_t max_query = 0;
	for (int64_t i = 0; i < n;) {
		int64_t j = i + 1;
		while (j < n && r[j].query == r[i].query && r[j].target == r[i].target) ++j;
		g.emplace_back(i, j - i);
		max_query = std::max(max_query, r[i].query);
		i = j;
	}
	auto by_rank = [&](const std::pair<int64_t, int64_t>& a, const std::pair<int64_t, int64_t>& b) { return less(r[a.first], r[b.first]); };
	if (g.empty()) return g;
	if ((uint64_t)max_query > 4 * (uint64_t)g.size() + 1024) {          // sparse query ids: one sort
		std::stable_sort(g.begin(), g.end(), [&](const std::pair<int64_t, int64_t>& a, const std::pair<int64_t, int64_t>& b) {
			return r[a.first].query < r[b.first].query || (r[a.first].query == r[b.first].query && less(r[a.first], r[b.first]));
		});
		return g;
	}
	std::vector<int64_t> start((size_t)max_query + 2, 0);
	for (const auto& x : g) ++start[(size_t)r[x.first].query + 1];
	for (size_t q = 1; q < start.size(); ++q) start[q] += start[q - 1];
	std::vector<std::pair<int64_t, int64_t>> out(g.size());
	{
		std::vector<int64_t> at(start.begin(), start.end() - 1);
		for (const auto& x : g) out[(size_t)at[r[x.first].query]++] = x;
	}
	for (size_t q = 0; q + 1 < start.size(); ++q)
		if (start[q + 1] - start[q] > 1) std::stable_sort(out.begin() + (ptrdiff_t)start[q], out.begin() + (ptrdiff_t)start[q + 1], by_rank);
	return out;
}

// the kept matches in their new order: gathered into a scratch array (independent reads, sequential writes), copied back. (An
// in-place permutation along the chains of the mapping moves every record once instead of twice and was slower: its reads depend
// on each other.)
void write_groups(dmnd_match* r, const std::vector<std::pair<int64_t, int64_t>>& keep, int64_t* n_out)
{
	int64_t total = 0;
	for (const auto& g : keep) total += g.second;
	std::vector<dmnd_match> out((size_t)total);
	int64_t w = 0;
	for (const auto& g : keep) for (int64_t k = 0; k < g.second; ++k) out[(size_t)w++] = r[g.first + k];
	std::copy(out.begin(), out.end(), r);
	*n_out = total;
}

}

extern "C" int dmnd_join_blocks_top(dmnd_match* r, int64_t n, double top_percent, int64_t* n_out)
{
	if (!r || n < 0 || top_percent < 0.0 || !n_out) return fail(DMND_E_ARG, "dmnd_join_blocks_top: bad argument");
	const auto groups = join_groups(r, n, match_less_score);
	std::vector<std::pair<int64_t, int64_t>> keep;
	double top_score = 0.0;
	bool finished = false;
	for (size_t i = 0; i < groups.size(); ++i) {
		const dmnd_match& m = r[groups[i].first];
		if (i == 0 || m.query != r[groups[i - 1].first].query) { top_score = m.bit_score; finished = false; }
		if (finished) continue;
		if ((1.0 - m.bit_score / top_score) * 100.0 <= top_percent) keep.push_back(groups[i]);
		else finished = true;
	}
	write_groups(r, keep, n_out);
	return DMND_OK;
}

// join_query over the records of several reference blocks (output/join_blocks.cpp:180-256): every block's list of a query is
// already in match_less order, so the reference's heap merge by JoinRecord::cmp_evalue (join_blocks.cpp:129-142) is the sort of
// the union by (e-value, score descending, target ordinal); GlobalCulling keeps the first max_target_seqs (target_culling.h:70-88).
extern "C" int dmnd_join_blocks(dmnd_match* r, int64_t n, int max_target_seqs, int64_t* n_out)
{
	if (!r || n < 0 || max_target_seqs < 1 || !n_out) return fail(DMND_E_ARG, "dmnd_join_blocks: bad argument");
	const auto groups = join_groups(r, n, match_less);
	std::vector<std::pair<int64_t, int64_t>> keep;
	int64_t run = 0;
	for (size_t i = 0; i < groups.size(); ++i) {
		run = (i > 0 && r[groups[i].first].query == r[groups[i - 1].first].query) ? run + 1 : 0;
		if (run < max_target_seqs) keep.push_back(groups[i]);
	}
	write_groups(r, keep, n_out);
	return DMND_OK;
}

// join_query with --range-culling (blastx -F n --range-culling / --long-reads over a database of several reference blocks): the
// reference builds its join culler with TargetCulling::get (output/target_culling.cpp:22-28), which is RangeCulling then -- a
// target of the merged order (JoinRecord::cmp_evalue, or cmp_score with --top) is dropped (NEXT, never FINISHED) when
// range_cover per cent of its HSPs' read intervals are already covered: by max_target_seqs kept alignments
// (IntervalPartition::covered), or with --top by one kept alignment of at least score / (1 - top / 100)
// (covered(..., MaxScore), output/target_culling.h:123-150). Kept targets add their intervals (IntermediateRecord::
// absolute_query_range = dmnd_match::read_begin / read_end). Rounds 3-4 applied GlobalCulling here whatever the mode: every
// target outside the top per cent of the read's single best score was lost, also when it covers another part of the read.
extern "C" int dmnd_join_blocks_range(dmnd_match* r, int64_t n, int max_target_seqs, double top_percent, double range_cover, int64_t* n_out)
{
	if (!r || n < 0 || max_target_seqs < 1 || top_percent > 100.0 || !n_out) return fail(DMND_E_ARG, "dmnd_join_blocks_range: bad argument");
	if (top_percent >= 100.0) top_percent = -1.0;       // RangeCulling: toppercent == 100.0 means "no --top" (the count-based coverage test, cmp_evalue order)
	const auto groups = top_percent >= 0.0 ? join_groups(r, n, match_less_score) : join_groups(r, n, match_less);
	std::vector<std::pair<int64_t, int64_t>> keep;
	Coverage cover(max_target_seqs);
	for (size_t i = 0; i < groups.size(); ++i) {
		if (i == 0 || r[groups[i].first].query != r[groups[i - 1].first].query) cover = Coverage(max_target_seqs);
		int cv = 0, len = 0;
		for (int64_t k = 0; k < groups[i].second; ++k) {
			const dmnd_match& m = r[groups[i].first + k];
			if (m.read_end <= m.read_begin) return fail(DMND_E_ARG, "dmnd_join_blocks_range: a record without its interval of the read (read_begin / read_end are set by frameshift alignment)");
			const Interval iv{ m.read_begin, m.read_end };
			cv += top_percent < 0.0 ? cover.covered(iv) : cover.covered_max(iv, (int)((double)m.hsp.score / (1.0 - top_percent / 100.0)));
			len += iv.length();
		}
		if (!((double)cv / len * 100.0 < range_cover)) continue;
		for (int64_t k = 0; k < groups[i].second; ++k) {
			const dmnd_match& m = r[groups[i].first + k];
			cover.insert(Interval{ m.read_begin, m.read_end }, m.hsp.score);
		}
		keep.push_back(groups[i]);
	}
	write_groups(r, keep, n_out);
	return DMND_OK;
}

extern "C" int dmnd_join_blocks_device(dmnd_ctx* c, const dmnd_match* records_dev, int64_t n, int max_target_seqs, double top_percent, uint32_t max_query,
	dmnd_match* out_dev, int64_t* n_out);

extern "C" int dmnd_join_contexts_device(dmnd_ctx* join_ctx, dmnd_ctx* const* ctx, const uint32_t* target_offset, int n_ctx, int max_target_seqs, double top_percent,
	uint32_t max_query, dmnd_match* out, int64_t cap, int64_t* n_out)
{
	if (!join_ctx || !ctx || !target_offset || n_ctx < 1 || !n_out || cap < 0 || (cap > 0 && !out)) return fail(DMND_E_ARG, "dmnd_join_contexts_device: bad argument");
	*n_out = 0;
	int64_t total = 0;
	for (int k = 0; k < n_ctx; ++k) {
		if (!ctx[k] || ctx[k]->device != join_ctx->device) return fail(DMND_E_ARG, "dmnd_join_contexts_device: the contexts must be on the join context's device");
		if (ctx[k]->ext_records_n < 0) return fail(DMND_E_ARG, "dmnd_join_contexts_device: context " + std::to_string(k) + " holds no complete device copy of its last dmnd_extend's records (dmnd_extend_records_device)");
		total += ctx[k]->ext_records_n;
	}
	if (total == 0) return DMND_OK;
	if (total > 0xffffffffLL) return fail(DMND_E_CAP, "dmnd_join_contexts_device: more than 2^32 records");
	HIP_TRY(hipSetDevice(join_ctx->device));
	if (int rc = join_ctx->join_in.ensure((size_t)total * sizeof(dmnd_match))) return rc;
	if (int rc = join_ctx->join_out.ensure((size_t)total * sizeof(dmnd_match))) return rc;
	int64_t at = 0;
	for (int k = 0; k < n_ctx; ++k) {      // (dmnd_extend left every source's stream idle: the records are final)
		HIP_TRY(launch_ext_gather(join_ctx->join_in.as<dmnd_match>() + at, ctx[k]->ext_records_dev, (uint32_t)ctx[k]->ext_records_n, target_offset[k], join_ctx->stream));
		at += ctx[k]->ext_records_n;
	}
	int64_t kept = 0;
	if (int rc = dmnd_join_blocks_device(join_ctx, join_ctx->join_in.as<dmnd_match>(), total, max_target_seqs, top_percent, max_query, join_ctx->join_out.as<dmnd_match>(), &kept)) return rc;
	*n_out = kept;
	if (kept > cap) return fail(DMND_E_CAP, "dmnd_join_contexts_device: record buffer too small");
	if (kept > 0) if (int rc = download_bytes(join_ctx, out, join_ctx->join_out.p, (size_t)kept * sizeof(dmnd_match))) return rc;
	return DMND_OK;
}

