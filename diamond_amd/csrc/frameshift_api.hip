// frameshift_api.hip -- host side of the three-frame banded sweep (frameshift alignment, blastx -F): dmnd_frameshift_swipe.
//
// Replaces the dispatch point banded_3frame_swipe(query, strand, targets, stat, score_only, parallel)
// (/root/reference/src/dp/dp.h:296, dp/swipe/banded_3frame_swipe.cpp:597-647) for many (query, strand) calls at once. What the
// reference does around its vector kernel is restated here because it shapes the results:
//   * score-only: the targets of one call are ordered by DpTarget::operator< (dp/dp.h:105-111, a stable sort) and swept 16 at a
//     time -- the channels of one int16 vector (AVX2) -- on ONE band geometry: the widest band of the 16, rows starting at the
//     lowest band end (banded_3frame_swipe.cpp:432-444). A narrower target therefore gets a band widened downwards, and the query
//     range its score-only Hsp reports (:398-414) comes from the vector's geometry. dmnd_frameshift_swipe forms the same groups of
//     `channels` and gives every item the geometry of its group;
//   * traceback: the reference's 32-bit "vector" holds one target: every item on its own band.
// The device sweeps 64 items per wavefront (frameshift_kernels.hip); items go to wavefronts in an order by band width so that a
// wavefront's state columns -- laid out for its widest band -- are not mostly padding.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>
#include "ctx.h"
#include "frameshift_core.h"
#include "frameshift_kernels.h"

using namespace dmnd;

namespace {

int c_div(int a, int b) { return a / b; }                     // (C division: DpTarget::cols may be negative)

struct Pending { int64_t item; int i0, i1, pos0; };

int run_launch(dmnd_ctx* c, bool traceback, int frame_shift, const dmnd_fs_target* items, std::vector<Pending>& work, std::vector<F3Result>& results,
	std::vector<uint8_t>* transcripts, std::vector<int64_t>* transcript_at)
{
	const int64_t n = (int64_t)work.size();
	if (n == 0) return DMND_OK;
	// wavefronts of similar band width (and, inside a width, similar column counts: the reference's own batching order)
	std::stable_sort(work.begin(), work.end(), [&](const Pending& x, const Pending& y) {
		const int bx = x.i1 - x.i0, by = y.i1 - y.i0;
		return bx > by || (bx == by && items[x.item].target_len > items[y.item].target_len);
	});
	const size_t trace_budget = c->trace_arena_max / 2;
	results.assign((size_t)n, F3Result());
	if (transcripts) { transcripts->clear(); transcript_at->assign((size_t)n, -1); }
	for (int64_t c0 = 0; c0 < n;) {
		// a chunk = whole wavefronts whose kept columns fit the trace budget
		std::vector<F3DevItem> dev;
		std::vector<int64_t> wave_off;
		std::vector<int32_t> wave_rows;
		int64_t state = 0, trace = 0, tr_bytes = 0, c1 = c0;
		while (c1 < n) {
			const int64_t w1 = std::min(n, c1 + 64);
			int rows = 0;
			int64_t t_add = 0;
			for (int64_t k = c1; k < w1; ++k) {
				const Pending& p = work[(size_t)k];
				const dmnd_fs_target& it = items[p.item];
				const int B = (p.i1 - p.i0 + 1) * 3;
				rows = std::max(rows, B);
				if (traceback) {
					F3Item tmp; tmp.len[0] = it.frame_len[0]; tmp.tlen = it.target_len; tmp.i0 = p.i0; tmp.i1 = p.i1; tmp.pos0 = p.pos0;
					t_add += (int64_t)(f3_trace_cols(tmp) + 2) * (B + 1);
				}
			}
			if (c1 > c0 && (size_t)(trace + t_add) * sizeof(int32_t) > trace_budget) break;
			// ... and (score-only passes keep no trace, so the budget above never ends a chunk) at 1 GB of interleaved DP state or 4 M
			// items: the state and item arrays of a large read block are not sized for all of it at once
			if (c1 > c0 && ((size_t)state * sizeof(int32_t) > ((size_t)1 << 30) || c1 - c0 >= ((int64_t)4 << 20))) break;
			for (int64_t k = c1; k < w1; ++k) {
				const Pending& p = work[(size_t)k];
				const dmnd_fs_target& it = items[p.item];
				F3DevItem d;
				std::memset(&d, 0, sizeof d);
				for (int f = 0; f < 3; ++f) { d.frame_off[f] = it.frame_off[f]; d.len[f] = it.frame_len[f]; }
				d.target_off = it.target_off; d.tlen = it.target_len; d.i0 = p.i0; d.i1 = p.i1; d.pos0 = p.pos0;
				d.strand = it.strand; d.dna_len = it.dna_len; d.out = (int32_t)k;
				if (traceback) {
					const int B = (p.i1 - p.i0 + 1) * 3;
					F3Item tmp; tmp.len[0] = it.frame_len[0]; tmp.tlen = it.target_len; tmp.i0 = p.i0; tmp.i1 = p.i1; tmp.pos0 = p.pos0;
					d.trace_off = trace; trace += (int64_t)(f3_trace_cols(tmp) + 2) * (B + 1);
					d.transcript_off = tr_bytes; d.transcript_cap = 2 * it.target_len + it.frame_len[0] + 64; tr_bytes += d.transcript_cap;
				}
				dev.push_back(d);
			}
			wave_off.push_back(state); wave_rows.push_back(rows);
			state += (int64_t)(2 * rows + 5) * 64;                // score column rows + 2, gap column rows + 3
			c1 = w1;
		}
		const int64_t m = c1 - c0;
		HIP_TRY(hipSetDevice(c->device));
		// work buffers of the context (the banded sweeps' buffers serve: the two paths never run at the same time on one context)
		if (int rc = c->items.ensure(dev.size() * sizeof(F3DevItem))) return rc;
		if (int rc = c->order.ensure(wave_off.size() * sizeof(int64_t))) return rc;
		if (int rc = c->p_of_slot.ensure(wave_rows.size() * sizeof(int32_t))) return rc;
		if (int rc = c->ends.ensure((size_t)state * sizeof(int32_t) + 256)) return rc;
		if (int rc = c->hsps.ensure(dev.size() * sizeof(F3Result))) return rc;
		if (traceback) {
			if (int rc = c->trace.ensure((size_t)trace * sizeof(int32_t) + 256)) return rc;
			if (int rc = c->transcript.ensure((size_t)tr_bytes + 256)) return rc;
			HIP_TRY(hipMemsetAsync(c->trace.p, 0, (size_t)trace * sizeof(int32_t), c->stream));
		}
		HIP_TRY(hipMemsetAsync(c->ends.p, 0, (size_t)state * sizeof(int32_t), c->stream));
		HIP_TRY(hipMemcpyAsync(c->items.p, dev.data(), dev.size() * sizeof(F3DevItem), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(c->order.p, wave_off.data(), wave_off.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(c->p_of_slot.p, wave_rows.data(), wave_rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
		F3Args a;
		a.qblock = c->block[DMND_QUERY].as<int8_t>(); a.tblock = c->block[DMND_TARGET].as<int8_t>(); a.matrix = c->matrix.as<int8_t>();
		a.items = c->items.as<F3DevItem>(); a.wave_off = c->order.as<int64_t>(); a.wave_rows = c->p_of_slot.as<int32_t>();
		a.state = c->ends.as<int32_t>(); a.trace = traceback ? c->trace.as<int32_t>() : nullptr; a.transcript = traceback ? c->transcript.as<uint8_t>() : nullptr;
		a.results = c->hsps.as<F3Result>(); a.n = m;
		a.gap_open = c->params.gap_open; a.gap_extend = c->params.gap_extend; a.frame_shift = frame_shift;
		HIP_TRY(hipEventRecord(c->ev0, c->stream));
		HIP_TRY(launch_frameshift_sweep(traceback, a, c->stream));
		HIP_TRY(hipEventRecord(c->ev1, c->stream));
		HIP_TRY(copy_now(c->stream, results.data() + c0, c->hsps.p, (size_t)m * sizeof(F3Result), hipMemcpyDeviceToHost));
		float ms = 0.f;
		HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
		c->swipe_ms += ms;
		if (traceback && transcripts) {
			std::vector<uint8_t> raw((size_t)tr_bytes);
			if (tr_bytes) HIP_TRY(copy_now(c->stream, raw.data(), c->transcript.p, (size_t)tr_bytes, hipMemcpyDeviceToHost));
			for (int64_t k = 0; k < m; ++k) {
				const F3Result& r = results[(size_t)(c0 + k)];
				if (r.status != 0) return fail(r.status, r.status == DMND_E_TRACEBACK ? "Traceback error." : "transcript slot too small");
				if (r.score <= 0) continue;
				const F3DevItem& d = dev[(size_t)k];
				(*transcript_at)[(size_t)(c0 + k)] = (int64_t)transcripts->size();
				const uint8_t* src = raw.data() + d.transcript_off + d.transcript_cap - r.transcript_len;      // the walk fills its slot from the back
				transcripts->insert(transcripts->end(), src, src + r.transcript_len);
				transcripts->push_back(0);
			}
		}
		c0 = c1;
	}
	return DMND_OK;
}

}  // namespace

extern "C" int dmnd_frameshift_swipe(dmnd_ctx* c, const dmnd_fs_target* items, int64_t n, int score_only, int frame_shift, int channels,
	dmnd_fs_hsp* out, uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used)
{
	if (!c) return fail(DMND_E_ARG, "ctx is NULL");
	if (transcript_used) *transcript_used = 0;
	if (n == 0) return DMND_OK;
	if (!items || !out || n < 0 || n > 0x7fffffff || frame_shift <= 0 || channels < 1) return fail(DMND_E_ARG, "dmnd_frameshift_swipe: bad argument");
	const int64_t ql = c->block_len[DMND_QUERY], tl = c->block_len[DMND_TARGET];
	if (ql == 0 || tl == 0) return fail(DMND_E_ARG, "dmnd_frameshift_swipe: sequence blocks not uploaded");
	for (int64_t i = 0; i < n; ++i) {
		const dmnd_fs_target& it = items[i];
		bool ok = it.target_len > 0 && it.target_off >= 0 && it.target_off + it.target_len <= tl && it.d_end > it.d_begin && it.frame_len[0] > 0
			&& it.frame_len[1] <= it.frame_len[0] && it.frame_len[2] <= it.frame_len[1] && it.frame_len[2] >= it.frame_len[0] - 1 && (it.strand == 0 || it.strand == 1);
		for (int f = 0; f < 3 && ok; ++f) ok = it.frame_len[f] >= 0 && it.frame_off[f] >= 0 && it.frame_off[f] + it.frame_len[f] <= ql;
		if (!ok) return fail(DMND_E_ARG, "dmnd_frameshift_swipe: item " + std::to_string(i) + " out of range");
		if ((int64_t)(it.d_end - it.d_begin) > DMND_MAX_BAND) return fail(DMND_E_BAND, "Band size exceeds the supported maximum");
	}
	c->swipe_ms = 0;
	std::vector<Pending> work;
	work.reserve((size_t)n);
	if (!score_only) {
		for (int64_t i = 0; i < n; ++i) {
			F3Item g; f3_own_geometry(g, items[i].d_begin, items[i].d_end);
			work.push_back(Pending{ i, g.i0, g.i1, g.pos0 });
		}
	}
	else {
		// the calls (group = one query strand) one after the other: order, then `channels` at a time on one geometry
		const int band_bin = 24, col_bin = 400;                   // config.band_bin, config.col_bin (basic/config.cpp:562-563)
		std::vector<int64_t> idx;
		for (int64_t g0 = 0; g0 < n;) {
			int64_t g1 = g0;
			while (g1 < n && items[g1].group == items[g0].group) ++g1;
			idx.resize((size_t)(g1 - g0));
			std::iota(idx.begin(), idx.end(), g0);
			std::stable_sort(idx.begin(), idx.end(), [&](int64_t x, int64_t y) {
				const dmnd_fs_target &a = items[x], &b = items[y];
				const int ba = c_div(a.d_end - a.d_begin, band_bin), bb = c_div(b.d_end - b.d_begin, band_bin), ta = c_div(a.cols, col_bin), tb = c_div(b.cols, col_bin);
				return ba < bb || (ba == bb && (ta < tb || (ta == tb && std::max(a.d_end - 1, 0) < std::max(b.d_end - 1, 0))));
			});
			for (size_t b0 = 0; b0 < idx.size(); b0 += (size_t)channels) {
				const size_t b1 = std::min(idx.size(), b0 + (size_t)channels);
				int band = 0, i1 = 0x7fffffff;
				for (size_t k = b0; k < b1; ++k) {
					band = std::max(band, items[idx[k]].d_end - items[idx[k]].d_begin);
					i1 = std::min(i1, std::max(items[idx[k]].d_end - 1, 0));
				}
				for (size_t k = b0; k < b1; ++k) work.push_back(Pending{ idx[k], i1 + 1 - band, i1, i1 - (items[idx[k]].d_end - 1) });
			}
			g0 = g1;
		}
	}
	std::vector<F3Result> res;
	std::vector<uint8_t> tr;
	std::vector<int64_t> tr_at;
	const bool want_tr = !score_only && transcript != nullptr;
	if (int rc = run_launch(c, !score_only, frame_shift, items, work, res, want_tr ? &tr : nullptr, want_tr ? &tr_at : nullptr)) return rc;
	std::vector<Pending> again;
	for (size_t k = 0; k < work.size(); ++k) {
		const F3Result& r = res[k];
		if (!score_only && r.status != 0) return fail(r.status, r.status == DMND_E_TRACEBACK ? "Traceback error." : "transcript slot too small");
		dmnd_fs_hsp h;
		std::memset(&h, 0, sizeof h);
		h.score = r.score; h.frame = r.frame; h.q_begin = r.q_begin; h.q_end = r.q_end; h.s_begin = r.s_begin; h.s_end = r.s_end;
		h.read_begin = r.read_begin; h.read_end = r.read_end; h.length = r.length; h.identities = r.identities; h.mismatches = r.mismatches;
		h.positives = r.positives; h.gap_openings = r.gap_openings; h.gaps = r.gaps; h.transcript_len = r.transcript_len; h.max_col = r.max_col;
		h.transcript_off = -1;
		out[work[k].item] = h;
		// the reference's int16 vectors saturate at 65535: such a target is swept again alone, on its own band (banded_3frame_swipe.cpp:610-640)
		if (score_only && r.score >= 65535) { F3Item g; f3_own_geometry(g, items[work[k].item].d_begin, items[work[k].item].d_end); again.push_back(Pending{ work[k].item, g.i0, g.i1, g.pos0 }); }
	}
	if (!again.empty()) {
		std::vector<F3Result> res2;
		if (int rc = run_launch(c, false, frame_shift, items, again, res2, nullptr, nullptr)) return rc;
		for (size_t k = 0; k < again.size(); ++k) {
			dmnd_fs_hsp& h = out[again[k].item];
			h.score = res2[k].score; h.max_col = res2[k].max_col; h.q_begin = res2[k].q_begin; h.q_end = res2[k].q_end; h.read_begin = res2[k].read_begin; h.read_end = res2[k].read_end;
		}
	}
	if (want_tr) {
		// transcripts in the caller's item order
		std::vector<int64_t> at_item((size_t)n, -1);
		for (size_t k = 0; k < work.size(); ++k) at_item[(size_t)work[k].item] = tr_at[k];
		int64_t used = 0;
		for (int64_t i = 0; i < n; ++i) {
			if (at_item[(size_t)i] < 0) continue;
			const int64_t len = out[i].transcript_len + 1;
			if (used + len > transcript_cap) return fail(DMND_E_CAP, "dmnd_frameshift_swipe: transcript arena too small");
			std::memcpy(transcript + used, tr.data() + at_item[(size_t)i], (size_t)len);
			out[i].transcript_off = used;
			used += len;
		}
		if (transcript_used) *transcript_used = used;
	}
	return DMND_OK;
}
