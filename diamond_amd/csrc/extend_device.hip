// extend_device.hip -- the host side of the device half of dmnd_extend (round 6; split out of extend_host.hip): plan_on_device
// launches the planner (plan_kernels.hip: /root/reference/src/align/load_hits.h:44-127, ungapped.cpp:62-126, chaining/greedy_align.cpp,
// gapped_score.cpp:41-180) and reads its counters; extend_on_device drives the ranking iterations of the planned queries in HBM
// (extend_kernels.hip: /root/reference/src/align/extend.cpp:289-336, gapped_score.cpp:182-268, culling.cpp:97-113, gapped_final.cpp:66-160)
// -- per iteration one counter read-back, the sweeps of its band classes (api.hip dmnd_sweep_classes), append_hits on the device --
// then round 2, the records, the host's own e-value and bit score written back into the records where they stay for the join.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ctx.h"
#include "host_pool.h"
#include "bias_kernels.h"
#include "plan_kernels.h"
#include "extend_kernels.h"
#include "extend_device.h"
#include "match_order.h"

using namespace dmnd;

// Runs the planner over the call's hits (in c->xd_hits, with their x-drop extensions in c->xd_out and -- gf_on -- their gapped
// filter flags in c->gf_flags), waits for it and copies its lists to the host. planned = false: the hits are not in
// (query, location, seed offset) order, the host has to plan.
int dmnd::plan_on_device(dmnd_ctx* c, const DeviceCfg& h, int64_t n_hits, bool gf_on, DevPlan& plan, bool& planned)
{
	planned = false;
	const size_t n = (size_t)n_hits;
	auto align = [](size_t x) { return (x + 63) & ~(size_t)63; };
	const size_t o_tgt = 0, o_heads = align(o_tgt + n * sizeof(uint32_t)), o_scan = align(o_heads + n * sizeof(uint64_t)),
		o_groups = align(o_scan + n * sizeof(uint64_t)), o_queries = align(o_groups + (n + 1) * sizeof(PlanGroup)),
		o_segs = align(o_queries + (n + 1) * sizeof(PlanQuery)), o_slots = align(o_segs + n * 4 * sizeof(int32_t)),
		o_count = align(o_slots + n * sizeof(PlanBand)), o_off = align(o_count + (n + 1) * sizeof(uint32_t)),
		o_bands = align(o_off + (n + 1) * sizeof(uint32_t)), o_counters = align(o_bands + n * sizeof(PlanBand)),
		o_chain = align(o_counters + sizeof(PlanCounters)), bytes = o_chain + (n + 2) * sizeof(uint32_t);      // (both chaining lists, and a small group listed again)
	TraceLaps tr("dmnd_extend (planner)");
	if (int rc = c->plan_dev.ensure(bytes)) return rc;
	tr.lap("work arrays");
	char* d = c->plan_dev.as<char>();
	PlanArgs a;
	a.qblock = c->block[DMND_QUERY].as<int8_t>(); a.tblock = c->block[DMND_TARGET].as<int8_t>();
	a.qlimits = c->d_limits[DMND_QUERY].as<int64_t>(); a.tlimits = c->d_limits[DMND_TARGET].as<int64_t>();
	a.n_targets = (int64_t)c->limits[DMND_TARGET].size() - 1;
	a.matrix = c->matrix.as<int8_t>();
	a.hits = c->xd_hits.as<dmnd_seed_hit>(); a.n_hits = n_hits;
	a.gf_flags = gf_on ? c->gf_flags.as<uint8_t>() : nullptr;
	a.xd = c->xd_out.as<XdropSeg>();
	a.gap_open = h.gap_open; a.gap_extend = h.gap_extend; a.band_fast = h.band_mode_fast;
	a.small_segs = n_hits >= ((int64_t)1 << 18) ? 4 : 0;
	a.tgt = reinterpret_cast<uint32_t*>(d + o_tgt); a.heads = reinterpret_cast<uint64_t*>(d + o_heads); a.head_scan = reinterpret_cast<uint64_t*>(d + o_scan);
	a.groups = reinterpret_cast<PlanGroup*>(d + o_groups); a.queries = reinterpret_cast<PlanQuery*>(d + o_queries);
	a.segs = reinterpret_cast<int32_t*>(d + o_segs); a.band_slots = reinterpret_cast<PlanBand*>(d + o_slots);
	a.band_count = reinterpret_cast<uint32_t*>(d + o_count); a.band_off = reinterpret_cast<uint32_t*>(d + o_off);
	a.bands = reinterpret_cast<PlanBand*>(d + o_bands); a.counters = reinterpret_cast<PlanCounters*>(d + o_counters);
	a.chain_list = reinterpret_cast<uint32_t*>(d + o_chain); a.chain_cap = (uint32_t)(n + 2);
	a.scan_tmp = &c->plan_tmp; a.scan_tmp_bytes = &c->plan_tmp_bytes;
	HIP_TRY(launch_plan(a, c->stream));
	tr.lap("launched");
	if (int rc = c->plan_host.ensure(sizeof(PlanCounters))) return rc;
	tr.lap("host buffer");
	HIP_TRY(copy_now(c->stream, c->plan_host.p, a.counters, sizeof(PlanCounters), hipMemcpyDeviceToHost));
	tr.lap("counters back");
	const PlanCounters cn = *c->plan_host.as<PlanCounters>();
	if (cn.unsorted || cn.n_groups == 0) return DMND_OK;
	plan.n_groups = cn.n_groups; plan.n_queries = cn.n_queries; plan.n_bands = cn.n_bands; plan.n_on_host = cn.n_on_host;
	plan.dev = a;
	planned = true;
	return DMND_OK;
}

// The planner's lists on the host (page-locked copies, valid until the context's next dmnd_extend): only the host path reads them
int dmnd::plan_fetch_lists(dmnd_ctx* c, DevPlan& plan)
{
	if (plan.groups) return DMND_OK;
	auto align = [](size_t x) { return (x + 63) & ~(size_t)63; };
	const size_t h_groups = align(sizeof(PlanCounters)), h_queries = align(h_groups + (size_t)plan.n_groups * sizeof(PlanGroup)),
		h_bands = align(h_queries + ((size_t)plan.n_queries + 1) * sizeof(PlanQuery)), h_bytes = h_bands + (size_t)plan.n_bands * sizeof(PlanBand);
	if (int rc = c->plan_host.ensure(h_bytes)) return rc;
	char* hp = c->plan_host.as<char>();
	HIP_TRY(hipMemcpyAsync(hp + h_groups, plan.dev.groups, (size_t)plan.n_groups * sizeof(PlanGroup), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipMemcpyAsync(hp + h_queries, plan.dev.queries, ((size_t)plan.n_queries + 1) * sizeof(PlanQuery), hipMemcpyDeviceToHost, c->stream));
	if (plan.n_bands) HIP_TRY(hipMemcpyAsync(hp + h_bands, plan.dev.bands, (size_t)plan.n_bands * sizeof(PlanBand), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(sync_stream(c->stream));
	plan.groups = reinterpret_cast<const PlanGroup*>(hp + h_groups);
	plan.queries = reinterpret_cast<const PlanQuery*>(hp + h_queries);
	plan.bands = reinterpret_cast<const PlanBand*>(hp + h_bands);
	return DMND_OK;
}

static bool keep_traces_dev() { static const bool v = [] { const char* e = std::getenv("DMND_EXTEND_KEEP_TRACE"); return !e || e[0] != '0'; }(); return v; }

// The extension of the queries whose targets fit one ranking chunk, in HBM from the planner's bands to the match records
// (extend_kernels.h). records: those queries' matches in output order (query ascending; e-value, score, target inside a query) with
// the HOST's e-value and bit score; qstate[k] (k = index into plan.queries): EXT_Q_DEVICE = done here, anything else = the host path
// has to extend the query. done = false: nothing was done here (no eligible query, or the kept traces would not fit the context's
// trace budget), every query goes to the host path.
int dmnd::extend_on_device(dmnd_ctx* c, const DeviceCfg& h, const DevPlan& plan, int threads, std::vector<dmnd_match>& records, std::vector<uint8_t>& qstate, bool& done)
{
	done = false;
	TraceLaps tr("dmnd_extend (device half)");
	const int64_t chunk = h.ranking_chunk;
	if (chunk > EXT_MAX_CHUNK || plan.n_bands == 0) return DMND_OK;
	if (((size_t)h.max_target_seqs + 2 * (size_t)chunk) * 24 > ((size_t)60 << 10)) return DMND_OK;      // (the LDS lists of ext_append_kernel)
	const size_t nG = plan.n_groups, nQ = plan.n_queries, nB = plan.n_bands, nR = std::min(nG, nQ * (size_t)std::max(h.max_target_seqs, 1));
	const size_t nI = nB + nR;                            // items: every band once + a copy of every survivor (round 2 without kept traces)
	size_t at = 0;
	auto take = [&](size_t bytes) { const size_t o = at; at = (at + bytes + 63) & ~(size_t)63; return o; };
	const size_t o_qstate = take(nQ), o_qactive = take(nQ), o_qi0 = take(nQ * 4), o_qi1 = take(nQ * 4), o_qtail = take(nQ * 4), o_qprev = take(nQ * 4), o_qswept = take(nQ * 4),
		o_okeys = take(nG * 8), o_okeys2 = take(nG * 8), o_oidx = take(nG * 4), o_gorder = take(nG * 4), o_aligned = take(nG), o_gfirst = take(nG * 4), o_gcnt = take(nG * 4),
		o_cnt = take((nG + 1) * 4), o_item_off = take((nG + 1) * 4), o_kept = take((nG + 1) * 4), o_kept_pos = take((nG + 1) * 4), o_cand_item = take(nG * 4), o_cand_ev = take(nG * 8),
		o_items = take(nI * sizeof(dmnd_dp_target)), o_off_item = take(nI * 8), o_p = take(nI * 4), o_ends = take(nI * sizeof(SwipeEnd)), o_hsps = take(nI * sizeof(dmnd_hsp)),
		o_keys = take(nI * 4), o_keys_sorted = take(nI * 4), o_idx = take(nI * 4), o_order = take(nI * 4), o_rows = take(nI * 8), o_rows_slot = take((nI + 1) * 8), o_off_slot = take((nI + 1) * 8),
		o_pairs = take((nI + 8 * EXT_CLASSES) * 4), o_r2_order = take(nR * 4), o_r2_p = take(nR * 4), o_r2_off = take(nR * 8), o_r2_tr = take((nR + 1) * 8), o_r2_group = take(nR * 4),
		o_records = take(nR * sizeof(dmnd_match)), o_ctr = take(sizeof(ExtCounters));
	if (int rc = c->ext_dev.ensure(at)) return rc;
	tr.lap("work arrays");
	char* d = c->ext_dev.as<char>();
	ExtArgs a;
	a.groups = plan.dev.groups; a.queries = plan.dev.queries; a.bands = plan.dev.bands;
	a.n_groups = plan.n_groups; a.n_queries = plan.n_queries; a.n_bands = plan.n_bands;
	a.hits = plan.dev.hits; a.qlimits = plan.dev.qlimits; a.tlimits = plan.dev.tlimits;
	a.use_cbs = h.use_cbs ? 1 : 0; a.row_min_items = (uint32_t)std::min<int64_t>(sweep_rows_min_items(), 0xffffffffll); a.chunk_size = (uint32_t)chunk; a.k = h.max_target_seqs; a.max_swipe_dp = h.max_swipe_dp;
	const Evaluer& E = c->evaluer;
	a.ev = ExtEvalue{ E.lambda, E.K, E.ln_k, E.db_letters, E.a, E.b, E.alpha, E.beta, E.sigma, E.tau, E.v_thr, E.c_thr, h.max_evalue };
	a.qstate = reinterpret_cast<uint8_t*>(d + o_qstate); a.q_active = reinterpret_cast<uint8_t*>(d + o_qactive);
	a.q_i0 = reinterpret_cast<uint32_t*>(d + o_qi0); a.q_i1 = reinterpret_cast<uint32_t*>(d + o_qi1);
	a.q_tail = reinterpret_cast<int32_t*>(d + o_qtail); a.q_prev = reinterpret_cast<int32_t*>(d + o_qprev); a.q_swept = reinterpret_cast<uint32_t*>(d + o_qswept);
	a.okeys = reinterpret_cast<uint64_t*>(d + o_okeys); a.okeys_sorted = reinterpret_cast<uint64_t*>(d + o_okeys2);
	a.oidx = reinterpret_cast<uint32_t*>(d + o_oidx); a.gorder = reinterpret_cast<uint32_t*>(d + o_gorder);
	a.aligned = reinterpret_cast<uint8_t*>(d + o_aligned); a.g_first = reinterpret_cast<uint32_t*>(d + o_gfirst); a.g_cnt = reinterpret_cast<uint32_t*>(d + o_gcnt);
	a.cnt = reinterpret_cast<uint32_t*>(d + o_cnt); a.item_off = reinterpret_cast<uint32_t*>(d + o_item_off);
	a.kept = reinterpret_cast<uint32_t*>(d + o_kept); a.kept_pos = reinterpret_cast<uint32_t*>(d + o_kept_pos);
	a.cand_item = reinterpret_cast<uint32_t*>(d + o_cand_item); a.cand_ev = reinterpret_cast<double*>(d + o_cand_ev);
	a.item_base = 0; a.item_cap = (uint32_t)nI;
	a.items = reinterpret_cast<dmnd_dp_target*>(d + o_items); a.off_item = reinterpret_cast<int64_t*>(d + o_off_item);
	a.p_of_item = reinterpret_cast<int32_t*>(d + o_p); a.ends = reinterpret_cast<SwipeEnd*>(d + o_ends); a.hsps = reinterpret_cast<dmnd_hsp*>(d + o_hsps);
	a.keys = reinterpret_cast<uint32_t*>(d + o_keys); a.keys_sorted = reinterpret_cast<uint32_t*>(d + o_keys_sorted);
	a.idx = reinterpret_cast<uint32_t*>(d + o_idx); a.order = reinterpret_cast<uint32_t*>(d + o_order);
	a.rows = reinterpret_cast<int64_t*>(d + o_rows); a.rows_slot = reinterpret_cast<int64_t*>(d + o_rows_slot); a.off_slot = reinterpret_cast<int64_t*>(d + o_off_slot);
	a.pairs = reinterpret_cast<int32_t*>(d + o_pairs);
	a.r2_order = reinterpret_cast<int32_t*>(d + o_r2_order); a.r2_p = reinterpret_cast<int32_t*>(d + o_r2_p);
	a.r2_off = reinterpret_cast<int64_t*>(d + o_r2_off); a.r2_tr = reinterpret_cast<int64_t*>(d + o_r2_tr); a.r2_group = reinterpret_cast<uint32_t*>(d + o_r2_group);
	a.records = reinterpret_cast<dmnd_match*>(d + o_records);
	a.ctr = reinterpret_cast<ExtCounters*>(d + o_ctr);
	a.scan_tmp = &c->plan_tmp; a.scan_tmp_bytes = &c->plan_tmp_bytes;
	hipStream_t st = c->stream;
	if (int rc = c->ext_host.ensure(sizeof(ExtCounters))) return rc;
	// Round 1 sweeps in traceback mode and keeps the trace rows (round 2 then only walks them) while the iterations' rows fit the
	// context's trace budget: the first iteration's in ext_trace, every later one's in an arena of its own, all addressed from
	// ext_trace's base (64-bit offsets). An iteration that does not fit is swept for scores only, and round 2 sweeps its survivors
	// again with traceback -- what the reference's round 2 does for every survivor.
	const size_t trace_budget = c->trace_arena_max;
	size_t trace_used = 0, n_more = 0;
	auto arena_for = [&](size_t bytes, DevBuf*& arena, int64_t& rel) -> int {
		arena = &c->ext_trace;
		if (trace_used > 0) {
			if (c->ext_trace_more.size() <= n_more) c->ext_trace_more.resize(n_more + 1);
			arena = &c->ext_trace_more[n_more++];
		}
		if (int rc = arena->ensure(bytes + 64)) return rc;
		if (!c->ext_trace.p) { if (int rc = c->ext_trace.ensure(64)) return rc; }
		rel = (int64_t)(arena->as<char>() - c->ext_trace.as<char>());
		trace_used += bytes;
		return DMND_OK;
	};
	HIP_TRY(launch_ext_begin(a, st));
	ExtCounters ctr;
	double ms_sweeps = 0, ms_sweeps2 = 0, ms_walk = 0;
	uint64_t items_total = 0;
	for (int iter = 0;; ++iter) {
		if (iter >= EXT_MAX_ITERATIONS) return fail(DMND_E_CAP, "dmnd_extend: a query's ranking did not end within the supported number of chunks");
		// 1. the chunk's items, launch order, trace offsets, pairs
		HIP_TRY(launch_ext_prepare(a, st));
		HIP_TRY(copy_now(st, c->ext_host.p, a.ctr, sizeof(ExtCounters), hipMemcpyDeviceToHost));
		ctr = *c->ext_host.as<ExtCounters>();
		if (iter == 0) tr.lap("items, launch order, trace offsets");
		if (iter == 0 && ctr.n_items == 0) return DMND_OK;
		// a call whose first ranking iteration took the row classes keeps them for its later, smaller iterations and for the copies
		// of round 2 (they are short next to the first one, and a row launch of 10^5 items still beats the wavefront classes)
		if (iter == 0 && ctr.n_items >= a.row_min_items && a.row_min_items > 4096) a.row_min_items = 4096;
		// 2. round 1 (one launch per band class)
		int64_t rel = 0;
		bool kept = false;
		if (ctr.n_items > 0) {
			// Trace rows are kept for the walk of round 2 -- unless they do not fit, or so few of the targets can survive the
			// culling (at most -k per query) that sweeping all of them for scores only (18 VALU instructions per packed cell
			// against 31 with trace bits) and the survivors a second time is less work: 18 + 31 f < 31 for a surviving fraction
			// f < 0.42 (C2skew: 125 targets per query, f <= 0.2; C2, C3: f = 0.8 / 0.56, rows kept)
			const bool few_survive = ctr.window_bound * 100 < ctr.window_targets * (unsigned long long)tuning().extend_resweep_below_pct;
			kept = keep_traces_dev() && !few_survive && trace_used + (size_t)ctr.total_rows <= trace_budget;
			DevBuf* arena = nullptr;
			if (kept) if (int rc = arena_for((size_t)ctr.total_rows, arena, rel)) return rc;
			if (iter == 0) tr.lap("trace arena");
			HIP_TRY(hipEventRecord(c->ev0, st));
			if (int rc = dmnd_sweep_classes(c, c, a.items + a.item_base, ctr.class_count, ctr.class_max_steps, EXT_CLASSES, reinterpret_cast<const int32_t*>(a.order), a.off_slot, a.pairs,
				a.off_item + a.item_base, kept ? arena->as<uint8_t>() : nullptr, a.ends + a.item_base)) return rc;
			HIP_TRY(hipEventRecord(c->ev1, st));
		}
		// 3. best HSP per target, append_hits, next window; and -- in case that was the last chunk of every query -- final culling + round-2 list
		HIP_TRY(launch_ext_append(a, ctr.n_items, kept, rel, st));
		const uint32_t n_items_iter = ctr.n_items;
		HIP_TRY(copy_now(st, c->ext_host.p, a.ctr, sizeof(ExtCounters), hipMemcpyDeviceToHost));
		ctr = *c->ext_host.as<ExtCounters>();
		if (n_items_iter > 0) { float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1)); ms_sweeps += ms; }
		items_total += n_items_iter;
		a.item_base += n_items_iter;
		if (tr.on && (iter > 0 || ctr.n_active > 0)) std::fprintf(stderr, "dmnd_extend (device half): chunk %d: %u DpTargets%s, %.1f MB of trace rows kept so far, %u queries go on\n", iter, n_items_iter, kept ? "" : " (scores only)", (double)trace_used / 1048576.0, ctr.n_active);
		if (ctr.n_active == 0) break;
	}
	tr.lap("sweeps, culling");
	// 4. round 2: the survivors whose trace rows were not kept are swept again with traceback (copies of their items, one more
	// iteration), then one walk over all survivors' traces, then the records
	if (ctr.n_kept > nR) return fail(DMND_E_CAP, "dmnd_extend: more device records than -k allows");
	const uint32_t n_kept = ctr.n_kept;
	if (ctr.n_resweep > 0) {
		HIP_TRY(launch_ext_resweep(a, n_kept, st));
		HIP_TRY(copy_now(st, c->ext_host.p, a.ctr, sizeof(ExtCounters), hipMemcpyDeviceToHost));
		ctr = *c->ext_host.as<ExtCounters>();
		// (at most -k survivors per query: their rows are not held against the budget of round 1, only against a hard limit)
		if ((size_t)ctr.total_rows > std::max(c->trace_arena_max * 4, (size_t)4 << 30)) return fail(DMND_E_NOMEM, "dmnd_extend: the trace rows of round 2 exceed the trace limit (4 x DMND_TRACE_ARENA_MB, at least 4 GB)");
		DevBuf* arena = nullptr;
		int64_t rel = 0;
		if (int rc = arena_for((size_t)ctr.total_rows, arena, rel)) return rc;
		HIP_TRY(hipEventRecord(c->ev0, st));
		if (int rc = dmnd_sweep_classes(c, c, a.items + a.item_base, ctr.class_count, ctr.class_max_steps, EXT_CLASSES, reinterpret_cast<const int32_t*>(a.order), a.off_slot, a.pairs,
			a.off_item + a.item_base, arena->as<uint8_t>(), a.ends + a.item_base)) return rc;
		HIP_TRY(hipEventRecord(c->ev1, st));
		HIP_TRY(launch_ext_rewalk(a, ctr.n_items, n_kept, rel, st));
		HIP_TRY(sync_stream(st));
		float ms = 0.f;
		HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
		ms_sweeps2 += ms;
		tr.lap("round-2 sweeps");
	}
	ctr.n_kept = n_kept;
	HIP_TRY(hipEventRecord(c->ev1, st));
	if (ctr.n_kept > 0) {
		TracebackArgs t;
		t.qblock = c->block[DMND_QUERY].as<int8_t>(); t.tblock = c->block[DMND_TARGET].as<int8_t>(); t.cbs = c->cbs_len > 0 ? c->cbs.as<int8_t>() : nullptr;
		t.matrix = c->matrix.as<int8_t>(); t.matrices = nullptr;
		t.items = a.items; t.order = a.r2_order; t.p_of_slot = a.r2_p; t.trace_off = a.r2_off; t.transcript_off = a.r2_tr;
		t.trace = c->ext_trace.as<uint8_t>(); t.transcript = nullptr; t.ends = a.ends; t.hsps = a.hsps; t.status = &a.ctr->tb_status;
		t.n = ctr.n_kept; t.gap_open = c->params.gap_open; t.gap_extend = c->params.gap_extend;
		HIP_TRY(launch_traceback(t, st));
	}
	HIP_TRY(hipEventRecord(c->ev2, st));
	HIP_TRY(launch_ext_records(a, ctr.n_kept, st));
	const size_t h_ctr = 0, h_qstate = (sizeof(ExtCounters) + 63) & ~(size_t)63, h_records = (h_qstate + nQ + 63) & ~(size_t)63,
		h_bytes = h_records + (size_t)ctr.n_kept * sizeof(dmnd_match);
	if (int rc = c->ext_host.ensure(h_bytes)) return rc;
	char* hp = c->ext_host.as<char>();
	HIP_TRY(hipMemcpyAsync(hp + h_ctr, a.ctr, sizeof(ExtCounters), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(hp + h_qstate, a.qstate, nQ, hipMemcpyDeviceToHost, st));
	if (ctr.n_kept) HIP_TRY(hipMemcpyAsync(hp + h_records, a.records, (size_t)ctr.n_kept * sizeof(dmnd_match), hipMemcpyDeviceToHost, st));
	HIP_TRY(sync_stream(st));
	ctr = *reinterpret_cast<const ExtCounters*>(hp + h_ctr);
	tr.lap("walk, records, copy");
	if (ctr.tb_status != 0) return fail(ctr.tb_status, ctr.tb_status == DMND_E_TRACEBACK ? "Traceback error." : "transcript slot too small");
	float ms2 = 0.f;
	HIP_TRY(hipEventElapsedTime(&ms2, c->ev1, c->ev2));
	ms_walk = ms2;
	qstate.assign(hp + h_qstate, hp + h_qstate + nQ);
	// 4. the host's own e-value and bit score in every record; the device ordered a query's records by ITS e-values -- checked,
	// and put right where the two disagree
	const dmnd_match* rec = reinterpret_cast<const dmnd_match*>(hp + h_records);
	records.assign(rec, rec + ctr.n_kept);
	const std::vector<int64_t>& ql = c->limits[DMND_QUERY];
	const std::vector<int64_t>& tl = c->limits[DMND_TARGET];
	const size_t n = records.size(), per = 4096, n_chunks = (n + per - 1) / per;
	parallel_for(n_chunks, std::max(1, std::min(threads, (int)((n + 16383) / 16384))), [&](size_t ci, int) {
		for (size_t i = ci * per; i < std::min(n, (ci + 1) * per); ++i) {
			dmnd_match& m = records[i];
			m.evalue = E.evalue(m.hsp.score, (unsigned)(ql[m.query + 1] - ql[m.query] - 1), (unsigned)(tl[m.target + 1] - tl[m.target] - 1));
			m.bit_score = E.bitscore(m.hsp.score);
		}
	});
	bool reordered = false;
	for (size_t b = 0; b < n;) {
		size_t e = b + 1;
		bool sorted = true;
		while (e < n && records[e].query == records[b].query) { sorted &= !match_less(records[e], records[e - 1]); ++e; }
		if (!sorted) { std::sort(records.begin() + (ptrdiff_t)b, records.begin() + (ptrdiff_t)e, match_less); reordered = true; }
		b = e;
	}
	// 5. ... and back into the copy in HBM: the records stay there, complete, for a join on the device (dmnd_extend_records_device,
	// dmnd_join_contexts_device) -- 16 bytes per record up instead of 104 down and up again
	if (n > 0) {
		if (reordered) HIP_TRY(hipMemcpyAsync(a.records, records.data(), n * sizeof(dmnd_match), hipMemcpyHostToDevice, st));
		else {
			if (int rc = c->ext_ev.ensure(n * 2 * sizeof(double))) return rc;
			if (int rc = c->ext_host.ensure(h_bytes + n * 2 * sizeof(double) + 64)) return rc;      // (the pairs go up from page-locked memory, behind the records)
			double* pairs = reinterpret_cast<double*>(c->ext_host.as<char>() + ((h_bytes + 63) & ~(size_t)63));
			for (size_t i = 0; i < n; ++i) { pairs[2 * i] = records[i].evalue; pairs[2 * i + 1] = records[i].bit_score; }
			HIP_TRY(hipMemcpyAsync(c->ext_ev.p, pairs, n * 2 * sizeof(double), hipMemcpyHostToDevice, st));
			HIP_TRY(launch_ext_patch(a.records, c->ext_ev.as<double>(), (uint32_t)n, st));
		}
		HIP_TRY(sync_stream(st));
	}
	c->ext_records_dev = a.records; c->ext_records_n = (int64_t)n;
	tr.lap("host e-values, order check");
	c->ext_stats[0] += (double)items_total; c->ext_stats[1] += (double)ctr.n_kept;
	c->ext_stats[2] += (double)ctr.cells1; c->ext_stats[3] += (double)ctr.cells2;
	c->ext_stats[9] += ms_sweeps; c->ext_stats[10] += ms_sweeps2; c->ext_stats[11] += ms_walk;
	size_t n_eligible = 0;
	for (uint8_t x : qstate) n_eligible += x != EXT_Q_HOST;
	c->ext_dev_stats[0] = (double)n_eligible; c->ext_dev_stats[1] = (double)(ctr.n_ambiguous + ctr.n_saturated); c->ext_dev_stats[2] = (double)items_total; c->ext_dev_stats[3] = (double)ctr.n_kept;
	c->ext_dev_stats[4] = (double)ctr.diag_steps; c->ext_dev_stats[5] = (double)ctr.lane_steps;
	c->ext_dev_stats[6] = (double)ctr.cells2; c->ext_dev_stats[7] = (double)ctr.cells_again; c->ext_dev_stats[8] = ms_sweeps2;
	done = true;
	return DMND_OK;
}

