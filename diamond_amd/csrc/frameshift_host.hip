// frameshift_host.hip -- the extension stage of frameshift alignment (blastx -F N [--range-culling]).
//
// With a frameshift penalty the reference leaves Extension::extend aside and runs its older per-query "mapper"
// (/root/reference/src/align/align.cpp:168-172 -> legacy_pipeline :119-155):
//   QueryMapper::init / count_targets / load_targets      src/align/legacy/query_mapper.cpp:88-160   x-drop extension of every seed hit, targets
//   Pipeline::run                                         src/align/legacy/banded_swipe_pipeline.cpp:189-264
//     Target::ungapped_stage, rank_targets / range_ranking  :36-44, query_mapper.cpp:162-185, banded_swipe_pipeline.cpp:136-151
//     run_swipe(score only) + score_only_culling            :157-171, query_mapper.cpp:187-213   (more targets than -k, or --top)
//     run_swipe(traceback), Target::finish (inner culling)  :103-110, query_mapper.cpp:319-336
//   QueryMapper::generate_output                          query_mapper.cpp:215-317               order, target culling, filters
//   GlobalCulling / RangeCulling, IntervalPartition       src/output/target_culling.h, src/util/geo/interval_partition.h
// Here the same decisions are taken for all reads of a query block in step: ONE x-drop launch over every seed hit, ONE score-only
// three-frame launch over the DpTargets of all reads that need it, ONE traceback launch (dmnd_frameshift_swipe), host threads in
// between. No composition bias anywhere on this path (the mapper's x-drop stage and the three-frame sweep have none).
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>
#include "ctx.h"
#include "host_pool.h"
#include "bias_kernels.h"
#include "frameshift_core.h"
#include "read_coverage.h"

using namespace dmnd;

namespace {

struct Hit3 { int frame, target, spos, qpos, score; };         // SeedHit with ungapped.score > 0 (query_mapper.h:33-100)
struct Hsp3 { dmnd_fs_hsp h; double evalue; int strand; };

struct Target3 {
	int target = 0;
	size_t begin = 0, end = 0;                                 // its hits
	int filter_score = 0;
	double filter_evalue = DBL_MAX;
	Hit3 top;
	std::vector<Hsp3> hsps;
};

struct Cfg {
	int frame_shift, k, max_hsps, channels;
	bool range_culling;
	double range_cover, top, min_id, query_cover, subject_cover, min_bit_score, max_evalue;
	const Evaluer* ev;
	bool reported(int score, double evalue) const { return min_bit_score != 0.0 ? ev->bitscore(score) >= min_bit_score : evalue <= max_evalue; }
};

bool by_score(const Target3& a, const Target3& b) { return a.filter_score > b.filter_score || (a.filter_score == b.filter_score && a.target < b.target); }
bool by_evalue(const Target3& a, const Target3& b) { return a.filter_evalue < b.filter_evalue || (a.filter_evalue == b.filter_evalue && by_score(a, b)); }

int frame_length(int dna_len, int offset) { return std::max((dna_len - offset) / 3, 0); }

// TranslatedPosition::absolute_interval (basic/translated_position.h:130-136)
Interval read_interval(int b, int e, int frame, int dna_len)
{
	const int off = frame % 3;
	if (frame < 3) return Interval{ off + 3 * b, off + 3 * e };
	return Interval{ dna_len - (off + 3 * e), dna_len - (off + 3 * b) };
}

// Target::ungapped_query_range (banded_swipe_pipeline.cpp:46-53)
Interval ungapped_read_range(const Target3& t, int tlen, int dna_len)
{
	const int i0 = std::max(t.top.qpos - t.top.spos, 0), i1 = std::min(t.top.qpos + tlen - t.top.spos, frame_length(dna_len, t.top.frame % 3));
	return read_interval(i0, i1, t.top.frame, dna_len);
}

// the reference's TargetCulling for one query: GlobalCulling (-k / --top over whole targets) or RangeCulling (per read range)
struct Culling {
	enum { FINISHED = 0, NEXT = 1, INCLUDE = 2 };
	const Cfg& c;
	int64_t n = 0;
	double top_score = 0;
	Coverage cover;
	explicit Culling(const Cfg& c) : c(c), cover(c.k) {}
	int cull(const Target3& t, double* cov_out = nullptr) const
	{
		if (cov_out) *cov_out = 0;
		if (!c.range_culling) {
			if (top_score == 0) return INCLUDE;
			if (c.top >= 0) return (1.0 - c.ev->bitscore(t.filter_score) / top_score) * 100.0 <= c.top ? INCLUDE : FINISHED;
			return n < c.k ? INCLUDE : FINISHED;
		}
		int cv = 0, l = 0;
		for (const Hsp3& h : t.hsps) {
			const Interval r{ h.h.read_begin, h.h.read_end };
			if (c.top < 0) cv += cover.covered(r);
			else cv += cover.covered_max(r, (int)((double)h.h.score / (1.0 - c.top / 100.0)));
			l += r.length();
		}
		const double cov = (double)cv / l;
		if (cov_out) *cov_out = cov;
		return cov * 100.0 < c.range_cover ? INCLUDE : NEXT;
	}
	void add(const Target3& t)
	{
		if (!c.range_culling) { if (top_score == 0) top_score = c.ev->bitscore(t.filter_score); ++n; return; }
		for (const Hsp3& h : t.hsps) cover.insert(Interval{ h.h.read_begin, h.h.read_end }, h.h.score);
	}
};

struct Read {
	uint32_t query = 0;
	int dna_len = 0;
	std::vector<Hit3> hits;
	std::vector<Target3> targets;
	bool score_pass = false;
	size_t item_begin = 0, item_end = 0;
	std::vector<int> item_target;                              // per item of the launch at hand: index into targets
};

// Target::add / add_strand (banded_swipe_pipeline.cpp:55-96): the DpTargets of one target, forward strand into vf, reverse into vr
void add_target(Read& r, size_t ti, int qlen0, int tlen, std::vector<dmnd_fs_target>& vf, std::vector<dmnd_fs_target>& vr, std::vector<int>& tf, std::vector<int>& tr)
{
	Target3& t = r.targets[ti];
	std::stable_sort(r.hits.begin() + (ptrdiff_t)t.begin, r.hits.begin() + (ptrdiff_t)t.end, [](const Hit3& x, const Hit3& y) {      // SeedHit::compare_diag_strand2
		const int sx = x.frame >= 3, sy = y.frame >= 3, dx = x.qpos - x.spos, dy = y.qpos - y.spos;
		return sx < sy || (sx == sy && (dx < dy || (dx == dy && x.spos < y.spos)));
	});
	size_t split = t.begin;
	while (split < t.end && r.hits[split].frame < 3) ++split;
	const int band = 32, d_min = -(tlen - 1), d_max = qlen0 - 1;      // config.padding (set to 32 by Pipeline::run)
	auto strand = [&](size_t b, size_t e, int s, std::vector<dmnd_fs_target>& v, std::vector<int>& owner) {
		if (b == e) return;
		auto emit = [&](int d0, int d1) {
			dmnd_fs_target it;
			std::memset(&it, 0, sizeof it);
			it.d_begin = d0; it.d_end = d1; it.strand = s; it.target_len = tlen;
			it.cols = dmnd_banded_cols(0, tlen, d0, d1);          // DpTarget(subject, subject.length(), d0, d1, target_idx, 0)
			v.push_back(it); owner.push_back((int)ti);
		};
		int d0 = std::max(r.hits[b].qpos - r.hits[b].spos - band, d_min), d1 = std::min(r.hits[b].qpos - r.hits[b].spos + band, d_max);
		for (size_t i = b + 1; i < e; ++i) {
			const int d = r.hits[i].qpos - r.hits[i].spos;
			if (d - d1 <= band) d1 = std::min(d + band, d_max);
			else { emit(d0, d1); d0 = std::max(d - band, d_min); d1 = std::min(d + band, d_max); }
		}
		emit(d0, d1);
	};
	strand(t.begin, split, 0, vf, tf);
	strand(split, t.end, 1, vr, tr);
}

// Target::inner_culling (query_mapper.cpp:319-336): best first, an alignment gone if half of its read range lies in a better one
void inner_culling(Target3& t)
{
	std::stable_sort(t.hsps.begin(), t.hsps.end(), [](const Hsp3& a, const Hsp3& b) {      // Hsp::operator< (d_begin is 0 on this path)
		return a.h.score > b.h.score || (a.h.score == b.h.score && a.h.read_begin < b.h.read_begin);
	});
	if (!t.hsps.empty()) { t.filter_score = t.hsps[0].h.score; t.filter_evalue = t.hsps[0].evalue; }
	else { t.filter_score = 0; t.filter_evalue = DBL_MAX; }
	std::vector<Hsp3> kept;
	for (const Hsp3& h : t.hsps) {
		const Interval r{ h.h.read_begin, h.h.read_end };
		bool enveloped = false;
		for (const Hsp3& k : kept)
			if ((double)r.overlap(Interval{ k.h.read_begin, k.h.read_end }) / (double)r.length() >= 0.5) { enveloped = true; break; }
		if (!enveloped) kept.push_back(h);
	}
	t.hsps.swap(kept);
}

}  // namespace

int dmnd_extend_frameshift(dmnd_ctx* c, const int8_t* qdata, const int8_t* tdata, const dmnd_seed_hit* hits, int64_t n_hits, int threads,
	std::vector<dmnd_match>& out, uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used)
{
	(void)qdata;
	const std::vector<int64_t>& ql = c->limits[DMND_QUERY];
	const std::vector<int64_t>& tl = c->limits[DMND_TARGET];
	if (c->query_contexts != 6) return fail(DMND_E_ARG, "Frameshift alignments are only supported for translated searches.");
	if (c->source_lens.size() != (ql.size() - 1) / 6) return fail(DMND_E_ARG, "dmnd_extend: frameshift alignment needs the read lengths (dmnd_set_query_source_lengths)");
	Cfg cfg;
	cfg.frame_shift = c->frame_shift; cfg.k = c->max_target_seqs; cfg.max_hsps = c->max_hsps; cfg.channels = c->fs_channels;
	cfg.range_culling = c->range_culling; cfg.range_cover = c->range_cover; cfg.top = c->top_percent;
	cfg.min_id = c->min_id; cfg.query_cover = c->query_cover; cfg.subject_cover = c->subject_cover; cfg.min_bit_score = c->min_bit_score;
	cfg.max_evalue = c->params.max_evalue; cfg.ev = &c->evaluer;
	threads = std::max(1, threads);
	out.clear();
	if (transcript_used) *transcript_used = 0;
	if (n_hits == 0) return DMND_OK;
	// 1. x-drop extension of every seed hit, without bias (QueryMapper::count_targets, query_mapper.cpp:104-134)
	HIP_TRY(hipSetDevice(c->device));
	if (int rc = c->xd_hits.ensure((size_t)n_hits * sizeof(dmnd_seed_hit))) return rc;
	if (int rc = c->xd_out.ensure((size_t)n_hits * sizeof(XdropSeg))) return rc;
	if (int rc = c->xd_host.ensure((size_t)n_hits * sizeof(XdropSeg))) return rc;
	HIP_TRY(hipMemcpyAsync(c->xd_hits.p, hits, (size_t)n_hits * sizeof(dmnd_seed_hit), hipMemcpyHostToDevice, c->stream));
	XdropArgs xa;
	xa.qblock = c->block[DMND_QUERY].as<int8_t>(); xa.tblock = c->block[DMND_TARGET].as<int8_t>(); xa.cbs = nullptr;
	xa.qlimits = c->d_limits[DMND_QUERY].as<int64_t>(); xa.matrix = c->matrix.as<int8_t>();
	xa.hits = c->xd_hits.as<dmnd_seed_hit>(); xa.n_hits = n_hits; xa.out = c->xd_out.as<XdropSeg>();
	xa.xdrop = (int)std::ceil((12.3 * 0.69314718055994530941723212145818 + std::log(c->params.K)) / c->params.lambda);      // config.raw_ungapped_xdrop
	HIP_TRY(launch_xdrop_segs(xa, c->stream));
	HIP_TRY(copy_now(c->stream, c->xd_host.p, c->xd_out.p, (size_t)n_hits * sizeof(XdropSeg), hipMemcpyDeviceToHost));
	const XdropSeg* xd = c->xd_host.as<XdropSeg>();

	// 2. per read: targets, their best ungapped hit, ranking
	std::vector<std::pair<int64_t, int64_t>> ranges;
	for (int64_t i = 0; i < n_hits;) { int64_t j = i; while (j < n_hits && hits[j].query / 6 == hits[i].query / 6) ++j; ranges.push_back({ i, j }); i = j; }
	std::vector<Read> reads(ranges.size());
	const int64_t nt = (int64_t)tl.size() - 1;
	auto target_len = [&](int t) { return (int)(tl[(size_t)t + 1] - tl[(size_t)t] - 1); };
	parallel_for(ranges.size(), threads, [&](size_t ri, int) {
		Read& r = reads[ri];
		const int64_t b = ranges[ri].first, e = ranges[ri].second;
		r.query = hits[b].query / 6;
		r.dna_len = c->source_lens[r.query];
		std::vector<int64_t> idx((size_t)(e - b));
		for (int64_t i = b; i < e; ++i) idx[(size_t)(i - b)] = i;
		std::sort(idx.begin(), idx.end(), [&](int64_t x, int64_t y) {      // Search::Hit::CmpSubject
			const dmnd_seed_hit &a = hits[x], &h = hits[y];
			return a.subject < h.subject || (a.subject == h.subject && (a.query < h.query || (a.query == h.query && a.seed_offset < h.seed_offset)));
		});
		const int64_t* it = tl.data();
		for (int64_t x : idx) {
			if (xd[x].score <= 0) continue;
			it = std::upper_bound(it, tl.data() + nt + 1, hits[x].subject) - 1;
			const int t = (int)(it - tl.data());
			if (r.targets.empty() || r.targets.back().target != t) { Target3 nt3; nt3.target = t; nt3.begin = r.hits.size(); r.targets.push_back(nt3); }
			r.hits.push_back(Hit3{ (int)(hits[x].query % 6), t, (int)(hits[x].subject - tl[(size_t)t]), hits[x].seed_offset, xd[x].score });
			r.targets.back().end = r.hits.size();
		}
		for (Target3& t : r.targets) {                              // Target::ungapped_stage
			t.top = r.hits[t.begin];
			for (size_t i = t.begin + 1; i < t.end; ++i) if (r.hits[i].score > t.top.score) t.top = r.hits[i];
			t.filter_score = t.top.score;
		}
		if (r.targets.empty()) return;
		std::stable_sort(r.targets.begin(), r.targets.end(), by_score);
		if (!cfg.range_culling) {                                   // QueryMapper::rank_targets(0.4, 1e3, k)
			const double ratio = 0.4, factor = 1e3;
			int score;
			if (cfg.top >= 0) score = (int)((double)r.targets[0].filter_score * (1.0 - cfg.top / 100.0) * ratio);
			else score = (int)((double)r.targets[std::min<size_t>(r.targets.size(), (size_t)cfg.k) - 1].filter_score * ratio);
			const int64_t cap = cfg.top >= 0 ? INT64_MAX : (int64_t)((double)cfg.k * factor);
			size_t i = 0;
			for (; i < r.targets.size(); ++i) if (r.targets[i].filter_score < score || (int64_t)i >= cap) break;
			r.targets.resize(i);
		}
		else {                                                      // Pipeline::range_ranking
			const double rr = 0.4;
			Coverage ip(cfg.k);
			std::vector<Target3> kept;
			for (Target3& t : r.targets) {
				const Interval rg = ungapped_read_range(t, target_len(t.target), r.dna_len);
				bool outranked;
				if (cfg.top < 0) outranked = (double)ip.covered_min(rg, (int)((double)t.filter_score / rr)) / rg.length() * 100.0 >= cfg.range_cover;
				else outranked = (double)ip.covered_max(rg, (int)((double)t.filter_score / rr / (1.0 - cfg.top / 100.0))) / rg.length() * 100.0 >= cfg.range_cover;
				if (!outranked) { ip.insert(rg, t.filter_score); kept.push_back(std::move(t)); }
			}
			r.targets.swap(kept);
		}
		r.score_pass = (int64_t)r.targets.size() > cfg.k || cfg.top >= 0;
	});

	// 3. / 4. the two sweeps: score only for the reads with more targets than they may report (then culling), traceback for all
	std::vector<dmnd_fs_target> items;
	std::vector<dmnd_fs_hsp> res;
	std::vector<uint8_t> arena;
	std::vector<Hsp3> kept_hsps;
	for (int pass = 0; pass < 2; ++pass) {
		const bool score_only = pass == 0;
		items.clear();
		int group = 0;
		for (Read& r : reads) {
			r.item_begin = items.size();
			r.item_target.clear();
			if (r.targets.empty() || (score_only && !r.score_pass)) { r.item_end = items.size(); continue; }
			const uint32_t q0 = r.query * 6;
			const int qlen0 = (int)(ql[q0 + 1] - ql[q0] - 1);
			std::vector<dmnd_fs_target> vf, vr;
			std::vector<int> tf, tr;
			for (size_t ti = 0; ti < r.targets.size(); ++ti) { r.targets[ti].hsps.clear(); add_target(r, ti, qlen0, target_len(r.targets[ti].target), vf, vr, tf, tr); }
			for (int s = 0; s < 2; ++s) {
				std::vector<dmnd_fs_target>& v = s ? vr : vf;
				std::vector<int>& owner = s ? tr : tf;
				if (!score_only) {
					// the traceback sweep takes the targets one by one in DpTarget order: the order its Hsps reach their targets' lists
					std::vector<size_t> ord(v.size());
					for (size_t i = 0; i < ord.size(); ++i) ord[i] = i;
					std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) {
						const dmnd_fs_target &a = v[x], &b = v[y];
						const int ba = (a.d_end - a.d_begin) / 24, bb = (b.d_end - b.d_begin) / 24, ta = a.cols / 400, tb = b.cols / 400;
						return ba < bb || (ba == bb && (ta < tb || (ta == tb && std::max(a.d_end - 1, 0) < std::max(b.d_end - 1, 0))));
					});
					std::vector<dmnd_fs_target> v2; std::vector<int> o2;
					for (size_t i : ord) { v2.push_back(v[i]); o2.push_back(owner[i]); }
					v.swap(v2); owner.swap(o2);
				}
				for (size_t i = 0; i < v.size(); ++i) {
					dmnd_fs_target& it = v[i];
					for (int f = 0; f < 3; ++f) { const uint32_t qc = q0 + (uint32_t)(3 * s + f); it.frame_off[f] = ql[qc]; it.frame_len[f] = (int32_t)(ql[qc + 1] - ql[qc] - 1); }
					it.target_off = tl[(size_t)r.targets[(size_t)owner[i]].target];
					it.dna_len = r.dna_len; it.group = group;
					items.push_back(it);
					r.item_target.push_back(owner[i]);
				}
				++group;
			}
			r.item_end = items.size();
		}
		if (items.empty()) continue;
		res.assign(items.size(), dmnd_fs_hsp());
		int64_t used = 0;
		if (!score_only) {
			size_t need = 16;
			for (const dmnd_fs_target& it : items) need += (size_t)(2 * it.target_len + it.frame_len[0] + 65);
			arena.resize(need);
		}
		if (int rc = dmnd_frameshift_swipe(c, items.data(), (int64_t)items.size(), score_only ? 1 : 0, cfg.frame_shift, cfg.channels, res.data(),
			score_only ? nullptr : arena.data(), score_only ? 0 : (int64_t)arena.size(), &used)) return rc;
		parallel_for(reads.size(), threads, [&](size_t ri, int) {
			Read& r = reads[ri];
			if (r.item_end == r.item_begin) return;
			for (size_t x = r.item_begin; x < r.item_end; ++x) {
				const dmnd_fs_hsp& h = res[x];
				if (h.score <= 0) continue;
				const double ev = c->evaluer.evalue(h.score, (unsigned)items[x].frame_len[0], (unsigned)items[x].target_len);
				if (!cfg.reported(h.score, ev)) continue;             // score_matrix.report_cutoff, banded_3frame_swipe.cpp:522-524
				r.targets[(size_t)r.item_target[x - r.item_begin]].hsps.push_back(Hsp3{ h, ev, items[x].strand });
			}
			if (score_only) {
				// Target::set_filter_score, then QueryMapper::score_only_culling (query_mapper.cpp:187-213)
				for (Target3& t : r.targets) {
					t.filter_score = 0; t.filter_evalue = DBL_MAX;
					for (const Hsp3& h : t.hsps) { t.filter_score = std::max(t.filter_score, h.h.score); t.filter_evalue = std::min(t.filter_evalue, h.evalue); }
				}
				std::stable_sort(r.targets.begin(), r.targets.end(), cfg.top < 0 ? by_evalue : by_score);
				Culling cull(cfg);
				std::vector<Target3> kept;
				for (Target3& t : r.targets) {
					if (!cfg.reported(t.filter_score, t.filter_evalue)) break;
					double cov = 0;
					const int code = cull.cull(t, &cov);
					if (code == Culling::FINISHED) break;
					if (code == Culling::NEXT) continue;
					if (cov < 0.1) cull.add(t);
					kept.push_back(std::move(t));
				}
				r.targets.swap(kept);
			}
			else
				for (Target3& t : r.targets) inner_culling(t);      // Target::finish
		});
	}

	// 5. QueryMapper::generate_output: order, target culling, filters, -max-hsps
	int64_t used = 0;
	for (Read& r : reads) {
		std::stable_sort(r.targets.begin(), r.targets.end(), cfg.top < 0 ? by_evalue : by_score);
		Culling cull(cfg);
		for (Target3& t : r.targets) {
			const int tlen = target_len(t.target);
			std::vector<Hsp3> passed;                               // Target::apply_filters
			for (const Hsp3& h : t.hsps) {
				const double id = (double)h.h.identities * 100.0 / h.h.length, qc = (double)(h.h.read_end - h.h.read_begin) * 100.0 / r.dna_len,
					sc = (double)(h.h.s_end - h.h.s_begin) * 100.0 / tlen;
				if (id < cfg.min_id || qc < cfg.query_cover || sc < cfg.subject_cover) continue;
				passed.push_back(h);
			}
			t.hsps.swap(passed);
			if (t.hsps.empty()) continue;
			const int code = cull.cull(t);
			if (code == Culling::NEXT) continue;
			if (code == Culling::FINISHED) break;
			cull.add(t);
			int n = 0;
			for (const Hsp3& h : t.hsps) {
				if (cfg.max_hsps > 0 && n >= cfg.max_hsps) break;
				dmnd_match m;
				std::memset(&m, 0, sizeof m);
				m.query = r.query; m.target = (uint32_t)t.target; m.frame = h.h.frame; m.evalue = h.evalue; m.bit_score = c->evaluer.bitscore(h.h.score);
				m.read_begin = h.h.read_begin; m.read_end = h.h.read_end;
				m.hsp.score = h.h.score; m.hsp.q_begin = h.h.q_begin; m.hsp.q_end = h.h.q_end; m.hsp.s_begin = h.h.s_begin; m.hsp.s_end = h.h.s_end;
				m.hsp.length = h.h.length; m.hsp.identities = h.h.identities; m.hsp.mismatches = h.h.mismatches; m.hsp.positives = h.h.positives;
				m.hsp.gap_openings = h.h.gap_openings; m.hsp.gaps = h.h.gaps; m.hsp.transcript_len = h.h.transcript_len; m.hsp.transcript_off = -1;
				if (transcript && h.h.transcript_off >= 0) {
					const int64_t len = (int64_t)h.h.transcript_len + 1;
					if (used + len > transcript_cap) return fail(DMND_E_CAP, "dmnd_extend: transcript arena too small");
					std::memcpy(transcript + used, arena.data() + h.h.transcript_off, (size_t)len);
					m.hsp.transcript_off = used;
					used += len;
				}
				out.push_back(m);
				++n;
			}
		}
	}
	if (transcript_used) *transcript_used = used;
	return DMND_OK;
}
