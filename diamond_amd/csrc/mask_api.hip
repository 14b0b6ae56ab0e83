// mask_api.hip -- C ABI of tantan repeat masking: dmnd_mask_block.
// Replaces mask_seqs(seqs, Masking::get(), true, MaskingAlgo::TANTAN) (src/masking/masking.cpp:225-251) as the reference
// applies it to the reference block (run/double_indexed.cpp:122-127) and to the query block (:737-740).
#pragma clang fp contract(off)
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>
#include "ctx.h"
#include "mask_kernels.h"
#include "seg_mask.h"
#include <atomic>
#include <thread>

using namespace dmnd;

namespace {

// f(lambda) = sum(inverse(exp(lambda * S))) - 1 over the 20 standard residues (cbrc::LambdaCalculator, lib/tantan/LambdaCalculator.cc).
// valid (optional): the row and column sums of the inverse -- the letter probabilities the matrix implies -- all lie in [0, 1]
// (LambdaCalculator::check_lambda).
bool inv_sum(const int8_t* m8, double lambda, double& f, bool* valid = nullptr)
{
	const int n = 20;
	double A[20][40];
	for (int i = 0; i < n; ++i)
		for (int j = 0; j < n; ++j) { A[i][j] = std::exp(lambda * (double)m8[i * 32 + j]); A[i][n + j] = i == j ? 1.0 : 0.0; }
	for (int k = 0; k < n; ++k) {
		int p = k;
		for (int i = k + 1; i < n; ++i) if (std::fabs(A[i][k]) > std::fabs(A[p][k])) p = i;
		if (std::fabs(A[p][k]) < 1e-12) return false;
		if (p != k) for (int j = 0; j < 2 * n; ++j) std::swap(A[k][j], A[p][j]);
		const double piv = A[k][k];
		for (int j = 0; j < 2 * n; ++j) A[k][j] /= piv;
		for (int i = 0; i < n; ++i) {
			if (i == k) continue;
			const double m = A[i][k];
			if (m != 0.0) for (int j = 0; j < 2 * n; ++j) A[i][j] -= m * A[k][j];
		}
	}
	long double acc = 0;
	for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) acc += A[i][n + j];
	f = (double)(acc - 1.0L);
	if (valid) {
		*valid = true;
		for (int i = 0; i < n; ++i) {
			double row = 0, col = 0;
			for (int j = 0; j < n; ++j) { row += A[i][n + j]; col += A[j][n + i]; }
			if (!(row >= 0.0 && row <= 1.0) || !(col >= 0.0 && col <= 1.0)) *valid = false;
		}
	}
	return std::isfinite(f);
}

// The scale of the matrix as the reference's masking derives it (Masking::Masking, masking/masking.cpp:132-143): the lambda in
// (ub * 1e-6, ub] with sum(inverse(exp(lambda * S))) = 1 whose implied letter probabilities are valid, ub from the smallest row /
// column maximum (LambdaCalculator::find_ub). The reference brackets the root with random pairs until one bisection ends on a valid
// lambda and gives up with lambda = -1 after 1000 attempts; this scan over the same interval visits every sign change in
// ascending order instead. A matrix without a valid root (PAM250) gets the reference's fallback: -1, i.e. likelihood ratios
// exp(-score) -- which masks nearly everything, exactly as the reference then does.
double masking_lambda(const int8_t* m8)
{
	const int n = 20;
	double r_max_min = 1e300, c_max_min = 1e300;
	int zero_rows = 0, zero_cols = 0;
	for (int i = 0; i < n; ++i) {
		int rmax = -128, rmin = 127, cmax = -128, cmin = 127;
		for (int j = 0; j < n; ++j) {
			rmax = std::max(rmax, (int)m8[i * 32 + j]); rmin = std::min(rmin, (int)m8[i * 32 + j]);
			cmax = std::max(cmax, (int)m8[j * 32 + i]); cmin = std::min(cmin, (int)m8[j * 32 + i]);
		}
		if (rmax == 0 && rmin == 0) ++zero_rows; else if (rmax <= 0 || rmin >= 0) return -1.0; else r_max_min = std::min(r_max_min, (double)rmax);
		if (cmax == 0 && cmin == 0) ++zero_cols; else if (cmax <= 0 || cmin >= 0) return -1.0; else c_max_min = std::min(c_max_min, (double)cmax);
	}
	if (zero_rows == n) return -1.0;
	const double ub = r_max_min > c_max_min ? 1.1 * std::log(1.0 * (n - zero_rows)) / r_max_min : 1.1 * std::log(1.0 * (n - zero_cols)) / c_max_min;
	const double lb = ub * 1e-6;
	const int GRID = 4096;
	double x0 = lb, f0 = 0;
	bool have0 = inv_sum(m8, x0, f0);
	for (int g = 1; g <= GRID; ++g) {
		const double x1 = lb + (ub - lb) * g / GRID;
		double f1 = 0;
		const bool have1 = inv_sum(m8, x1, f1);
		if (have0 && have1 && (f0 < 0) != (f1 < 0)) {
			double lo = x0, hi = x1, flo = f0, fm = 0;
			bool ok = true;
			for (int it = 0; it < 200 && ok; ++it) {
				const double mid = 0.5 * (lo + hi);
				if (mid == lo || mid == hi) break;
				ok = inv_sum(m8, mid, fm);
				if (ok) { if ((fm < 0) == (flo < 0)) { lo = mid; flo = fm; } else hi = mid; }
			}
			bool valid = false;
			double fl = 0, fh = 0;
			if (ok && inv_sum(m8, lo, fl) && inv_sum(m8, hi, fh)) {
				const double lambda = std::fabs(fl) < std::fabs(fh) ? lo : hi;       // the end closer to the root, LambdaCalculator::binary_search
				if (inv_sum(m8, lambda, fm, &valid) && valid && std::fabs(fm) < 1e-3) return lambda;
			}
		}
		x0 = x1; f0 = f1; have0 = have1;
	}
	return -1.0;
}

// (the root search above is 4096 inversions of a 20 x 20 matrix, 6 ms: the reference does it once per run, in Masking's constructor
// -- here once per scoring matrix and process, whatever context asks)
double cached_masking_lambda(const int8_t* m8)
{
	static std::mutex mutex;
	static std::vector<std::pair<std::vector<int8_t>, double>> known;
	std::lock_guard<std::mutex> lock(mutex);
	for (const auto& k : known)
		if (std::memcmp(k.first.data(), m8, 1024) == 0) return k.second;
	known.emplace_back(std::vector<int8_t>(m8, m8 + 1024), masking_lambda(m8));
	return known.back().second;
}

void likelihood_ratios(const int8_t* m8, float* lr)
{
	const double lambda = cached_masking_lambda(m8);
	for (int i = 0; i < 32; ++i)
		for (int j = 0; j < 32; ++j)      // Masking::Masking, masking.cpp:147-153: the 26 alphabet letters, 0 elsewhere
			lr[i * 32 + j] = (i < 26 && j < 26) ? (float)std::exp(lambda * (double)m8[i * 32 + j]) : 0.0f;
}

}

// the process-wide motif table and the lock that makes a context's use of it a snapshot (dmnd_set_context_motif_table gives a
// context a table of its own)
namespace { std::vector<uint64_t> g_motifs; std::mutex g_motifs_mutex; }

extern "C" double dmnd_masking_lambda(const dmnd_params* p)
{
	if (!p) { fail(DMND_E_ARG, "dmnd_masking_lambda: params is NULL"); return 0.0; }
	return cached_masking_lambda(p->matrix8);
}

// ---- SEG (--masking seg): host side, as in the reference ------------------------------------------------------------------------
extern "C" int dmnd_seg_ranges(const int8_t* seq, int32_t len, int32_t* ranges, int32_t cap, int32_t* n)
{
	if (!seq || len < 0 || !n || cap < 0 || (cap > 0 && !ranges)) return fail(DMND_E_ARG, "dmnd_seg_ranges: bad argument");
	const std::vector<seg::Range> r = seg::segments(seq, len);
	*n = (int32_t)r.size();
	if ((int32_t)r.size() > cap) return fail(DMND_E_CAP, "dmnd_seg_ranges: more segments than the output holds");
	for (size_t i = 0; i < r.size(); ++i) { ranges[2 * i] = r[i].begin; ranges[2 * i + 1] = r[i].end; }
	return DMND_OK;
}

extern "C" int dmnd_seg_mask_block(int8_t* data, const int64_t* limits, int64_t n_seqs, int threads, int64_t* n_masked)
{
	if (!data || !limits || n_seqs < 0) return fail(DMND_E_ARG, "dmnd_seg_mask_block: bad argument");
	for (int64_t i = 0; i < n_seqs; ++i)
		if (limits[i + 1] <= limits[i] || limits[i + 1] - limits[i] - 1 > INT32_MAX) return fail(DMND_E_ARG, "dmnd_seg_mask_block: limits are not a SequenceSet layout");
	(void)seg::lnfact();                                    // built once, before the workers start
	std::atomic<int64_t> next(0), masked(0);
	auto worker = [&] {
		int64_t mine = 0;
		for (;;) {
			const int64_t i0 = next.fetch_add(64, std::memory_order_relaxed);
			if (i0 >= n_seqs) break;
			for (int64_t i = i0; i < std::min(i0 + 64, n_seqs); ++i) {
				int8_t* s = data + limits[i];
				const int len = (int)(limits[i + 1] - limits[i] - 1);
				for (const seg::Range& r : seg::segments(s, len))       // Masking::operator(), masking.cpp:186-190: the mask letter over [left, right]
					for (int x = r.begin; x <= r.end; ++x) { s[x] = 23; ++mine; }
			}
		}
		masked += mine;
	};
	const int T = (int)std::max<int64_t>(1, std::min<int64_t>(threads, (n_seqs + 63) / 64));
	std::vector<std::thread> pool;
	for (int t = 1; t < T; ++t) pool.emplace_back(worker);
	worker();
	for (std::thread& t : pool) t.join();
	if (n_masked) *n_masked = masked.load();
	return DMND_OK;
}

extern "C" double dmnd_seg_lnfact(uint32_t n) { return seg::lnfact()(n); }

extern "C" int dmnd_set_motif_table(const uint64_t* codes, int64_t n)
{
	if (n < 0 || (n > 0 && !codes) || n > 8192) return fail(DMND_E_ARG, "dmnd_set_motif_table: at most 8192 motifs");
	std::vector<uint64_t> v(codes, codes + n);
	std::sort(v.begin(), v.end());
	v.erase(std::unique(v.begin(), v.end()), v.end());
	std::lock_guard<std::mutex> lock(g_motifs_mutex);
	g_motifs.swap(v);
	return DMND_OK;
}

extern "C" int dmnd_set_context_motif_table(dmnd_ctx* c, const uint64_t* codes, int64_t n)
{
	if (!c || n < -1 || (n > 0 && !codes) || n > 8192) return fail(DMND_E_ARG, "dmnd_set_context_motif_table: at most 8192 motifs");
	c->own_motifs = n >= 0;
	c->motifs.clear();
	if (n > 0) {
		c->motifs.assign(codes, codes + n);
		std::sort(c->motifs.begin(), c->motifs.end());
		c->motifs.erase(std::unique(c->motifs.begin(), c->motifs.end()), c->motifs.end());
	}
	return DMND_OK;
}

extern "C" int64_t dmnd_motif_table_size(void) { std::lock_guard<std::mutex> lock(g_motifs_mutex); return (int64_t)g_motifs.size(); }

extern "C" int dmnd_soft_mask_block(dmnd_ctx* c, int which, int64_t* n_covered)
{
	if (!c || (which != DMND_QUERY && which != DMND_TARGET)) return fail(DMND_E_ARG, "dmnd_soft_mask_block: bad argument");
	if (!c->block[which].p || c->limits[which].size() < 2) return fail(DMND_E_ARG, "dmnd_soft_mask_block: block must be uploaded with limits");
	if (n_covered) *n_covered = 0;
	c->soft_valid[which] = false;
	if (which == DMND_QUERY) ++c->query_generation;
	// the table of this call: the context's own, or a snapshot of the process-wide one
	std::vector<uint64_t> snapshot;
	if (!c->own_motifs) { std::lock_guard<std::mutex> lock(g_motifs_mutex); snapshot = g_motifs; }
	const std::vector<uint64_t>& g_motifs = c->own_motifs ? c->motifs : snapshot;
	if (g_motifs.empty()) return DMND_OK;                     // no table: nothing is soft-masked (as --motif-masking 0)
	HIP_TRY(hipSetDevice(c->device));
	hipStream_t st = c->stream;
	const std::vector<int64_t>& lim = c->limits[which];
	const int64_t raw = c->block_len[which];
	if (int rc = c->soft[which].ensure((size_t)raw + 64)) return rc;
	if (int rc = c->motif_hit.ensure((size_t)raw + 64)) return rc;
	if (int rc = c->motif_table.ensure(g_motifs.size() * sizeof(uint64_t))) return rc;
	if (int rc = c->counters.ensure(64 * sizeof(unsigned long long))) return rc;
	HIP_TRY(hipMemcpyAsync(c->motif_table.p, g_motifs.data(), g_motifs.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(c->soft[which].p, c->block[which].p, (size_t)raw, hipMemcpyDeviceToDevice, st));
	HIP_TRY(hipMemsetAsync(c->counters.p, 0, sizeof(unsigned long long), st));
	MotifArgs a;
	a.data = c->block[which].as<int8_t>(); a.soft = c->soft[which].as<int8_t>(); a.hit = c->motif_hit.as<uint8_t>();
	a.limits = c->d_limits[which].as<int64_t>(); a.n_seqs = (int64_t)lim.size() - 1; a.begin = lim.front(); a.end = lim.back();
	a.table = c->motif_table.as<uint64_t>(); a.n_table = (int)g_motifs.size(); a.max_range = MOTIF_MAX_RANGE;
	a.n_covered = c->counters.as<unsigned long long>();
	HIP_TRY(launch_motif_mask(a, st));
	unsigned long long n = 0;
	HIP_TRY(copy_now(st, &n, c->counters.p, sizeof(n), hipMemcpyDeviceToHost));
	if (n_covered) *n_covered = (int64_t)n;
	c->soft_valid[which] = true;
	return DMND_OK;
}

static int mask_impl(dmnd_ctx* c, int which, int8_t* host_data, const int32_t* ids, int64_t n_ids, int64_t* n_masked);

extern "C" int dmnd_mask_block(dmnd_ctx* c, int which, int8_t* host_data, int64_t* n_masked)
{
	return mask_impl(c, which, host_data, nullptr, 0, n_masked);
}

extern "C" int dmnd_mask_sequences(dmnd_ctx* c, int which, int8_t* host_data, const int32_t* seq_ids, int64_t n, int64_t* n_masked)
{
	if (n < 0 || (n > 0 && !seq_ids)) return fail(DMND_E_ARG, "dmnd_mask_sequences: bad argument");
	if (n_masked) *n_masked = 0;
	if (n == 0) return c && (which == DMND_QUERY || which == DMND_TARGET) ? DMND_OK : fail(DMND_E_ARG, "dmnd_mask_sequences: bad argument");
	return mask_impl(c, which, host_data, seq_ids, n, n_masked);
}

static int mask_impl(dmnd_ctx* c, int which, int8_t* host_data, const int32_t* ids, int64_t n_ids, int64_t* n_masked)
{
	if (!c || (which != DMND_QUERY && which != DMND_TARGET)) return fail(DMND_E_ARG, "dmnd_mask_block: bad argument");
	c->soft_valid[which] = false;
	if (which == DMND_QUERY) ++c->query_generation;
	if (!c->block[which].p || c->limits[which].size() < 2) return fail(DMND_E_ARG, "dmnd_mask_block: block must be uploaded with limits");
	if (!c->block[which].own) return fail(DMND_E_ARG, "dmnd_mask_block: the block is shared from another context (dmnd_share_block); mask it there");
	HIP_TRY(hipSetDevice(c->device));
	hipStream_t st = c->stream;
	const std::vector<int64_t>& lim = c->limits[which];
	const int64_t n = (int64_t)lim.size() - 1, raw = c->block_len[which];
	TraceLaps tr(ids ? "dmnd_mask_sequences" : "dmnd_mask_block");
	std::vector<float> lr(32 * 32);
	likelihood_ratios(c->params.matrix8, lr.data());
	tr.lap("likelihood ratios");
	TantanArgs a;
	// Masking::operator() (masking.cpp:176): p_repeat 0.005, p_repeat_end 0.05, growth 1/0.9, minMaskProb 0.9 (config.cpp:402)
	const float p_repeat = 0.005f, p_repeat_end = 0.05f, growth = 1.0f / 0.9f;
	a.p.p_repeat_end = p_repeat_end;
	a.p.b2b = 1.0f - p_repeat;
	a.p.f2f = 1.0f - p_repeat_end;
	a.p.p_mask = (float)0.9;
	const float b2f0 = p_repeat * (1.0f - growth) / (1.0f - std::pow(growth, (float)TANTAN_WINDOW));     // tantan.cpp:130-136
	a.p.d[TANTAN_WINDOW - 1] = b2f0;
	for (int i = TANTAN_WINDOW - 2; i >= 0; --i) a.p.d[i] = a.p.d[i + 1] * growth;
	if (int rc = c->mask_lr.ensure(lr.size() * sizeof(float))) return rc;
	// a subset of the sequences (lazy masking: only the targets that reach the extension stage): compact scratch, ids and scratch
	// offsets uploaded
	std::vector<int64_t> soff;
	int64_t scratch_letters = raw, n_work = n;
	if (ids) {
		soff.resize((size_t)n_ids);
		int64_t acc = 0;
		for (int64_t k = 0; k < n_ids; ++k) {
			if (ids[k] < 0 || ids[k] >= n) return fail(DMND_E_ARG, "dmnd_mask_sequences: sequence id outside the block");
			soff[(size_t)k] = acc;
			acc += lim[(size_t)ids[k] + 1] - lim[(size_t)ids[k]];
		}
		scratch_letters = acc + 64; n_work = n_ids;
		if (int rc = c->mask_ids.ensure((size_t)n_ids * sizeof(int32_t))) return rc;
		if (int rc = c->mask_soff.ensure((size_t)n_ids * sizeof(int64_t))) return rc;
		HIP_TRY(hipMemcpyAsync(c->mask_ids.p, ids, (size_t)n_ids * sizeof(int32_t), hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemcpyAsync(c->mask_soff.p, soff.data(), (size_t)n_ids * sizeof(int64_t), hipMemcpyHostToDevice, st));
	}
	// DMND_TANTAN_WAVE=1: the round-2 kernel (one wavefront per sequence, one lane per repeat offset) instead of one lane per sequence
	static const bool wave_kernel = [] { const char* e = std::getenv("DMND_TANTAN_WAVE"); return e && e[0] == '1'; }();
	// The lane-per-sequence kernel runs a wavefront's 64 sequences in lock step: 2 len steps of ~0.5 - 0.9 us each for the
	// longest of them, whatever else the device has to do -- 6 ms for a 3.0e8-letter block, but 6 ms too for a block of 10 000
	// queries with a 7 000-letter one among them, and 70 ms for a titin. The wavefront-per-sequence kernel is 7 times slower per
	// letter on a full device and 3 times faster on a single sequence. So: sequences longer than `long_len` -- the length whose
	// lock-step time (0.9 us per step on a full device) equals what the lanes kernel needs for the block's letters anyway (20 ps per
	// letter), and never below 1024 letters (measured on 51 000 targets of 15e6 letters, the lazy masking of C2's stock run: all
	// in the lanes kernel 1.2 ms, those above 256 letters in the wavefront kernel 3.0 ms) -- go to the wavefront kernel (their ids
	// and scratch offsets listed here), the lanes kernel leaves them out (its length keys of them are 0).
	int64_t max_len = 0, total_len = 0, long_len = 0, n_long = 0, long_letters = 0;
	std::vector<int32_t> long_ids;
	std::vector<int64_t> long_soff;
	auto work_id = [&](int64_t k) { return ids ? (int64_t)ids[k] : k; };
	if (!wave_kernel) {
		if (ids) for (int64_t k = 0; k < n_ids; ++k) total_len += lim[(size_t)ids[k] + 1] - lim[(size_t)ids[k]] - 1;
		else total_len = raw;
		static const int64_t long_env = [] { const char* e = std::getenv("DMND_TANTAN_LONG"); return e ? std::atoll(e) : (long long)0; }();
		long_len = long_env > 0 ? long_env : std::max<int64_t>(1024, total_len / 45000);
		for (int64_t k = 0; k < n_work; ++k) {
			const int64_t id = work_id(k), l = lim[(size_t)id + 1] - lim[(size_t)id] - 1;
			if (l > long_len) { long_ids.push_back((int32_t)id); long_soff.push_back(long_letters); long_letters += l + 1; }
			else max_len = std::max(max_len, l);
		}
		n_long = (int64_t)long_ids.size();
	}
	const int64_t lanes_floats = wave_kernel ? 0 : tantan_lanes_scratch(n_work, total_len - (long_letters - n_long), max_len), n_waves = (n_work + 63) / 64;
	if (n_long > 0) {
		if (int rc = c->mask_long_ids.ensure((size_t)n_long * sizeof(int32_t))) return rc;
		if (int rc = c->mask_long_soff.ensure((size_t)n_long * sizeof(int64_t))) return rc;
		if (int rc = c->mask_long_pb.ensure((size_t)(long_letters + 64) * sizeof(float))) return rc;
		if (int rc = c->mask_long_scale.ensure((size_t)(long_letters / 16 + n_long + 16) * sizeof(float))) return rc;
		HIP_TRY(hipMemcpyAsync(c->mask_long_ids.p, long_ids.data(), (size_t)n_long * sizeof(int32_t), hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemcpyAsync(c->mask_long_soff.p, long_soff.data(), (size_t)n_long * sizeof(int64_t), hipMemcpyHostToDevice, st));
	}
	if (wave_kernel) {
		if (int rc = c->mask_pb.ensure((size_t)scratch_letters * sizeof(float))) return rc;
		if (int rc = c->mask_scale.ensure((size_t)(scratch_letters / 16 + n_work + 16) * sizeof(float))) return rc;
	}
	else {
		if (n_work >= ((int64_t)1 << 32)) return fail(DMND_E_ARG, "dmnd_mask_block: more than 2^32 sequences");
		if (int rc = c->mask_pb.ensure((size_t)lanes_floats * sizeof(float))) return rc;
		if (int rc = c->sort_keys[0].ensure((size_t)n_work * sizeof(uint32_t))) return rc;
		if (int rc = c->sort_keys[1].ensure((size_t)n_work * sizeof(uint32_t))) return rc;
		if (int rc = c->sort_idx[0].ensure((size_t)n_work * sizeof(uint32_t))) return rc;
		if (int rc = c->mask_scale.ensure((size_t)(2 * (n_waves + 1) + 1) * sizeof(int64_t))) return rc;
	}
	if (int rc = c->counters.ensure(64 * sizeof(unsigned long long))) return rc;
	tr.lap("lengths, buffers");
	HIP_TRY(hipMemcpyAsync(c->mask_lr.p, lr.data(), lr.size() * sizeof(float), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemsetAsync(c->counters.p, 0, 2 * sizeof(unsigned long long), st));
	// the host copy is patched from a list of the masked positions (a few per mille of the letters) instead of copying the
	// whole block back over PCIe; a block with more than 1/16 of its letters masked takes the full copy
	const unsigned long long pos_cap = host_data && raw < ((int64_t)1 << 32) ? (unsigned long long)raw / 16 + 1024 : 0;
	if (pos_cap) if (int rc = c->mask_pos.ensure((size_t)pos_cap * sizeof(uint32_t))) return rc;
	a.data = c->block[which].as<int8_t>();
	a.limits = c->d_limits[which].as<int64_t>();
	a.n_seqs = n_work;
	a.ids = ids ? c->mask_ids.as<int32_t>() : nullptr;
	a.scratch_off = ids ? c->mask_soff.as<int64_t>() : nullptr;
	a.lr = c->mask_lr.as<float>();
	a.pb = c->mask_pb.as<float>();
	a.scale = c->mask_scale.as<float>();
	a.n_masked = c->counters.as<unsigned long long>();
	a.masked_pos = pos_cap ? c->mask_pos.as<uint32_t>() : nullptr;
	a.n_pos = c->counters.as<unsigned long long>() + 1;
	a.pos_cap = pos_cap;
	TantanLanesArgs la;
	if (!wave_kernel) {
		la.t = a;
		la.keys[0] = c->sort_keys[0].as<uint32_t>(); la.keys[1] = c->sort_keys[1].as<uint32_t>(); la.order = c->sort_idx[0].as<uint32_t>();
		la.wave_off = c->mask_scale.as<int64_t>(); la.scratch = c->mask_pb.as<float>(); la.scratch_floats = lanes_floats;
		la.sort_tmp = &c->sort_tmp; la.sort_tmp_bytes = &c->sort_tmp_bytes;
		la.long_len = long_len;
		HIP_TRY(prepare_tantan_lanes(la, st));
	}
	tr.lap("work space");
	HIP_TRY(hipEventRecord(c->ev0, st));
	if (!wave_kernel) {
		if (n_long < n_work) HIP_TRY(launch_tantan_lanes(la, st));
		tr.lap("lanes chain enqueued");
		if (n_long > 0) {
			TantanArgs w = a;
			w.n_seqs = n_long;
			w.ids = c->mask_long_ids.as<int32_t>(); w.scratch_off = c->mask_long_soff.as<int64_t>();
			w.pb = c->mask_long_pb.as<float>(); w.scale = c->mask_long_scale.as<float>();
			HIP_TRY(launch_tantan(w, st));
		}
	}
	else
	HIP_TRY(launch_tantan(a, st));
	HIP_TRY(hipEventRecord(c->ev1, st));
	tr.lap("all enqueued");
	// (through the context's page-locked chunk: the runtime's own path to a pageable destination took 6 ms here for these 16 bytes)
	unsigned long long cnt[2] = { 0, 0 };
	if (int rc = download_bytes(c, cnt, c->counters.p, sizeof(cnt))) return rc;
	tr.lap("kernels done, counters read");
	const unsigned long long nm = cnt[0];
	if (host_data && pos_cap && cnt[1] <= pos_cap) {
		std::vector<uint32_t> pos((size_t)cnt[1]);
		if (cnt[1]) if (int rc = download_bytes(c, pos.data(), c->mask_pos.p, pos.size() * sizeof(uint32_t))) return rc;
		for (uint32_t x : pos) host_data[x] = 23;
	}
	else if (host_data) HIP_TRY(copy_now(st, host_data, c->block[which].p, (size_t)raw, hipMemcpyDeviceToHost));
	tr.lap("host copy patched");
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
	c->mask_ms = ms;
	if (tr.on) std::fprintf(stderr, "%s: %lld sequences (%lld long, > %lld letters), kernels %.3f ms between the events\n", tr.call, (long long)n_work, (long long)n_long, (long long)long_len, (double)ms);
	if (n_masked) *n_masked = (int64_t)nm;
	// The scratch (4.25 B per letter: 1.3 GB for a 3.0e8-letter block) stays with the context for the next block. Rounds 3-4 gave it
	// back above 256 MB: the hipMalloc / hipFree pair per call (hipFree = a device-wide wait plus page-table work) was 12.9 of the
	// 19 ms a block's masking took in the pipelined step, twice the kernel's own 6.1 ms. 288 GB of HBM hold it easily; the bound
	// (DMND_MASK_SCRATCH_KEEP_MB, default 16 GiB) only exists so that one pathological block does not pin its scratch for good.
	static const size_t keep_bytes = [] { const char* e = std::getenv("DMND_MASK_SCRATCH_KEEP_MB"); return (size_t)(e ? std::max(0ll, std::atoll(e)) : 16384ll) << 20; }();
	for (DevBuf* b : { &c->mask_pb, &c->mask_scale, &c->mask_pos, &c->mask_long_pb, &c->mask_long_scale })
		if (b->cap > keep_bytes) b->release();
	tr.lap("scratch kept");
	return DMND_OK;
}

extern "C" double dmnd_mask_kernel_ms(const dmnd_ctx* c) { return c ? c->mask_ms : 0.0; }
