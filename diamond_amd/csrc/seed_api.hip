// seed_api.hip -- host side of the seed stage (include/diamond_hip.h: dmnd_seed_search / dmnd_seed_hits).
// Orchestrates seed_kernels.hip per shape: index queries -> stream the reference -> complexity masks (all shapes
// first, because a pair's left-most test needs the mask times of every earlier shape/chunk), then the pair filter
// per shape. Replaces the control flow of Search::search_shape (/root/reference/src/search/stage0.cpp:101-217).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <vector>
#include <chrono>
#include "ctx.h"
#include "seed_core.h"
#include "seed_kernels.h"

using namespace dmnd;

enum { SEED_CLASSES_LONG_DEFAULT = 1 };      // C5 (100 000 queries, 16 MB level-1 filter): stream 3.0 -> 2.06 ms per 1.9e8-letter block, seed stage 33.2 -> 25.4 ms per 8 blocks (tools/gpu_r05b.sh)

static_assert(sizeof(dmnd_seed_params) == sizeof(SeedParams), "dmnd_seed_params must mirror dmnd::SeedParams");
static_assert(sizeof(dmnd_seed_hit) == 24, "dmnd_seed_hit layout");

extern "C" int dmnd_seed_params_fast(dmnd_seed_params* p, int threads)
{
	if (!p || threads < 1) return fail(DMND_E_ARG, "dmnd_seed_params_fast: bad argument");
	std::memset(p, 0, sizeof(*p));
	const char* code = "1101110101101111";                 // shape_codes[FAST], search/setup.cpp:211-212
	p->n_shapes = 1;
	int w = 0, len = (int)std::strlen(code);
	for (int i = 0; i < len; ++i)
		if (code[i] == '1') { p->shape_pos[0][w++] = (int8_t)i; p->shape_mask[0] |= 1u << i; }
	p->shape_len[0] = len; p->shape_weight[0] = w;
	// murphy10 "A KR EDNQ C G H ILVM FYW P ST" over ARNDCQEGHILKMFPSTWYV (stats/stats.cpp:48, basic/value.h:53);
	// X and '*' reduce to the mask letter, every other code to class 0 (Reduction ctor, basic.cpp:267-297)
	static const int8_t murphy10[32] = { 0, 1, 2, 2, 3, 2, 2, 4, 5, 6, 6, 1, 6, 7, 8, 9, 9, 7, 7, 6, 0, 0, 0, 23, 23, 0, 0, 0, 0, 0, 0, 0 };
	std::memcpy(p->reduction, murphy10, 32);
	p->reduction_size = 10;
	p->index_chunks = 4;                                    // sensitivity_traits[FAST].index_chunks
	p->hamming_filter_id = 11;
	// seedp_bits(): max(bit_length(size^weight - 1) - 32, bit_length(threads*4*chunks - 1), 8), setup.cpp:306-309
	auto bit_length = [](uint64_t x) { int b = 0; while (x) { ++b; x >>= 1; } return b; };
	uint64_t space = 1;
	for (int i = 0; i < w; ++i) space *= 10;
	p->seedp_bits = std::max(std::max(bit_length(space - 1) - 32, bit_length((uint64_t)threads * 4 * p->index_chunks - 1)), 8);
	p->ungapped_window = 48;
	p->left_most_interval = 32;
	p->seed_complexity_cut = 0.9 * 0.69314718055994530942 * w;      // seed_cut * log(2) * weight, setup.cpp:369-370
	p->use_ungapped = 0; p->short_query_max_len = 60; p->tile_size = 1024; p->simd_lanes = 32;
	return DMND_OK;
}

namespace {

// shapes + the stage-2 ungapped e-value filter (10000) shared by the default and sensitive presets
int spaced_preset(dmnd_seed_params* p, int threads, const dmnd_params* sc, const char* const* codes, int n, double seed_cut,
	double ungapped_evalue = 10000.0, double ungapped_evalue_short = 10000.0, int hamming_id = 11, int index_chunks = 4)
{
	if (int rc = dmnd_seed_params_fast(p, threads)) return rc;
	p->n_shapes = n;
	p->hamming_filter_id = hamming_id;
	p->index_chunks = index_chunks;
	int weight = 0;
	for (int sid = 0; sid < n; ++sid) {
		int w = 0;
		const int len = (int)std::strlen(codes[sid]);
		p->shape_mask[sid] = 0;
		std::memset(p->shape_pos[sid], 0, sizeof(p->shape_pos[sid]));
		for (int i = 0; i < len; ++i)
			if (codes[sid][i] == '1') { p->shape_pos[sid][w++] = (int8_t)i; p->shape_mask[sid] |= 1u << i; }
		p->shape_len[sid] = len; p->shape_weight[sid] = w;
		weight = w;
	}
	auto bit_length = [](uint64_t x) { int b = 0; while (x) { ++b; x >>= 1; } return b; };
	uint64_t space = 1;
	for (int i = 0; i < weight; ++i) space *= (uint64_t)p->reduction_size;
	p->seedp_bits = std::max(std::max(bit_length(space - 1) - 32, bit_length((uint64_t)threads * 4 * p->index_chunks - 1)), 8);      // setup.cpp:306-309
	p->seed_complexity_cut = seed_cut * 0.69314718055994530942 * weight;       // setup.cpp:369-370
	// ungapped e-value 10000: CutoffTable + short-query cutoff (cutoff_table.h:30-35, score_matrix.h:130-151, config.cpp:431)
	const double LN2 = 0.69314718055994530941723212145818, ln_k = std::log(sc->K);
	auto raw = [&](double bits) { return (int32_t)std::ceil((bits * LN2 + ln_k) / sc->lambda); };
	p->use_ungapped = 1;
	p->short_query_cutoff = raw(25.0);
	p->cutoff_table[0] = 0;
	p->cutoff_table_short[0] = 0;
	for (int b = 1; b < 32; ++b) {
		p->cutoff_table[b] = raw(-std::log(ungapped_evalue / 1e9 / (double)(1u << (b - 1))) / std::log(2.0));
		p->cutoff_table_short[b] = raw(-std::log(ungapped_evalue_short / 1e9 / (double)(1u << (b - 1))) / std::log(2.0));
	}
	return DMND_OK;
}

}

extern "C" int dmnd_seed_params_default(dmnd_seed_params* p, int threads, const dmnd_params* sc)
{
	if (!p || !sc || threads < 1) return fail(DMND_E_ARG, "dmnd_seed_params_default: bad argument");
	static const char* const codes[2] = { "111101110111", "111011010010111" };          // shape_codes[DEFAULT], search/setup.cpp:82-84
	return spaced_preset(p, threads, sc, codes, 2, 0.8);
}

extern "C" int dmnd_seed_params_sensitive(dmnd_seed_params* p, int threads, const dmnd_params* sc)
{
	if (!p || !sc || threads < 1) return fail(DMND_E_ARG, "dmnd_seed_params_sensitive: bad argument");
	static const char* const codes[16] = {                                               // shape_codes[SENSITIVE], search/setup.cpp:86-102
		"1011110111", "110100100010111", "11001011111", "101110001111", "11011101100001", "1111010010101", "111001001001011",
		"10101001101011", "111101010011", "1111000010000111", "1100011011011", "1101010000011011", "1110001010101001",
		"110011000110011", "11011010001101", "1101001100010011" };
	return spaced_preset(p, threads, sc, codes, 16, 1.0);
}

// One entry for every sensitivity this library presets: shape set, seed cut and the gapped-filter e-value that goes with it
// (sensitivity_traits + shape_codes, search/setup.cpp:40-53,80-215)
extern "C" int dmnd_seed_params_preset(dmnd_seed_params* p, int sensitivity, int threads, const dmnd_params* sc, double* gapped_filter_evalue)
{
	if (!p || threads < 1) return fail(DMND_E_ARG, "dmnd_seed_params_preset: bad argument");
	if (gapped_filter_evalue) *gapped_filter_evalue = 0.0;
	switch (sensitivity) {
	case DMND_SENS_FAST: return dmnd_seed_params_fast(p, threads);
	case DMND_SENS_DEFAULT: return dmnd_seed_params_default(p, threads, sc);
	case DMND_SENS_MID_SENSITIVE: {
		if (!sc) return fail(DMND_E_ARG, "dmnd_seed_params_preset: scoring parameters needed");
		static const char* const codes[8] = { "11110110111", "1101100111101", "1110010101111", "11010101100111", "11101110001011",
			"1110100100010111", "1101000011010111", "1110011000011011" };                   // 8x9, setup.cpp:196-205
		return spaced_preset(p, threads, sc, codes, 8, 1.0);
	}
	case DMND_SENS_SENSITIVE:
	case DMND_SENS_MORE_SENSITIVE:       // same shapes and filters; differs only in freq_sd / motif masking, which this library does not apply
		if (gapped_filter_evalue) *gapped_filter_evalue = 1.0;
		return dmnd_seed_params_sensitive(p, threads, sc);
	case DMND_SENS_VERY_SENSITIVE: {
		if (!sc) return fail(DMND_E_ARG, "dmnd_seed_params_preset: scoring parameters needed");
		static const char* const codes[14] = { "11101111", "110110111", "111111001", "1010111011", "11110001011", "110100101011", "110110001101",
			"1010101000111", "1100101001011", "1101010101001", "1110010010011", "110110000010011", "111001000100011", "1101000100010011" };   // 14x7, setup.cpp:118-133
		if (gapped_filter_evalue) *gapped_filter_evalue = 1.0;
		// sensitivity_traits: min id 9, ungapped e-values 100000 / 30000 (short), one index chunk (setup.cpp:51)
		return spaced_preset(p, threads, sc, codes, 14, 1.0, 100000.0, 30000.0, 9, 1);
	}
	case DMND_SENS_ULTRA_SENSITIVE: {
		if (!sc) return fail(DMND_E_ARG, "dmnd_seed_params_preset: scoring parameters needed");
		static const char* const codes[64] = { "1111111", "11101111", "110011111", "110110111", "111111001", "1010111011", "1011110101", "1111000111",
			"10011110011", "10101101101", "10111010101", "11001010111", "11001100111", "11010101101", "11110001011", "100111010011", "101100110101",
			"101110000111", "110100101011", "110110001101", "111000110011", "1010001011011", "1010101000111", "1010110100011", "1100100110011",
			"1100101001011", "1101001100101", "1101010101001", "1110001010101", "1110010010011", "10100001101101", "11000100010111", "11010000100111",
			"11010100110001", "11101000011001", "11110000001101", "11110100000011", "101001000001111", "110000100101011", "110010010000111",
			"110101100001001", "110110000010011", "111001000100011", "111100000100101", "1000110010010101", "1001000100101101", "1001000110011001",
			"1010001001001011", "1010001010010011", "1010010001010101", "1010010100010011", "1010010101001001", "1010100000101011", "1010100011000101",
			"1011000010001011", "1100010000111001", "1100010010001011", "1100100001001011", "1100100100100011", "1100110000001101", "1101000100010011",
			"1101000110000101", "1110000001010011", "1110100000010101" };                                   // 64x7, setup.cpp:135-200
		if (gapped_filter_evalue) *gapped_filter_evalue = 1.0;
		// sensitivity_traits: min id 9, ungapped e-values 300000 / 30000 (short), one index chunk (setup.cpp:53)
		return spaced_preset(p, threads, sc, codes, 64, 1.0, 300000.0, 30000.0, 9, 1);
	}
	default:
		return fail(DMND_E_ARG, "dmnd_seed_params_preset: unknown sensitivity");
	}
}

extern "C" int dmnd_seed_params_set_index_chunks(dmnd_seed_params* p, int index_chunks, int threads)
{
	if (!p || index_chunks < 1 || threads < 1 || p->n_shapes < 1) return fail(DMND_E_ARG, "dmnd_seed_params_set_index_chunks: bad argument");
	auto bit_length = [](uint64_t x) { int b = 0; while (x) { ++b; x >>= 1; } return b; };
	uint64_t space = 1;
	for (int i = 0; i < p->shape_weight[0]; ++i) space *= (uint64_t)p->reduction_size;
	p->index_chunks = index_chunks;
	p->seedp_bits = std::max(std::max(bit_length(space - 1) - 32, bit_length((uint64_t)threads * 4 * index_chunks - 1)), 8);      // setup.cpp:306-309
	return DMND_OK;
}

extern "C" int dmnd_seed_params_set_query_indexed(dmnd_seed_params* p, int threads)
{
	if (!p || threads < 1 || p->n_shapes < 1) return fail(DMND_E_ARG, "dmnd_seed_params_set_query_indexed: bad argument");
	if (p->reduction_size > 16) return fail(DMND_E_ARG, "dmnd_seed_params_set_query_indexed: the hashed seed encoding packs 4 bits per letter");
	for (int i = 0; i < p->n_shapes; ++i)
		if (p->shape_len[i] > 16) return fail(DMND_E_ARG, "dmnd_seed_params_set_query_indexed: shapes longer than 16 letters");
	p->seed_encoding = SEED_HASHED;
	return dmnd_seed_params_set_index_chunks(p, 1, threads);       // run/double_indexed.cpp:297: one index chunk unless -c is given
}

extern "C" int dmnd_auto_query_indexed(const dmnd_seed_params* params, const int8_t* qdata, const int64_t* qlimits, int64_t nq, int64_t db_bytes, int* query_indexed)
{
	if (!params || !qdata || !qlimits || nq < 0 || !query_indexed) return fail(DMND_E_ARG, "dmnd_auto_query_indexed: bad argument");
	*query_indexed = 0;
	const int64_t MiB = (int64_t)1 << 20;
	const int64_t letters = nq > 0 ? qlimits[nq] - qlimits[0] - nq : 0;
	if (letters > 32 * MiB || db_bytes < 256 * MiB) return DMND_OK;                    // MAX_INDEX_QUERY_SIZE, MIN_QUERY_INDEXED_DB_SIZE
	if (params->reduction_size > 16) return DMND_OK;
	for (int i = 0; i < params->n_shapes; ++i) if (params->shape_len[i] > 16) return DMND_OK;
	auto next_pow2 = [](double x) { uint64_t n = (uint64_t)std::ceil(x), p = 1; while (p < n) p <<= 1; return p; };
	// a shape has at most one seed per letter: only query blocks close to the 32 Mi limit need their seeds counted
	if (next_pow2((double)letters * 1.25) <= (uint64_t)(32 * MiB)) { *query_indexed = 1; return DMND_OK; }
	SeedParams sp;
	std::memcpy(&sp, params, sizeof(sp));
	sp.seed_encoding = SEED_HASHED;
	// (3e7 keys per shape for a block at the limit: sorted on one thread that was 2.3 s per shape, more than the whole search of
	// such a block takes on the device. Eight threads: each counts the seeds of its eighth of the sequences per key class, then
	// writes them into ONE array grouped by class -- huge pages asked for: 240 MB of first-touched 4 KB pages cost 0.9 s under
	// eight faulting threads --, then each sorts and counts one class in place)
	uint64_t largest = 0;
	constexpr int T = 8;
	const size_t cap = (size_t)letters + 16;
	const size_t bytes = ((cap * sizeof(uint64_t) + ((size_t)2 << 20) - 1) >> 21) << 21;
	uint64_t* keys = static_cast<uint64_t*>(std::aligned_alloc((size_t)2 << 20, bytes));
	if (!keys) return fail(DMND_E_NOMEM, "dmnd_auto_query_indexed: out of memory");
	(void)madvise(keys, bytes, MADV_HUGEPAGE);
	auto key_class = [](uint64_t k) { return (int)((k * 0x9E3779B97F4A7C15ull) >> 61); };
	for (int sid = 0; sid < sp.n_shapes; ++sid) {
		size_t count[T][T] = { { 0 } }, at[T][T];             // [producer][class]
		uint64_t distinct[T] = { 0 };
		auto for_seeds = [&](int t, auto&& f) {
			for (int64_t i = nq * t / T; i < nq * (t + 1) / T; ++i)
				for (int64_t p = qlimits[i]; p + sp.shape_len[sid] < qlimits[i + 1]; ++p) {
					uint64_t k;
					if (seed_key_hashed(sp, sid, qdata + p, k)) f(k);
				}
		};
		auto team = [&](auto&& body) {
			std::vector<std::thread> th;
			for (int t = 0; t < T; ++t) th.emplace_back([&, t] { body(t); });
			for (std::thread& x : th) x.join();
		};
		team([&](int t) { for_seeds(t, [&](uint64_t k) { ++count[t][key_class(k)]; }); });
		size_t class_begin[T + 1], off = 0;
		for (int c = 0; c < T; ++c) { class_begin[c] = off; for (int t = 0; t < T; ++t) { at[t][c] = off; off += count[t][c]; } }
		class_begin[T] = off;
		team([&](int t) { for_seeds(t, [&](uint64_t k) { keys[at[t][key_class(k)]++] = k; }); });
		team([&](int c) {
			std::sort(keys + class_begin[c], keys + class_begin[c + 1]);
			distinct[c] = (uint64_t)(std::unique(keys + class_begin[c], keys + class_begin[c + 1]) - (keys + class_begin[c]));
		});
		uint64_t all = 0;
		for (int c = 0; c < T; ++c) all += distinct[c];
		largest = std::max(largest, next_pow2((double)all * 1.25));
		if (largest > (uint64_t)(32 * MiB)) break;          // the answer is no whatever the other shapes say
	}
	std::free(keys);
	*query_indexed = largest <= (uint64_t)(32 * MiB) ? 1 : 0;
	return DMND_OK;
}

extern "C" int dmnd_set_query_index_reuse(dmnd_ctx* c, int on)
{
	if (!c) return fail(DMND_E_ARG, "dmnd_set_query_index_reuse: ctx is NULL");
	c->reuse_query_index = on != 0;
	c->qindex_signature.clear();
	return DMND_OK;
}

extern "C" int dmnd_seed_kernel_ms(const dmnd_ctx* c, double ms[5])
{
	if (!c || !ms) return fail(DMND_E_ARG, "dmnd_seed_kernel_ms: bad argument");
	for (int i = 0; i < 5; ++i) ms[i] = c->seed_ms[i];
	return DMND_OK;
}

namespace {

// Device time of the phases of a search (dmnd_seed_kernel_ms), WITHOUT a host wait per phase: start / stop only record events on
// the stream, collect() reads every pair once the search has been waited for anyway. (Until round 5 stop() waited for its event:
// four host round trips per shape that served nothing but the clock -- an interrupt-driven wake-up and an idle device each,
// ~0.13 ms of the 2.46 ms a C2 seed stage takes.)
struct Timer {
	hipStream_t st;
	hipEvent_t open = nullptr;
	struct Span { hipEvent_t a, b; double* into; };
	std::vector<Span> spans;
	Timer(hipStream_t s) : st(s) {}
	~Timer() { if (open) (void)hipEventDestroy(open); for (const Span& x : spans) { (void)hipEventDestroy(x.a); (void)hipEventDestroy(x.b); } }
	void start()
	{
		if (!open) (void)hipEventCreate(&open);
		(void)hipEventRecord(open, st);
	}
	void stop(double& into)
	{
		if (!open) return;
		hipEvent_t b = nullptr;
		(void)hipEventCreateWithFlags(&b, spin_sync() ? hipEventDefault : hipEventBlockingSync);
		(void)hipEventRecord(b, st);
		spans.push_back(Span{ open, b, &into });
		open = nullptr;
		if (waits()) collect();
	}
	// DMND_SEED_TIMER_WAITS=1: the clock of rounds 1-4 (a host wait behind every phase), for A/B runs
	static bool waits() { static const bool v = [] { const char* e = std::getenv("DMND_SEED_TIMER_WAITS"); return e && e[0] == '1'; }(); return v; }
	// a phase that runs again after an overflow: the spans of the abandoned attempt do not count
	void discard(const double* into)
	{
		size_t k = 0;
		for (const Span& x : spans)
			if (x.into == into) { (void)hipEventDestroy(x.a); (void)hipEventDestroy(x.b); }
			else spans[k++] = x;
		spans.resize(k);
	}
	void collect()
	{
		if (!spans.empty()) (void)wait_event(spans.back().b);
		for (const Span& x : spans) {
			float ms = 0;
			if (hipEventElapsedTime(&ms, x.a, x.b) == hipSuccess) *x.into += ms;
			(void)hipEventDestroy(x.a); (void)hipEventDestroy(x.b);
		}
		spans.clear();
	}
};

}

// Buffer geometry of a seed search: a function of the seed parameters and the number of query positions only, so that the buffers
// can be reserved (dmnd_seed_reserve) before the blocks are resident
struct SeedSizes {
	uint64_t slots, bm_words, bm1_words;
	uint32_t bm1_k3;
	int stream_nt, probe_policy, SB, slot_shift;
	bool fused, reuse;
	size_t bm_total;
};

static SeedSizes seed_sizes(const dmnd_ctx* c, const SeedParams& sp, int64_t nq_pos)
{
	SeedSizes z;
	const int S = sp.n_shapes;
	// Table slots per query position: two for long seeds, FOUR for the short seeds of the fused pipeline, where a third of the
	// reference positions probe the table and the probe chains are what costs (most query seeds are distinct: two slots per
	// position = 36 % load). Measured on C3 (16 shapes): stream + filter 189 / 124 / 116 / 115 ms at 1 / 2 / 4 / 8 slots per position
	// (DMND_SEED_SLOTS_X8 = 8 / 16 / 32 / 64, slots per position in eighths).
	const uint64_t slots_x8 = (uint64_t)(tuning().seed_slots_x8 ? tuning().seed_slots_x8 : (seed_stream_can_fuse(sp) ? 32 : 16));
	z.slots = 1024;
	while (z.slots * 8 < (uint64_t)nq_pos * slots_x8) z.slots <<= 1;
	// slot numbers are 32-bit in the position lists and the joined-position records (LIST_END = all ones): a query block above 2^30
	// positions keeps the table at 2^31 slots, i.e. fewer slots per position (the query-block limit of 4 G letters is checked by the caller)
	while (z.slots > ((uint64_t)1 << 31) && z.slots / 2 >= (uint64_t)nq_pos + (uint64_t)nq_pos / 4) z.slots >>= 1;

	// bitmap: >= 16 bits per query position, at most 2^27 bits (16 MB); word mask for 32-bit words
	uint64_t bm_words = 1 << 15;
	while (bm_words * 32 < (uint64_t)nq_pos * 16 && bm_words < ((uint64_t)1 << 22)) bm_words <<= 1;
	// level-1 bitmap: 2^24 bits = 2 MB by default (half an XCD's L2), never larger than level 2
	// ... unless the query block is so large that 2 MB saturate (above 16 M positions more than 85 % of the bits are set and nearly
	// every probe is positive): then 4 bits per position, up to 2^27 -- a filter that works from the Infinity Cache beats one in
	// L2 that does not filter (100k queries: stream 34 -> 24 ms per 8 blocks). Below that the L2-resident size wins even at 70 %
	// density (blastx, 10 M positions: 3.7 against 5.8 ms).
	int bm1_log2 = 24;
	if (nq_pos > ((int64_t)1 << 24))
		while (bm1_log2 < 27 && ((uint64_t)1 << bm1_log2) < (uint64_t)nq_pos * 4) ++bm1_log2;
	if (tuning().seed_bitmap1_log2) bm1_log2 = tuning().seed_bitmap1_log2;      // word index = 22 bits of hash a
	uint64_t bm1_words = ((uint64_t)1 << bm1_log2) / 32;
	if (bm1_words > bm_words) bm1_words = bm_words;
	uint32_t bm1_k3 = 0;
	int stream_nt = 0;
	// Long seeds (level-2 path: nearly every level-1 positive is a false one) with a query block that the default size serves:
	// 3 MB and three bits per seed. Measured on C2 (3e6 query seeds, tools/stream_sweep.py): the stream kernel takes the same
	// 1.45 ms as with 2 MB / two bits -- its bound is the rate at which the L2s serve 4-byte probes, not the misses behind
	// them -- but the false positives drop from 9.7 % to 3.1 %, and with them the level-2 / table lines fetched over the fabric
	// (4 MB: 1.53 ms, 8 MB: 2.56 ms -- the filter has to fit an XCD's 4 MB L2 beside the stream)
	const bool long_seeds = !seed_stream_can_fuse(sp);
	if (long_seeds && bm1_log2 == 24 && bm_words >= 3 * bm1_words / 2) {
		bm1_words = 3 * bm1_words / 2;
		bm1_k3 = (uint64_t)nq_pos * 4 <= bm1_words * 32 ? 1u : 0u;      // a third bit pays from ~4.3 filter bits per seed up (optimum k = ln 2 x bits per seed)
	}
	// Short seeds by class (round 5 sweep on C3, tools/gpu_r05b.sh): 4 MB / three bits -- half a megabyte per XCD -- instead of 2 MB / two:
	// stream + filter 103.9 -> 102.6 ms per 16 shapes (8 MB: 103.4; fewer false positives = fewer slot lines fetched over the fabric)
	if (!long_seeds && bm1_log2 == 24 && !tuning().seed_bitmap1_log2 && bm_words >= 2 * bm1_words) { bm1_words *= 2; bm1_k3 = 1u; }
	if (tuning().seed_bm1_kb) bm1_words = (uint64_t)tuning().seed_bm1_kb * 256;       // (tuning.h: the sweeps' overrides)
	if (tuning().seed_bm1_k) bm1_k3 = tuning().seed_bm1_k == 3 ? 1u : 0u;
	if (tuning().seed_stream_nt >= 0) stream_nt = tuning().seed_stream_nt != 0;
	int probe_policy = 0;
	if (tuning().seed_probe_policy >= 0) probe_policy = tuning().seed_probe_policy;
	// The fused pipeline finishes a shape before it starts the next one: its table, lists and bitmaps are ONE shape's, reused
	// (64 shapes of --ultra-sensitive would otherwise hold 8 GB of tables for a 10k-query block)
	bool fused = seed_stream_can_fuse(sp);
	if (const char* e = getenv("DMND_SEED_FUSED")) fused = fused && atoi(e) != 0;
	// Query index reuse (dmnd_set_query_index_reuse; a query block against many reference blocks): the tables, lists and bitmaps of
	// ALL shapes stay resident between calls and a call that finds them built for the same query block and parameters only resets
	// the per-block state (join / erase marks) instead of rebuilding them. Needs S buffer sets; refused above 64 GiB.
	z.slot_shift = 4;
	const size_t set_bytes = (z.slots << z.slot_shift) + 2 * (size_t)nq_pos * sizeof(uint32_t) + (size_t)(bm_words + bm1_words) * sizeof(uint32_t);
	const bool reuse = c->reuse_query_index && set_bytes * (size_t)S <= ((size_t)64 << 30) && !getenv("DMND_SEED_MATCHED_CAP");
	const int SB = (fused && !reuse) ? 1 : S;            // shapes that own buffers at the same time
	const size_t bm_total = (size_t)SB * (bm_words + bm1_words) * sizeof(uint32_t);
	z.bm_words = bm_words; z.bm1_words = bm1_words; z.bm1_k3 = bm1_k3; z.stream_nt = stream_nt; z.probe_policy = probe_policy;
	z.fused = fused; z.reuse = reuse; z.SB = SB; z.bm_total = bm_total;
	return z;
}

// The buffers a search of this geometry needs before its first kernel. generous: also the buffers whose size depends on what the
// search finds, at their starting capacities (dmnd_seed_reserve; the search itself grows them where they overflow)
static int seed_reserve(dmnd_ctx* c, const SeedParams& sp, const SeedSizes& z, int64_t nq_pos, int64_t q_end, int64_t q_block_len, bool generous)
{
	if (int rc = c->seed_bitmap.ensure(z.bm_total)) return rc;
	if (int rc = c->qid_of.ensure((size_t)q_end * sizeof(uint32_t))) return rc;
	if (int rc = c->mask_time.ensure((size_t)q_block_len + 256)) return rc;
	if (int rc = c->seed_keys.ensure((size_t)z.SB * (z.slots << z.slot_shift))) return rc;
	if (int rc = c->seed_need.ensure((size_t)(z.slots / 32 + SEED_NEED_FOLD_WORDS) * sizeof(uint32_t))) return rc;      // the map and, behind it, its folded copy (launch_seed_collect)
	if (int rc = c->seed_next.ensure((size_t)z.SB * nq_pos * sizeof(uint32_t))) return rc;        // qslot
	if (int rc = c->seed_qlist.ensure((size_t)z.SB * nq_pos * sizeof(uint32_t))) return rc;
	if (int rc = c->seed_qkeys.ensure((size_t)nq_pos * sizeof(uint32_t))) return rc;
	// (c->counters is shared with the masking calls that may run beside a reservation: the search allocates it itself)
	if (!generous) return DMND_OK;
	const int64_t m_cap = std::max<int64_t>((int64_t)1 << 22, 4 * nq_pos);
	if (int rc = c->matched_slot.ensure((size_t)m_cap * sizeof(uint32_t))) return rc;
	if (int rc = c->matched_loc.ensure((size_t)m_cap * sizeof(int64_t))) return rc;
	if (int rc = c->seed_hits.ensure(((size_t)1 << 20) * sizeof(dmnd_seed_hit))) return rc;
	if (int rc = c->seed_deferred.ensure(((size_t)1 << 18) * sizeof(SeedDeferred))) return rc;
	if (z.fused) {
		if (int rc = c->seed_survivors.ensure(((size_t)1 << 20) * sizeof(SeedSurvivor))) return rc;
		if (int rc = c->seed_qfold.ensure((size_t)(q_block_len + 1) / 2 + 64)) return rc;
	}
	return DMND_OK;
}

extern "C" int dmnd_seed_reserve(dmnd_ctx* c, const dmnd_seed_params* params, int64_t query_block_len)
{
	if (!c || !params || query_block_len <= 512) return fail(DMND_E_ARG, "dmnd_seed_reserve: bad argument");
	SeedParams sp;
	std::memcpy(&sp, params, sizeof(sp));
	if (sp.n_shapes < 1 || sp.n_shapes > SEED_MAX_SHAPES) return fail(DMND_E_ARG, "dmnd_seed_reserve: unsupported seed configuration");
	HIP_TRY(hipSetDevice(c->device));
	const int64_t nq_pos = query_block_len - 512;             // a SequenceSet block: 256 padding letters on either side
	return seed_reserve(c, sp, seed_sizes(c, sp, nq_pos), nq_pos, query_block_len - 256, query_block_len, true);
}

extern "C" int dmnd_seed_search(dmnd_ctx* c, const dmnd_seed_params* params, int64_t* n_hits)
{
	if (!c || !params || !n_hits) return fail(DMND_E_ARG, "dmnd_seed_search: NULL argument");
	// DMND_TRACE: host wall time of the call's parts (the kernel times do not show allocations, copies and waits)
	const bool lap_on = getenv("DMND_TRACE") != nullptr;
	const auto lap_t0 = std::chrono::steady_clock::now();
	auto lap = [&](const char* what) {
		if (lap_on) std::fprintf(stderr, "dmnd_seed_search %8.2f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - lap_t0).count(), what);
	};
	*n_hits = 0;
	c->n_seed_hits = 0;                                  // a failed or empty search must not leave the previous search's hits behind
	const std::vector<int64_t>& ql = c->limits[DMND_QUERY];
	const std::vector<int64_t>& tl = c->limits[DMND_TARGET];
	if (ql.size() < 2 || tl.size() < 2) return fail(DMND_E_ARG, "dmnd_seed_search: both blocks must be uploaded with limits");
	SeedParams sp;
	std::memcpy(&sp, params, sizeof(sp));
	if (sp.n_shapes < 1 || sp.n_shapes > SEED_MAX_SHAPES || sp.index_chunks < 1 || sp.seedp_bits < 1 || sp.seedp_bits > 24
		|| sp.n_shapes * sp.index_chunks >= SEED_NEVER || sp.ungapped_window < 1 || sp.ungapped_window > 128 || sp.reduction_size < 2)
		return fail(DMND_E_ARG, "dmnd_seed_search: unsupported seed configuration");
	// seed_is_complex counts reduced letters in count[20] and indexes LNFACT[20] by the counts (<= weight): weight <= 19, classes <= 20
	if (sp.reduction_size > 20) return fail(DMND_E_ARG, "dmnd_seed_search: more than 20 reduced letter classes");
	for (int l = 0; l < 32; ++l)
		if (!((sp.reduction[l] >= 0 && sp.reduction[l] < sp.reduction_size) || sp.reduction[l] == L_MASK))
			return fail(DMND_E_ARG, "dmnd_seed_search: reduction map entry outside [0, reduction_size) and not the mask letter");
	for (int i = 0; i < sp.n_shapes; ++i) {
		if (sp.shape_len[i] < 1 || sp.shape_len[i] > 32 || sp.shape_weight[i] < 1 || sp.shape_weight[i] > 19 || sp.shape_weight[i] > sp.shape_len[i])
			return fail(DMND_E_ARG, "dmnd_seed_search: bad shape (length 1..32, weight 1..19)");
		uint32_t mask = 0;
		for (int k = 0; k < sp.shape_weight[i]; ++k) {
			if (sp.shape_pos[i][k] < 0 || sp.shape_pos[i][k] >= sp.shape_len[i]) return fail(DMND_E_ARG, "dmnd_seed_search: shape position outside the shape");
			mask |= 1u << sp.shape_pos[i][k];
		}
		if (mask != sp.shape_mask[i]) return fail(DMND_E_ARG, "dmnd_seed_search: shape mask and positions disagree");
	}
	if (sp.seed_encoding != SEED_SPACED && sp.seed_encoding != SEED_HASHED) return fail(DMND_E_ARG, "dmnd_seed_search: unknown seed encoding");
	if (sp.seed_encoding == SEED_HASHED) {
		if (sp.index_chunks != 1 || sp.reduction_size > 16) return fail(DMND_E_ARG, "dmnd_seed_search: the query-indexed mode runs with one index chunk and a 4-bit reduction");
		for (int i = 0; i < sp.n_shapes; ++i)
			if (sp.shape_len[i] > 16) return fail(DMND_E_ARG, "dmnd_seed_search: query-indexed mode with a shape longer than 16 letters");
	}
	HIP_TRY(hipSetDevice(c->device));
	hipStream_t st = c->stream;
	const int64_t q_begin = ql.front(), q_end = ql.back(), t_begin = tl.front(), t_end = tl.back();
	const int64_t nq_pos = q_end - q_begin;
	if (nq_pos <= 0 || t_end <= t_begin) return DMND_OK;
	if (nq_pos >= 0xffffffffLL) return fail(DMND_E_ARG, "dmnd_seed_search: query block larger than 4G letters");
	const int S = sp.n_shapes;
	const SeedSizes z = seed_sizes(c, sp, nq_pos);
	if (z.slots > ((uint64_t)1 << 31)) return fail(DMND_E_ARG, "dmnd_seed_search: query block too large for 32-bit slot numbers (more than 1.7 G seed positions): cut it into smaller blocks");
	const uint64_t slots = z.slots, bm_words = z.bm_words, bm1_words = z.bm1_words;
	const uint32_t bm1_k3 = z.bm1_k3;
	const int stream_nt = z.stream_nt, probe_policy = z.probe_policy, SB = z.SB;
	const size_t slot_bytes = (size_t)slots << z.slot_shift;      // one shape's table
	const bool fused = z.fused, reuse = z.reuse;
	const size_t bm_total = z.bm_total;
	int list_key_bits = 1;                                // of the list sort's keys: log2(slots) + 1 (launch_seed_lists)
	while (((uint64_t)1 << list_key_bits) <= slots) ++list_key_bits;
	// key classes (seed_core.h seed_class): the short-seed pipeline, when the geometry allows eighths (DMND_SEED_CLASSES=0: one range)
	static const bool classes_env = [] { const char* e = getenv("DMND_SEED_CLASSES"); return !e || atoi(e) != 0; }();
	// ... and (round 5) long seeds against a query block whose level-1 filter has outgrown an XCD's L2 (above 2^24 query positions the
	// filter is 4-16 MB: C5's 100 000 queries): by class every XCD probes its own eighth of it. DMND_SEED_CLASSES_LONG=0/1 forces it.
	const int classes_long_env = [] { const char* e = getenv("DMND_SEED_CLASSES_LONG"); return e ? atoi(e) : -1; }();      // (read per call: the tests switch it)
	bool nibble_shapes = true;
	for (int i = 0; i < S; ++i) nibble_shapes = nibble_shapes && seed_nibble_mode(sp, i);
	const bool classes_long = !fused && nibble_shapes && (classes_long_env >= 0 ? classes_long_env != 0 : SEED_CLASSES_LONG_DEFAULT && bm1_words * 32 >= ((uint64_t)1 << 25));      // (C2's 3 MB filter stays on the plain stream: by class it took 2.57 ms against 1.46)
	const int classes = (fused ? classes_env : classes_long) && slots >= 64 && bm1_words % 8 == 0 && bm1_words >= 64 ? 8 : 0;
	std::string signature;
	if (reuse) {
		signature.assign(reinterpret_cast<const char*>(&sp), sizeof(sp));
		const uint64_t extra[7] = { c->query_generation, (uint64_t)nq_pos, slots, bm_words, bm1_words, (uint64_t)fused + 2 * (uint64_t)classes, (uint64_t)z.slot_shift * 2 + bm1_k3 };
		signature.append(reinterpret_cast<const char*>(extra), sizeof(extra));
	}
	const bool index_ready = reuse && c->qindex_signature == signature && !signature.empty();
	c->qindex_signature.clear();                         // set again when this call has built (or kept) a complete index
	lap("parameters checked");
	if (int rc = seed_reserve(c, sp, z, nq_pos, q_end, c->block_len[DMND_QUERY], false)) return rc;
	lap("buffers ensured");
	static const bool phases = getenv("DMND_SEED_PHASES") != nullptr;
	const bool counters_new = c->counters.cap < (size_t)(S + 16) * sizeof(unsigned long long);
	if (int rc = c->counters.ensure((size_t)(S + 16) * sizeof(unsigned long long))) return rc;      // [S] hits, [S+1] deferred pairs, [S+2] collected positions, [S+3] Hamming survivors, [S+4] scored survivors
	// everything a search starts from, in ONE launch (launch_seed_clear): the counters, the mask times, the need map and -- unless the
	// query side is kept from the last call -- bitmaps, slots-of-positions and table
	{
		SeedClear z;
		z.add(c->counters.p, (size_t)((phases || counters_new) ? S + 16 : S + 5) * sizeof(unsigned long long), 0);
		z.add(c->mask_time.p, (size_t)c->block_len[DMND_QUERY] + 256, SEED_NEVER);
		z.add(c->seed_need.p, (size_t)(slots / 32) * sizeof(uint32_t), 0);
		if (!index_ready) {
			z.add(c->seed_bitmap.p, bm_total, 0);
			z.add(c->seed_next.p, (size_t)SB * nq_pos * sizeof(uint32_t), 0xff);
			z.add(c->seed_keys.p, (size_t)SB * slot_bytes, 0xff);
		}
		HIP_TRY(launch_seed_clear(z, st));
	}
	bool pristine = true;                                // the counters and the need map are as the clear above left them
	HIP_TRY(launch_seed_qid(c->d_limits[DMND_QUERY].as<int64_t>(), (int64_t)ql.size() - 1, c->qid_of.as<uint32_t>(), st));

	// fused pipeline: 4-bit copy of the query block for the Hamming pre-filter (DMND_SEED_FOLD=0 switches it off)
	static const bool fold_env = [] { const char* e = getenv("DMND_SEED_FOLD"); return !e || atoi(e) != 0; }();
	const bool use_fold = fused && fold_env;
	if (use_fold) {
		const int64_t n = c->block_len[DMND_QUERY];
		if (int rc = c->seed_qfold.ensure((size_t)(n + 1) / 2 + 64)) return rc;
		HIP_TRY(launch_seed_fold(c->block[DMND_QUERY].as<int8_t>(), n, c->seed_qfold.as<uint8_t>(), st));
	}
	// ... and of the reference block for the by-class stream (round 5): eight workgroups on eight XCDs read the letters around the
	// joins of every tile, each through its own L2 -- from the folded copy that is half the lines (DMND_SEED_TFOLD=0: from the letters)
	static const bool tfold_env = [] { const char* e = getenv("DMND_SEED_TFOLD"); return !e || atoi(e) != 0; }();
	const bool use_tfold = use_fold && classes && tfold_env;
	if (use_tfold) {
		const int64_t n = c->block_len[DMND_TARGET];
		if (int rc = c->seed_tfold.ensure((size_t)(n + 1) / 2 + 64)) return rc;
		HIP_TRY(launch_seed_fold(c->block[DMND_TARGET].as<int8_t>(), n, c->seed_tfold.as<uint8_t>(), st));
	}
	if (classes) {
		const int8_t* tseed = (c->soft_valid[DMND_TARGET] && sp.seed_encoding == SEED_SPACED) ? c->soft[DMND_TARGET].as<int8_t>() : c->block[DMND_TARGET].as<int8_t>();
		const int64_t n = seed_code_groups(t_begin, t_end);
		if (int rc = c->seed_tcodes.ensure((size_t)n * sizeof(uint64_t))) return rc;
		if (int rc = c->seed_tflags.ensure((size_t)n * sizeof(uint32_t))) return rc;
		if (int rc = c->seed_tplanes.ensure((size_t)n * sizeof(uint64_t))) return rc;
		if (int rc = c->seed_tclass.ensure((size_t)9 * (size_t)((n + 3) & ~(int64_t)3) * sizeof(uint16_t))) return rc;      // planes of a stride that is a multiple of four groups: 8-byte stores
		HIP_TRY(launch_seed_codes(sp, tseed, t_begin, t_end, c->seed_tcodes.as<uint64_t>(), c->seed_tflags.as<uint32_t>(), c->seed_tplanes.as<uint64_t>(), st));
	}
	const int level2_env = [] { const char* e = getenv("DMND_SEED_LEVEL2"); return e ? atoi(e) : -1; }();
	auto level2_of = [&](int sid) { return level2_env >= 0 ? level2_env : (sp.shape_weight[sid] >= 10 ? 1 : 0); };
	auto args_for = [&](int sid, int64_t matched_cap, int64_t matched_off) {
		const int own = sid % SB;                          // which of the SB buffer sets the shape uses
		SeedArgs a;
		a.params = sp;
		a.qdata = c->block[DMND_QUERY].as<int8_t>(); a.tdata = c->block[DMND_TARGET].as<int8_t>();
		// motif soft masking: seeds come from the masked views; the query-indexed algorithm does not soft-mask the reference
		// block (search/stage0.cpp:125-127)
		a.qseed = c->soft_valid[DMND_QUERY] ? c->soft[DMND_QUERY].as<int8_t>() : a.qdata;
		a.tseed = (c->soft_valid[DMND_TARGET] && sp.seed_encoding == SEED_SPACED) ? c->soft[DMND_TARGET].as<int8_t>() : a.tdata;
		a.qlimits = c->d_limits[DMND_QUERY].as<int64_t>();
		a.q_begin = q_begin; a.q_end = q_end; a.t_begin = t_begin; a.t_end = t_end;
		a.qid_of = c->qid_of.as<uint32_t>(); a.mask_time = c->mask_time.as<uint8_t>();
		a.slots = reinterpret_cast<SeedSlot*>(c->seed_keys.as<char>() + (size_t)own * slot_bytes);
		a.slot_shift = z.slot_shift;
		a.qslot = c->seed_next.as<uint32_t>() + (size_t)own * nq_pos;
		a.qlist = c->seed_qlist.as<uint32_t>() + (size_t)own * nq_pos;
		a.slot_mask = slots - 1;
		a.classes = classes;
		a.phase_ticks = phases ? c->counters.as<unsigned long long>() + S + 8 : nullptr;
		a.tclass = classes ? c->seed_tclass.as<uint16_t>() : nullptr; a.tclass_stride = (seed_code_groups(t_begin, t_end) + 3) & ~(int64_t)3;
		a.tcodes = classes ? c->seed_tcodes.as<uint64_t>() : nullptr; a.tflags = classes ? c->seed_tflags.as<uint32_t>() : nullptr; a.tplanes = classes ? c->seed_tplanes.as<uint64_t>() : nullptr;
		a.bitmap = c->seed_bitmap.as<uint32_t>() + (size_t)own * (bm_words + bm1_words);
		a.bitmap_mask = (uint32_t)(bm_words - 1);
		a.bitmap1 = a.bitmap + bm_words;
		a.bitmap1_words = (uint32_t)bm1_words; a.bitmap1_k3 = bm1_k3; a.stream_nt = stream_nt; a.probe_policy = probe_policy;
		a.matched_slot = c->matched_slot.as<uint32_t>() + matched_off;
		a.matched_loc = c->matched_loc.as<int64_t>() + matched_off;
		a.matched_count = c->counters.as<unsigned long long>() + sid;
		a.matched_cap = matched_cap;
		a.deferred = nullptr; a.deferred_count = c->counters.as<unsigned long long>() + S + 1; a.deferred_cap = 0;
		a.survivors = nullptr; a.survivor_count = c->counters.as<unsigned long long>() + S + 3; a.survivor_cap = 0;
		a.scored = nullptr; a.scored_count = c->counters.as<unsigned long long>() + S + 4;
		a.need_bits = c->seed_need.as<uint32_t>();
		a.e_key = nullptr; a.e_count = c->counters.as<unsigned long long>() + S + 2; a.e_n = 0;
		a.matrix = c->matrix.as<int8_t>();
		a.hits = c->seed_hits.as<dmnd_seed_hit>(); a.hit_count = c->counters.as<unsigned long long>() + S; a.hit_cap = 0;
		a.fused = fused ? 1 : 0;
		a.level2 = level2_of(sid);
		a.qfold = use_fold ? c->seed_qfold.as<uint8_t>() : nullptr;
		a.tfold = use_tfold ? c->seed_tfold.as<uint8_t>() : nullptr;
		return a;
	};

	Timer tm(st);
	for (int i = 0; i < 5; ++i) c->seed_ms[i] = 0;
	lap("clears and query ids enqueued");
	// the query side of one shape: built (table, position lists, bitmaps), or -- kept from an earlier call -- its marks reset
	auto query_side = [&](const SeedArgs& a, int sid, bool build) -> int {
		if (build) {
			HIP_TRY(launch_seed_index(a, sid, st));
			HIP_TRY(launch_seed_lists(a, sid, c->seed_qkeys.as<uint32_t>(), c->seed_qlist.as<uint32_t>() + (size_t)(sid % SB) * nq_pos, list_key_bits, &c->sort_tmp, &c->sort_tmp_bytes, st));
		}
		else HIP_TRY(launch_seed_reset(a, sid, st));
		return DMND_OK;
	};
	if (c->soft_valid[DMND_QUERY]) {
		SeedArgs a = args_for(0, 0, 0);
		HIP_TRY(launch_seed_soft_time(a, st));
	}
	std::vector<unsigned long long> counts((size_t)S + 1, 0);
	unsigned long long n_pairs = 0;
	// Short seeds: one pass per shape -- index, stream with the Hamming filter fused in, mask, stage-2 scoring of the survivors,
	// deferred pairs. A shape's masks only depend on this and earlier shapes, and so does the left-most rule (t_now).
	if (fused) {
		unsigned long long* ctr = c->counters.as<unsigned long long>();
		int64_t m_cap = std::max<int64_t>(std::max<int64_t>((int64_t)1 << 22, 4 * nq_pos), (int64_t)(std::min(c->matched_loc.cap / sizeof(int64_t), c->matched_slot.cap / sizeof(uint32_t))));
		if (const char* e = getenv("DMND_SEED_MATCHED_CAP")) m_cap = std::max<int64_t>(1, atoll(e));
		int64_t surv_cap = std::max<int64_t>((int64_t)1 << 20, (int64_t)(c->seed_survivors.cap / sizeof(SeedSurvivor)));
		if (const char* e = getenv("DMND_SEED_SURVIVOR_CAP")) surv_cap = std::max<int64_t>(1, atoll(e));
		int64_t hit_cap = std::max<int64_t>((int64_t)1 << 20, (int64_t)(c->seed_hits.cap / sizeof(dmnd_seed_hit)));
		if (const char* e = getenv("DMND_SEED_HIT_CAP")) hit_cap = std::max<int64_t>(1, atoll(e));
		if (int rc = c->seed_hits.ensure((size_t)hit_cap * sizeof(dmnd_seed_hit))) return rc;
		c->seed_trace.assign((size_t)2 * S, 0);
		int64_t hits_bound = 0;                              // every survivor gives at most one hit
		std::vector<unsigned long long> host_ctr((size_t)S + 4);
		// ---- stage 2: everything behind a shape's Hamming filter (stage-2 scores, left-most rule, deferred pairs), on the same stream
		hipStream_t sb = st;
		Timer tmb(sb);
		unsigned long long hits_seen = 0;                     // the hit counter as stage 2 last read it; fresh: nothing appended since
		bool hits_fresh = false;
		auto stage2 = [&](SeedArgs a, int sid, unsigned long long n, unsigned long long ns) -> int {
			HIP_TRY(hipSetDevice(c->device));
			if (hits_bound + (int64_t)ns > hit_cap) {            // grow, keeping the hits of the earlier shapes
				const int64_t new_cap = hits_bound + (int64_t)ns + (hits_bound + (int64_t)ns) / 2;
				DevBuf nb;
				if (int rc = nb.ensure((size_t)new_cap * sizeof(dmnd_seed_hit))) return rc;
				HIP_TRY(hipMemcpyAsync(nb.p, c->seed_hits.p, (size_t)hit_cap * sizeof(dmnd_seed_hit), hipMemcpyDeviceToDevice, sb));
				HIP_TRY(sync_stream(sb));
				c->seed_hits.release();
				c->seed_hits = nb;
				hit_cap = new_cap;
			}
			hits_bound += (int64_t)ns;
			if (int rc = c->seed_deferred.ensure((size_t)ns * sizeof(SeedDeferred))) return rc;      // deferred pairs <= survivors
			a.hits = c->seed_hits.as<dmnd_seed_hit>(); a.hit_cap = hit_cap;
			a.deferred = c->seed_deferred.as<SeedDeferred>(); a.deferred_cap = (int64_t)ns;
			if (int rc = c->seed_scored.ensure((size_t)ns * sizeof(SeedScored))) return rc;
			a.scored = c->seed_scored.as<SeedScored>();
			tmb.start();
			{
				// this lane's own counters (deferred pairs, collected positions, scored) and the need map
				SeedClear zb;
				zb.add(ctr + S + 1, 2 * sizeof(unsigned long long), 0);
				zb.add(ctr + S + 4, sizeof(unsigned long long), 0);
				zb.add(c->seed_need.p, (size_t)(slots / 32) * sizeof(uint32_t), 0);
				HIP_TRY(launch_seed_clear(zb, sb));
			}
			HIP_TRY(launch_seed_post(a, sid, (int64_t)ns, sb, false));
			tmb.stop(c->seed_ms[3]);
			hits_fresh = false;
			if (!sp.use_ungapped) return DMND_OK;
			// (hits so far, deferred pairs of this shape): neighbours in the counter block, one copy -- and if nothing was deferred, the
			// hit count of the search's last shape is already the final one
			unsigned long long both[2] = { 0, 0 };
			HIP_TRY(copy_now(sb, both, a.hit_count, sizeof(both), hipMemcpyDeviceToHost));
			const unsigned long long nd = both[1];
			c->seed_trace[S + sid] = nd;
			if (nd == 0) { hits_seen = both[0]; hits_fresh = true; return DMND_OK; }
			if (int rc = c->seed_eslot.ensure((size_t)n * sizeof(uint64_t))) return rc;
			if (int rc = c->seed_eloc.ensure((size_t)n * sizeof(uint64_t))) return rc;
			a.e_key = c->seed_eslot.as<uint64_t>();
			tmb.start();
			HIP_TRY(launch_seed_collect(a, (int64_t)n, sb));
			unsigned long long ne = 0;
			HIP_TRY(hipMemcpyAsync(&ne, a.e_count, sizeof(ne), hipMemcpyDeviceToHost, sb));
			HIP_TRY(sync_stream(sb));
			HIP_TRY(sort_keys_u64(c->seed_eslot.as<uint64_t>(), c->seed_eloc.as<uint64_t>(), (int64_t)ne, &c->sort_tmp, &c->sort_tmp_bytes, sb));
			a.e_key = c->seed_eloc.as<uint64_t>();
			a.e_n = (int64_t)ne;
			HIP_TRY(launch_seed_deferred(a, sid, (int64_t)nd, sb));
			tmb.stop(c->seed_ms[3]);
			return DMND_OK;
		};
		const bool recycle = SB < S;                          // a buffer set serves several shapes: cleared before its next one
		for (int sid = 0; sid < S; ++sid) {
			SeedArgs a = args_for(sid, 0, 0);
			tm.start();
			if (sid > 0) {
				// the survivor counter and, where a buffer set is used again, its table, slots-of-positions and bitmaps -- one launch
				SeedClear z;
				z.add(ctr + S + 3, sizeof(unsigned long long), 0);
				if (recycle && sid >= SB) {
					const size_t own = (size_t)(sid % SB);
					z.add(c->seed_keys.as<char>() + own * slot_bytes, slot_bytes, 0xff);
					z.add(c->seed_next.as<char>() + own * (size_t)nq_pos * sizeof(uint32_t), (size_t)nq_pos * sizeof(uint32_t), 0xff);
					z.add(c->seed_bitmap.as<char>() + own * (size_t)(bm_words + bm1_words) * sizeof(uint32_t), (size_t)(bm_words + bm1_words) * sizeof(uint32_t), 0);
				}
				HIP_TRY(launch_seed_clear(z, st));
			}
			if (int rc = query_side(a, sid, !index_ready)) return rc;
			tm.stop(c->seed_ms[0]);
			unsigned long long n = 0, ns = 0;
			for (int attempt = 0;; ++attempt) {
				if (int rc = c->matched_slot.ensure((size_t)m_cap * sizeof(uint32_t))) return rc;
				if (int rc = c->matched_loc.ensure((size_t)m_cap * sizeof(int64_t))) return rc;
				if (int rc = c->seed_survivors.ensure((size_t)surv_cap * sizeof(SeedSurvivor))) return rc;
				a.matched_slot = c->matched_slot.as<uint32_t>(); a.matched_loc = c->matched_loc.as<int64_t>(); a.matched_cap = m_cap;
				a.survivors = c->seed_survivors.as<SeedSurvivor>(); a.survivor_cap = surv_cap;
				if (attempt > 0) {                               // (the first attempt finds them zero: the clears above)
					HIP_TRY(hipMemsetAsync(a.matched_count, 0, sizeof(unsigned long long), st));
					HIP_TRY(hipMemsetAsync(a.survivor_count, 0, sizeof(unsigned long long), st));
				}
				tm.start();
				HIP_TRY(launch_seed_stream(a, sid, st, true));
				tm.stop(c->seed_ms[1]);
				HIP_TRY(copy_now(c->stream, host_ctr.data(), ctr, host_ctr.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
				n = host_ctr[sid]; ns = host_ctr[S + 3];
				if ((int64_t)n <= m_cap && (int64_t)ns <= surv_cap) break;
				if (attempt >= 2) return fail(DMND_E_NOMEM, "dmnd_seed_search: joined-position / survivor buffer overflow");
				if ((int64_t)n > m_cap) m_cap = (int64_t)n + (int64_t)n / 8 + 1024;
				if ((int64_t)ns > surv_cap) surv_cap = (int64_t)ns + (int64_t)ns / 8 + 1024;
			}
			counts[sid] = n;
			c->seed_trace[sid] = ns;
			if (sp.seed_encoding == SEED_SPACED) {
				tm.start();
				HIP_TRY(launch_seed_mask(a, sid, st));
				tm.stop(c->seed_ms[2]);
			}
			if (ns == 0) continue;
			if (int rc = stage2(a, sid, n, ns)) return rc;
		}
		unsigned long long nh = hits_seen;
		if (!hits_fresh) HIP_TRY(copy_now(c->stream, &nh, ctr + S, sizeof(nh), hipMemcpyDeviceToHost));
		if ((int64_t)nh > hit_cap) return fail(DMND_E_NOMEM, "dmnd_seed_search: hit buffer overflow");
		c->n_seed_hits = (int64_t)nh;
		tmb.collect();
	}
	else {
	// phase 1: index + stream + mask, every shape. The joined-position lists of all shapes share one buffer.
	std::vector<int64_t> m_off((size_t)S + 1, 0);
	// start from what earlier calls already grew the buffers to: a repeated search of the same scale never takes the overflow path
	int64_t cap_total = std::max<int64_t>(std::max<int64_t>((int64_t)1 << 22, 4 * nq_pos), (int64_t)std::min(c->matched_loc.cap / sizeof(int64_t), c->matched_slot.cap / sizeof(uint32_t)));
	if (const char* e = getenv("DMND_SEED_MATCHED_CAP")) cap_total = std::max<int64_t>(1, atoll(e));      // tests: force the overflow/retry path
	for (int attempt = 0;; ++attempt) {
		if (int rc = c->matched_slot.ensure((size_t)cap_total * sizeof(uint32_t))) return rc;
		if (int rc = c->matched_loc.ensure((size_t)cap_total * sizeof(int64_t))) return rc;
		if (attempt > 0) {
			HIP_TRY(hipMemsetAsync(c->counters.p, 0, (size_t)(S + 5) * sizeof(unsigned long long), st));
			HIP_TRY(hipMemsetAsync(c->seed_keys.p, 0xff, (size_t)S * slot_bytes, st));
			HIP_TRY(hipMemsetAsync(c->seed_bitmap.p, 0, bm_total, st));
			HIP_TRY(hipMemsetAsync(c->seed_next.p, 0xff, (size_t)S * nq_pos * sizeof(uint32_t), st));
		}
		bool overflow = false;
		int64_t off = 0;
		for (int sid = 0; sid < S; ++sid) {
			m_off[sid] = off;
			// after an overflow the remaining shapes only count (capacity 0): a negative capacity would pass the kernels' unsigned bound test
			SeedArgs a = args_for(sid, std::max<int64_t>(cap_total - off, 0), std::min(off, cap_total));
			tm.start();
			if (int rc = query_side(a, sid, !index_ready || attempt > 0)) return rc;
			tm.stop(c->seed_ms[0]);
			tm.start();
			HIP_TRY(launch_seed_stream(a, sid, st));
			tm.stop(c->seed_ms[1]);
			HIP_TRY(copy_now(c->stream, &counts[sid], a.matched_count, sizeof(unsigned long long), hipMemcpyDeviceToHost));
			if ((int64_t)counts[sid] > cap_total - off) { overflow = true; off += (int64_t)counts[sid]; continue; }
			off += (int64_t)counts[sid];
		}
		m_off[S] = off;
		if (!overflow) break;
		if (attempt >= 2) return fail(DMND_E_NOMEM, "dmnd_seed_search: joined-position buffer overflow");
		if (lap_on) std::fprintf(stderr, "dmnd_seed_search: joined-position buffer overflow, %lld positions against a capacity of %lld: phase 1 runs again\n", (long long)off, (long long)cap_total);
		cap_total = off + off / 8 + 1024;
	}
	// Search::mask_seeds (per joined group) only exists for spaced seeds; the query-indexed mode masked at enumeration
	for (int sid = 0; sid < S && sp.seed_encoding == SEED_SPACED; ++sid) {
		SeedArgs a = args_for(sid, (int64_t)counts[sid], m_off[sid]);
		tm.start();
		HIP_TRY(launch_seed_mask(a, sid, st, (int64_t)counts[sid]));
		tm.stop(c->seed_ms[2]);
	}
	if (getenv("DMND_TRACE")) {
		pristine = false;
		HIP_TRY(hipMemsetAsync(c->counters.as<unsigned long long>() + S + 3, 0, sizeof(unsigned long long), st));
		for (int sid = 0; sid < S; ++sid) {
			SeedArgs a = args_for(sid, (int64_t)counts[sid], m_off[sid]);
			HIP_TRY(launch_seed_count_pairs(a, (int64_t)counts[sid], c->counters.as<unsigned long long>() + S + 3, st));
		}
		HIP_TRY(copy_now(c->stream, &n_pairs, c->counters.as<unsigned long long>() + S + 3, sizeof(n_pairs), hipMemcpyDeviceToHost));
	}
	lap("phase 1 done (index, stream, mask of every shape)");
	// phase 2: pair filter per shape; hit and deferred-pair buffers grow on overflow
	int64_t hit_cap = std::max<int64_t>(std::max<int64_t>((int64_t)1 << 20, m_off[S] / 8), (int64_t)(c->seed_hits.cap / sizeof(dmnd_seed_hit)));
	if (const char* e = getenv("DMND_SEED_HIT_CAP")) hit_cap = std::max<int64_t>(1, atoll(e));
	int64_t def_cap = std::max<int64_t>((int64_t)1 << 18, (int64_t)(c->seed_deferred.cap / sizeof(SeedDeferred)));
	if (const char* e = getenv("DMND_SEED_DEFERRED_CAP")) def_cap = std::max<int64_t>(1, atoll(e));
	for (int attempt = 0;; ++attempt) {
		if (int rc = c->seed_hits.ensure((size_t)hit_cap * sizeof(dmnd_seed_hit))) return rc;
		if (int rc = c->seed_deferred.ensure((size_t)def_cap * sizeof(SeedDeferred))) return rc;
		if (!pristine) HIP_TRY(hipMemsetAsync(c->counters.as<unsigned long long>() + S, 0, sizeof(unsigned long long), st));
		if (attempt > 0) tm.discard(&c->seed_ms[3]);
		bool def_overflow = false;
		unsigned long long hits_seen = 0;                     // the hit counter as last read with a shape's deferred count; fresh: nothing appended since
		bool hits_fresh = false;
		c->seed_trace.assign((size_t)2 * S, 0);
		unsigned long long def_max = 0;
		for (int sid = 0; sid < S; ++sid) {
			SeedArgs a = args_for(sid, (int64_t)counts[sid], m_off[sid]);
			a.hits = c->seed_hits.as<dmnd_seed_hit>();
			a.hit_cap = hit_cap;
			a.deferred = c->seed_deferred.as<SeedDeferred>(); a.deferred_cap = def_cap;
			if (!pristine) {
				SeedClear z;
				z.add(a.deferred_count, 2 * sizeof(unsigned long long), 0);
				z.add(a.need_bits, (size_t)(slots / 32) * sizeof(uint32_t), 0);
				HIP_TRY(launch_seed_clear(z, st));
			}
			tm.start();
			// many joined positions (short seeds): sort them by seed and run the LDS-tiled filter; otherwise one thread per position
			bool tiled = (int64_t)counts[sid] >= ((int64_t)1 << 22);
			if (const char* e = getenv("DMND_SEED_TILED")) tiled = atoi(e) != 0;
			if (lap_on && tiled) std::fprintf(stderr, "dmnd_seed_search: shape %d, %llu joined positions: sorted by seed, tiled pair filter\n", sid, counts[sid]);
			if (tiled) {
				if (int rc = c->seed_slot2.ensure((size_t)counts[sid] * sizeof(uint32_t))) return rc;
				if (int rc = c->seed_loc2.ensure((size_t)counts[sid] * sizeof(int64_t))) return rc;
				int slot_bits = 1;
				while (((uint64_t)1 << slot_bits) < slots) ++slot_bits;
				HIP_TRY(sort_matched_by_slot(a.matched_slot, c->seed_slot2.as<uint32_t>(), a.matched_loc, c->seed_loc2.as<int64_t>(), (int64_t)counts[sid], slot_bits,
					&c->sort_tmp, &c->sort_tmp_bytes, st));
				a.matched_slot = c->seed_slot2.as<uint32_t>(); a.matched_loc = c->seed_loc2.as<int64_t>();      // also for the deferred pass below
			}
			// Hamming filter -> survivor list -> stage-2 kernels
			{
				int64_t surv_cap = std::max<int64_t>(std::max<int64_t>((int64_t)1 << 20, tiled ? 2 * (int64_t)counts[sid] : (int64_t)counts[sid]), (int64_t)(c->seed_survivors.cap / sizeof(SeedSurvivor)));
				if (const char* e = getenv("DMND_SEED_SURVIVOR_CAP")) surv_cap = std::max<int64_t>(1, atoll(e));
				unsigned long long ns = 0;
				for (int pass = 0;; ++pass) {
					if (int rc = c->seed_survivors.ensure((size_t)surv_cap * sizeof(SeedSurvivor))) return rc;
					a.survivors = c->seed_survivors.as<SeedSurvivor>(); a.survivor_cap = surv_cap;
					if (!pristine || pass > 0) HIP_TRY(hipMemsetAsync(a.survivor_count, 0, sizeof(unsigned long long), st));
					if (tiled) HIP_TRY(launch_seed_pairs_tiled(a, sid, (int64_t)counts[sid], st));
					else HIP_TRY(launch_seed_pairs(a, sid, (int64_t)counts[sid], st));
					HIP_TRY(hipMemcpyAsync(&ns, a.survivor_count, sizeof(ns), hipMemcpyDeviceToHost, st));
					HIP_TRY(sync_stream(st));
					if ((int64_t)ns <= surv_cap) break;
					if (pass >= 2) return fail(DMND_E_NOMEM, "dmnd_seed_search: survivor buffer overflow");
					surv_cap = (int64_t)ns + 1024;
				}
				c->seed_trace[sid] = ns;
				if (int rc = c->seed_scored.ensure((size_t)std::max<unsigned long long>(ns, 1) * sizeof(SeedScored))) return rc;
				a.scored = c->seed_scored.as<SeedScored>();
				HIP_TRY(launch_seed_post(a, sid, (int64_t)ns, st, !pristine));
				pristine = false;                                // from here on the counters hold this shape's numbers
			}
			tm.stop(c->seed_ms[3]);
			hits_fresh = false;
			if (!sp.use_ungapped) continue;
			// pairs scoring above 255 (rare): resolve the reference's SIMD-batch saturation rule in a second pass.
			// (hits so far, deferred pairs of this shape) are neighbours in the counter block: one copy, and with nothing deferred
			// behind the last shape the hit count is final
			unsigned long long both[2] = { 0, 0 };
			HIP_TRY(copy_now(c->stream, both, a.hit_count, sizeof(both), hipMemcpyDeviceToHost));
			const unsigned long long nd = both[1];
			c->seed_trace[S + sid] = nd;
			if (nd == 0) { hits_seen = both[0]; hits_fresh = true; continue; }
			def_max = std::max(def_max, nd);
			if ((int64_t)nd > def_cap) { def_overflow = true; continue; }
			if (int rc = c->seed_eslot.ensure((size_t)counts[sid] * sizeof(uint64_t))) return rc;      // unsorted keys
			if (int rc = c->seed_eloc.ensure((size_t)counts[sid] * sizeof(uint64_t))) return rc;       // sorted keys
			a.e_key = c->seed_eslot.as<uint64_t>();
			tm.start();
			HIP_TRY(launch_seed_collect(a, (int64_t)counts[sid], st));
			unsigned long long ne = 0;
			HIP_TRY(hipMemcpyAsync(&ne, a.e_count, sizeof(ne), hipMemcpyDeviceToHost, st));
			HIP_TRY(sync_stream(st));
			HIP_TRY(sort_keys_u64(c->seed_eslot.as<uint64_t>(), c->seed_eloc.as<uint64_t>(), (int64_t)ne, &c->sort_tmp, &c->sort_tmp_bytes, st));
			a.e_key = c->seed_eloc.as<uint64_t>();
			a.e_n = (int64_t)ne;
			HIP_TRY(launch_seed_deferred(a, sid, (int64_t)nd, st));
			tm.stop(c->seed_ms[3]);
		}
		unsigned long long nh = hits_seen;
		if (!hits_fresh) HIP_TRY(copy_now(c->stream, &nh, c->counters.as<unsigned long long>() + S, sizeof(nh), hipMemcpyDeviceToHost));
		if ((int64_t)nh <= hit_cap && !def_overflow) { c->n_seed_hits = (int64_t)nh; break; }
		if (attempt >= 3) return fail(DMND_E_NOMEM, "dmnd_seed_search: hit buffer overflow");
		if ((int64_t)nh > hit_cap) hit_cap = (int64_t)nh + 1024;
		if (def_overflow) def_cap = (int64_t)def_max + 1024;
	}
	}
	lap("filters done");
	if (reuse) c->qindex_signature = signature;          // every shape's query side is complete and resident
	// order the hits by (query, subject, seed_offset, score) on the device: what align_queries needs (hits grouped by query),
	// made deterministic (the append order of the kernels is not)
	if (c->n_seed_hits > 0) {
		const int64_t n = c->n_seed_hits;
		if (n > 0xffffffffLL) return fail(DMND_E_CAP, "dmnd_seed_search: more than 2^32 seed hits in one block pair");
		if (int rc = c->seed_hits_sorted.ensure((size_t)n * sizeof(dmnd_seed_hit))) return rc;
		if (int rc = c->sort_keys[0].ensure((size_t)n * sizeof(uint64_t))) return rc;
		if (int rc = c->sort_keys[1].ensure((size_t)n * sizeof(uint64_t))) return rc;
		if (int rc = c->sort_idx[0].ensure((size_t)n * sizeof(uint32_t))) return rc;
		if (int rc = c->sort_idx[1].ensure((size_t)n * sizeof(uint32_t))) return rc;
		uint64_t* keys[2] = { c->sort_keys[0].as<uint64_t>(), c->sort_keys[1].as<uint64_t>() };
		uint32_t* idx[2] = { c->sort_idx[0].as<uint32_t>(), c->sort_idx[1].as<uint32_t>() };
		auto bits_of = [](uint64_t v) { int b = 1; while (b < 64 && (v >> b) != 0) ++b; return b; };
		const std::vector<int64_t>& qlim = c->limits[DMND_QUERY];
		if (c->max_query_len_generation != c->query_generation || c->max_query_len <= 0) {
			int64_t m = 1;
			for (size_t i = 0; i + 1 < qlim.size(); ++i) m = std::max(m, qlim[i + 1] - qlim[i]);
			c->max_query_len = m; c->max_query_len_generation = c->query_generation;
		}
		HIP_TRY(sort_seed_hits(c->seed_hits.as<dmnd_seed_hit>(), c->seed_hits_sorted.as<dmnd_seed_hit>(), n, keys, idx, &c->sort_tmp, &c->sort_tmp_bytes, st,
			bits_of((uint64_t)std::max<size_t>(qlim.size(), 2) - 1), bits_of((uint64_t)c->block_len[DMND_TARGET]), bits_of((uint64_t)c->max_query_len), !sp.use_ungapped));
		HIP_TRY(sync_stream(st));
	}
	lap("hits sorted");
	if (phases) {
		unsigned long long t[8];
		HIP_TRY(copy_now(c->stream, t, c->counters.as<unsigned long long>() + S + 8, sizeof(t), hipMemcpyDeviceToHost));
		std::fprintf(stderr, "SEED_PHASES (ms of workgroup time, summed): start %.1f | windows %.1f | level 1 %.1f | table %.1f | light lists %.1f | heavy lists %.1f\n",
			t[0] / 1e5, t[1] / 1e5, t[2] / 1e5, t[3] / 1e5, t[4] / 1e5, t[5] / 1e5);
	}
	tm.collect();
	c->seed_ms[4] = c->seed_ms[0] + c->seed_ms[1] + c->seed_ms[2] + c->seed_ms[3];
	if (getenv("DMND_TRACE")) {
		std::fprintf(stderr, "dmnd_seed_search: %d shapes, %lld query positions, joined reference positions per shape:", S, (long long)nq_pos);
		for (int sid = 0; sid < S; ++sid) std::fprintf(stderr, " %llu", counts[sid]);
		std::fprintf(stderr, " | pairs %llu | Hamming survivors:", n_pairs);
		for (int sid = 0; sid < S && (size_t)sid < c->seed_trace.size(); ++sid) std::fprintf(stderr, " %llu", c->seed_trace[sid]);
		std::fprintf(stderr, " | deferred:");
		for (int sid = 0; sid < S && (size_t)(S + sid) < c->seed_trace.size(); ++sid) std::fprintf(stderr, " %llu", c->seed_trace[S + sid]);
		std::fprintf(stderr, " | hits %lld | ms index %.2f stream %.2f mask %.2f pairs %.2f\n", (long long)c->n_seed_hits, c->seed_ms[0], c->seed_ms[1], c->seed_ms[2], c->seed_ms[3]);
	}
	*n_hits = c->n_seed_hits;
	return DMND_OK;
}

extern "C" int dmnd_seed_hits(dmnd_ctx* c, dmnd_seed_hit* out, int64_t cap)
{
	if (!c || (!out && c->n_seed_hits > 0)) return fail(DMND_E_ARG, "dmnd_seed_hits: NULL argument");
	if (cap < c->n_seed_hits) return fail(DMND_E_CAP, "dmnd_seed_hits: buffer too small");
	if (c->n_seed_hits == 0) return DMND_OK;
	HIP_TRY(hipSetDevice(c->device));
	if (int rc = download_bytes(c, out, c->seed_hits_sorted.p, (size_t)c->n_seed_hits * sizeof(dmnd_seed_hit))) return rc;      // sorted by dmnd_seed_search
	return DMND_OK;
}
