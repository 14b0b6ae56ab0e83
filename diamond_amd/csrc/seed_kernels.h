// seed_kernels.h -- launch interface between seed_api.hip and seed_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/diamond_hip.h"
#include "seed_core.h"

namespace dmnd {

static const uint64_t SEED_EMPTY = ~0ull;
static const uint32_t LIST_END = 0xffffffffu;
// head = start of the seed's query-position list in SeedArgs::qlist -- or, for a list of ONE, the query position itself;
// flags = state (low byte) | list size << 8.
// A free slot is all ones (one memset initialises the table); the state of an occupied slot is written explicitly, and no
// kernel bit-tests the state of a free slot.
// LOWC: the seed of the group's first query position is not complex (written with the lists; spaced seeds only) -- if the seed
// joins, seed_mask_kernel turns it into ERASED.
enum : uint32_t { SLOT_JOINED = 1, SLOT_ERASED = 2, SLOT_LOWC = 8 };

// One entry of the open-addressing query seed table: key, head of the list of query positions, join state -- 16 bytes, so
// that the probe of the reference stream and the pair filter touch ONE cache line per seed instead of three arrays.
struct SeedSlot { uint64_t key; uint32_t head; uint32_t flags; };

// A (joined reference position, query position) pair whose stage-2 score exceeds 255: whether it saturates depends on the
// reference's SIMD batch it would have been scored in (simd_batch_size_sorted), resolved in a second pass.
struct SeedDeferred { int64_t sloc; uint32_t slot, x; int32_t score, pad; };

// A (joined reference position sloc of the seed in `slot`, query position x) pair that passed the Hamming filter
struct SeedSurvivor { uint32_t slot, x; int64_t sloc; };

// A survivor whose stage-2 ungapped score passed the cutoff: input of the left-most rule (seed_leftmost_kernel)
struct SeedScored { uint32_t slot, x; int64_t sloc; int32_t score, chunk; };

struct SeedArgs {
	SeedParams params;
	const int8_t* qdata; const int8_t* tdata;     // block letters (HBM)
	const int8_t* qseed; const int8_t* tseed;     // the letters seeds are generated from: the blocks, or their motif-soft-masked views
	const int64_t* qlimits;                       // query block limits (HBM)
	int64_t q_begin, q_end, t_begin, t_end;       // letter ranges [limits[0], limits[n])
	const uint32_t* qid_of;                       // query position -> query id
	uint8_t* mask_time;                           // per query letter: (shape, chunk) time of its SEED_MASK bit
	// per-shape query seed table
	SeedSlot* slots;              // 16-byte slots: key, list start (or the seed's one query position), state | list size
	int slot_shift;               // log2 of the slot stride in bytes (4). Tried and dropped in round 3: 64-byte slots carrying the folded
	                              // 48-letter window of single-position seeds, so that the fused Hamming pre-filter needs no second random
	                              // read -- the 512 MB table no longer fits the 256 MB Infinity Cache and the stream + filter kernel of C3
	                              // went from 124 to 137 ms per 16 shapes (L2 misses 3.15e8 -> 4.05e8 per launch)
	__host__ __device__ SeedSlot& slot(uint64_t i) const { return *reinterpret_cast<SeedSlot*>(reinterpret_cast<char*>(slots) + (i << slot_shift)); }
	uint32_t* qslot;                              // per query position: slot of its seed (LIST_END: no seed); input of the list sort
	const uint32_t* qlist;                        // query positions grouped by slot (SeedSlot::head = start, count in flags >> 8)
	uint64_t slot_mask;
	int classes;                                  // 8: slots and level-1 words are partitioned by seed_class(key) (short-seed pipeline); 0: one range
	// ... and then the reference letters as the stream wants them, made once per search by seed_codes_kernel: per group of 16
	// letters (from t_begin rounded down to 16) their class nibbles, and delimiter / no-class maps (low / high 16 bits)
	const uint64_t* tcodes; const uint32_t* tflags;
	const uint64_t* tplanes;                      // the same class nibbles bit-sliced: bits [16 b, 16 b + 16) = bit b of the group's 16 nibbles (seed_classify_kernel)
	const uint16_t* tclass; int64_t tclass_stride;  // per shape (seed_classify_kernel): plane c = the valid windows of class c, 16 per entry; plane 8 (hashed seeds): the special ones
	unsigned long long* phase_ticks;              // DMND_SEED_PHASES=1: 8 counters (seed_stream_fast_kernel PHASE_MARK), else NULL
	// home slot of a key (hh = seed_hash(key)) and word of its level-1 bits (h = seed_hash_a(key))
	__host__ __device__ uint64_t home(uint64_t hh, uint64_t key) const
	{
		if (!classes) return hh & slot_mask;
		const uint64_t low = slot_mask >> 3;
		return (uint64_t)seed_class(key) * (low + 1) | (hh & low);
	}
	__host__ __device__ uint32_t bm1_index(uint32_t h, uint64_t key) const
	{
		if (!classes) return bm1_word(h, bitmap1_words);
		const uint32_t w8 = bitmap1_words >> 3;
		return seed_class(key) * w8 + bm1_word(h, w8);
	}
	// two one-hash bitmaps of the query seeds: level 1 is sized to stay resident in every XCD's 4 MB L2 (the reference
	// stream probes it once per position), level 2 (>= 16 bits per query seed) filters level-1 false positives before
	// the open-addressing table is touched
	uint32_t* bitmap1; uint32_t bitmap1_words, bitmap1_k3;      // level-1 filter (seed_core.h bm1_word / bm1_bits)
	int probe_policy;                                           // cache policy of the level-1 probes (seed_kernels.hip bm1_probe)
	int stream_nt;                                              // reference letters are loaded non-temporally (they are read once)
	uint32_t* bitmap; uint32_t bitmap_mask;
	// joined reference positions of this shape
	uint32_t* matched_slot; int64_t* matched_loc; unsigned long long* matched_count; int64_t matched_cap;
	SeedDeferred* deferred; unsigned long long* deferred_count; int64_t deferred_cap;
	SeedSurvivor* survivors; unsigned long long* survivor_count; int64_t survivor_cap;
	SeedScored* scored; unsigned long long* scored_count;      // as many entries as survivors are scored (launch_seed_post)
	// need_bits: one bit per slot, set for the seeds that have a deferred pair (1 MB for the 10k-query table: the scan over the
	// joined positions tests it in L2 instead of reading 16-byte slots all over the table); e_key: the joined positions of those
	// seeds, sorted by (slot, position) for the second pass
	uint32_t* need_bits;
	uint64_t* e_key; unsigned long long* e_count; int64_t e_n;     // key = slot << 40 | position
	const int8_t* matrix;                         // 32x32 int8 substitution matrix (HBM) for the stage-2 ungapped window score
	// output
	dmnd_seed_hit* hits; unsigned long long* hit_count; int64_t hit_cap;
	const uint8_t* qfold;                         // fused pipeline: the query block with 4 bits per letter (letter & 15), or NULL: pre-filter of the Hamming test
	const uint8_t* tfold;                         // by-class stream: the reference block folded the same way, or NULL (the window is folded from the letters)
	int level2;                                   // the level-2 bitmap is filled and consulted (long seeds)
	int fused;                                    // short-seed pipeline: seed_lists_kernel decides SLOT_LOWC for every group (the stream needs it)
};

// Up to 8 byte ranges set to a byte value each by ONE kernel launch. A search used to start with six hipMemsetAsync calls and add
// four to eight per shape (table, slots-of-positions, bitmaps, mask times, need map, five counters): every one is its own
// fillBufferAligned launch on the stream, ~5 us of launch floor each even for 8 bytes (profiles/r04_kernel_stats_C2.csv: 21 fills
// per C2 step, a tenth of it).
struct SeedClear {
	void* p[8]; uint64_t bytes[8]; uint32_t value[8];
	int n = 0;
	void add(void* ptr, uint64_t nbytes, uint32_t byte_value) { if (ptr && nbytes && n < 8) { p[n] = ptr; bytes[n] = nbytes; value[n] = byte_value; ++n; } }
};
hipError_t launch_seed_clear(const SeedClear& z, hipStream_t st);
hipError_t launch_seed_qid(const int64_t* limits, int64_t n_seqs, uint32_t* qid_of, hipStream_t st);
// room for the folded need map that seed_collect's workgroups keep in LDS (2^13 words = 32 KB by default, up to 2^15); it lies behind
// SeedArgs::need_bits
enum { SEED_NEED_FOLD_WORDS = 32768 };
hipError_t launch_seed_index(const SeedArgs& a, int sid, hipStream_t st);
// query seed positions whose shape window touches a soft-masked stretch get their mask time (MaskingTable::remove's bit mask)
hipError_t launch_seed_soft_time(const SeedArgs& a, hipStream_t st);
// groups the query positions by slot: stable radix sort of (qslot, position) into (sorted_slot, qlist_out), then the list
// start/size of every occupied slot is written into the table
hipError_t launch_seed_lists(const SeedArgs& a, int sid, uint32_t* sorted_slot, uint32_t* qlist_out, int slot_bits, void** tmp, size_t* tmp_bytes, hipStream_t st);
// fused = true (short seeds, where a third of the reference positions join): the stream kernel also runs the Hamming filter on
// every joined pair while the reference letters are at hand, and fills a.survivors; a.matched_* then only serve the deferred pass
// a table kept from an earlier search of the same query block: clears the per-reference-block marks (joined, erased) of every slot
// and, for hashed seeds, repeats the masking of the non-complex query seeds that the index kernel does at enumeration
hipError_t launch_seed_reset(const SeedArgs& a, int sid, hipStream_t st);
// letters [0, n) of a block folded to 4 bits each, two per byte (out: (n + 1) / 2 bytes)
hipError_t launch_seed_fold(const int8_t* data, int64_t n, uint8_t* out, hipStream_t st);
// class nibbles + flag maps of the reference letters [t_begin & ~15, t_end + 32) (SeedArgs::tcodes / tflags); n_groups entries each
inline int64_t seed_code_groups(int64_t t_begin, int64_t t_end) { return (t_end - (t_begin & ~(int64_t)15) + 15) / 16 + 2; }
hipError_t launch_seed_codes(const SeedParams& c, const int8_t* tseed, int64_t t_begin, int64_t t_end, uint64_t* codes, uint32_t* flags, uint64_t* planes, hipStream_t st);
hipError_t launch_seed_stream(const SeedArgs& a, int sid, hipStream_t st, bool fused = false);
bool seed_stream_can_fuse(const SeedParams& c);
// n_matched >= 0: the number of joined positions in a.matched_* (few of them: the kernel walks that list instead of the table)
hipError_t launch_seed_mask(const SeedArgs& a, int sid, hipStream_t st, int64_t n_matched = -1);
hipError_t launch_seed_pairs(const SeedArgs& a, int sid, int64_t n_matched, hipStream_t st);
hipError_t launch_seed_pairs_tiled(const SeedArgs& a, int sid, int64_t n_matched, hipStream_t st);      // a.matched_* sorted by slot; fills a.survivors
hipError_t launch_seed_count_pairs(const SeedArgs& a, int64_t n_matched, unsigned long long* out, hipStream_t st);
// clear_scored = false: the caller has zeroed a.scored_count on the stream already (launch_seed_clear)
hipError_t launch_seed_post(const SeedArgs& a, int sid, int64_t n_survivors, hipStream_t st, bool clear_scored = true);
hipError_t sort_matched_by_slot(const uint32_t* slot_in, uint32_t* slot_out, const int64_t* loc_in, int64_t* loc_out, int64_t n, int slot_bits,
	void** tmp, size_t* tmp_bytes, hipStream_t st);
hipError_t launch_seed_collect(const SeedArgs& a, int64_t n_matched, hipStream_t st);
hipError_t launch_seed_deferred(const SeedArgs& a, int sid, int64_t n_deferred, hipStream_t st);
// ascending device sort of n 64-bit keys (rocPRIM radix sort); tmp is grown as needed
hipError_t sort_keys_u64(const uint64_t* in, uint64_t* out, int64_t n, void** tmp, size_t* tmp_bytes, hipStream_t st);
// Orders the n seed hits by (query, subject, seed_offset, score) on the device: three stable radix-sort passes over an index
// permutation (keys/idx: two buffers of n uint64 / uint32 each), then one gather into `out`.
hipError_t sort_seed_hits(const dmnd_seed_hit* hits, dmnd_seed_hit* out, int64_t n, uint64_t* keys[2], uint32_t* idx[2],
	void** tmp, size_t* tmp_bytes, hipStream_t st, int query_bits, int subject_bits, int off_bits, bool equal_scores = false);      // bits of the largest query id / subject position / seed offset; equal_scores: every hit carries the same score (no ungapped filter: --fast), the pass over the scores is left out

}  // namespace dmnd
