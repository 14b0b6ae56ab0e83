/* diamond_hip.h -- C ABI of libdiamond_hip.so: the MI355X (gfx950) back end for DIAMOND's
 * seed-and-extend hot path.
 *
 * This is the drop-in boundary.  The reference's own operator seam is its
 * DISPATCH_ARCH switch (/root/reference/src/util/simd/dispatch.h:45-228): the functions below are
 * what one more case of that switch binds.  Every entry point cites the reference interface it
 * replaces.  POD only (plain pointers + sizes), no STL, no torch types.
 *
 * Conventions
 *   - return value: 0 = ok, negative = error (DMND_E_*); dmnd_last_error() gives the text
 *     (the reference throws std::runtime_error across this seam, src/run/main.cpp:211-232).
 *   - a dmnd_ctx owns one HIP device, its streams and all device buffers; calls on one ctx are
 *     serialised by the caller (one ctx per host thread / per GPU); distinct ctxs are independent.
 *   - letters are the reference's Letter codes (int8; 0..19 amino acids, 20-22 BJZ, 23 X/mask,
 *     24 '*', 25 super-hard mask, 31 delimiter; bit 7 = SEED_MASK; src/basic/value.h:53-65).
 *   - there is NO CPU fallback: if no gfx950 device is usable every call fails with DMND_E_DEVICE.
 */
#ifndef DIAMOND_HIP_H
#define DIAMOND_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DMND_ABI_VERSION 13      /* 2: dmnd_match.frame, seed parameters (ungapped filter, translated queries), dmnd_extend_plan(query_contexts); 3: seed_encoding; 4: output formats; 5: DMND_MAX_SHAPES 64; 6: dmnd_host_alloc, dmnd_share_block; 7: dmnd_set_max_hsps, several dmnd_match records per target, global ranking; 8: --comp-based-stats 2..5 (dmnd_cbs_*, dmnd_upload_matrices, dmnd_dp_target::cbs_off <= -2); dmnd_mask_block patches host_data in place (the full copy-back of ABI <= 6 only above 1/16 masked letters); 9: frameshift alignment (dmnd_set_frameshift, dmnd_frameshift_swipe, dmnd_match.read_begin / read_end: the record is 104 bytes), dmnd_set_context_motif_table, dmnd_copy_block; 10: dmnd_join_blocks_range (round 5); 11: dmnd_extend_plan_stats (round 6: device planner); 12: dmnd_extend_device_stats (round 6: culling, round 2 and records on the device); 13: dmnd_extend_reserve, dmnd_extend_records_device, dmnd_join_contexts_device, dmnd_join_ranks_plan */

enum {
	DMND_OK = 0,
	DMND_E_ARG = -1,        /* bad argument */
	DMND_E_DEVICE = -2,     /* no usable gfx950 device / HIP error */
	DMND_E_NOMEM = -3,      /* device or host allocation failed */
	DMND_E_BAND = -4,       /* band wider than DMND_MAX_BAND ("Band size exceeds row counter maximum", banded_swipe.h:204) */
	DMND_E_CAP = -5,        /* caller-provided output arena too small */
	DMND_E_TRACEBACK = -6   /* "Traceback error." (banded_swipe.h:168) */
};

/* Widest band of a work item: 65536 diagonals (the reference's 16-bit RowCounter limit is 65535); DMND_SWIPE_STATS: half of it.
 * Up to 4096 (2048 with statistics) one wavefront sweeps an item, wider bands take up to 16 wavefronts. */
#define DMND_MAX_BAND 65536

/* which sequence block (reference: Search::Config::query / ::target Blocks, src/run/config.h) */
enum { DMND_QUERY = 0, DMND_TARGET = 1 };

/* HspValues bit set, identical to the reference's enum (src/basic/match.h: HspValues) */
enum {
	DMND_HSP_NONE = 0, DMND_HSP_TRANSCRIPT = 1, DMND_HSP_QUERY_START = 1 << 1, DMND_HSP_QUERY_END = 1 << 2,
	DMND_HSP_TARGET_START = 1 << 3, DMND_HSP_TARGET_END = 1 << 4, DMND_HSP_IDENT = 1 << 5, DMND_HSP_LENGTH = 1 << 6,
	DMND_HSP_MISMATCHES = 1 << 7, DMND_HSP_GAP_OPENINGS = 1 << 8,
	DMND_HSP_COORDS = (1 << 1) | (1 << 2) | (1 << 3) | (1 << 4)
};

/* how one target is to be computed; mirrors the Cfg chosen by dispatch_swipe()
 * (src/dp/swipe/swipe_wrapper.cpp:183-217) */
enum {
	DMND_SWIPE_SCORE = 0,      /* score only                         (HspValues::NONE)                 */
	DMND_SWIPE_COORDS = 1,     /* + end coordinates                  (VectorRowCounter)                */
	DMND_SWIPE_TRACEBACK = 2,  /* + start coords, transcript, stats  (bins 0-2, TracebackVectorMatrix) */
	DMND_SWIPE_STATS = 3       /* statistics without traceback: ForwardCell pass + reversed BackwardCell
	                              pass (bins 3-5, recompute_reversed, swipe_wrapper.cpp:364-444)      */
};

typedef struct dmnd_ctx dmnd_ctx;

/* Scoring + statistics parameters: the globals the reference reads across the seam
 * (`score_matrix`, `config`; SURVEY.md 8b "Global state read across the seam"). */
typedef struct {
	int8_t matrix8[32 * 32];   /* ScoreMatrix::matrix8(), src/stats/score_matrix.h:69 (row = letter, 32 columns) */
	int32_t gap_open;          /* ScoreMatrix::gap_open()   (BLOSUM62 default 11) */
	int32_t gap_extend;        /* ScoreMatrix::gap_extend() (default 1) */
	/* Gumbel constants for ScoreMatrix::evalue (src/stats/score_matrix.cpp:43-47,217): gapped row and ungapped row */
	double lambda, K, alpha, alpha_v, sigma, u_alpha, u_alpha_v;
	double db_letters;         /* ScoreMatrix::db_letters() */
	double max_evalue;         /* config.max_evalue (report_cutoff, score_matrix.cpp:234) */
} dmnd_params;

#define DMND_CBS_MATRIX_WITH_BIAS ((int64_t)1 << 40)
/* One banded-DP work item = one DpTarget of the reference (src/dp/dp.h:34-157) together with the
 * query it is aligned against (DP::Params::query / composition_bias, src/dp/dp.h:171-184).
 * Offsets address letters inside the blocks uploaded with dmnd_upload_block(). */
typedef struct {
	int64_t query_off;    /* first query letter, offset into the DMND_QUERY block data */
	int64_t target_off;   /* first target letter, offset into the DMND_TARGET block data */
	int64_t cbs_off;      /* offset of the query's int8 composition bias inside the uploaded bias buffer, or -1 (NoCBS);
	                         <= -2: the item has a composition-adjusted matrix of its own (DpTarget::matrix, dp/dp.h:143): number
	                         -2 - cbs_off of the matrices uploaded with dmnd_upload_matrices, and no bias (swipe.h:43-54); with
	                         DMND_CBS_MATRIX_WITH_BIAS subtracted as well, the item is biased too, by the bias bytes at offset query_off
	                         (the reference's FULL-MATRIX sweep keeps the bias in every channel: full_swipe.h:164) */
	int32_t query_len;
	int32_t target_len;   /* DpTarget::seq.length() (a prefix length in reversed passes) */
	int32_t d_begin;      /* diagonal band [d_begin, d_end), diagonal = i - j */
	int32_t d_end;
} dmnd_dp_target;

/* Result of one work item = the fields of Hsp the swipe fills (src/basic/match.h:45-330,
 * banded_swipe.h:88-183).  Fields a mode does not compute are 0. */
typedef struct {
	int32_t score;
	int32_t q_begin, q_end, s_begin, s_end;     /* query_range / subject_range, end exclusive */
	int32_t length, identities, mismatches, positives, gap_openings, gaps;
	int32_t transcript_len;                     /* PackedOperation bytes, without the terminator */
	int64_t transcript_off;                     /* offset into the transcript arena */
} dmnd_hsp;

/* -- lifecycle ---------------------------------------------------------------------------------- */
int dmnd_abi_version(void);
const char* dmnd_last_error(void);
/* Fills params with the reference's defaults: BLOSUM62, gap open 11 / extend 1
 * (ScoreMatrix ctor, src/stats/score_matrix.cpp:49-72), max_evalue 0.001. */
int dmnd_default_params(dmnd_params* params);
/* Scoring of `--matrix NAME --gapopen O --gapextend E`: one of the reference's standard matrices (blosum45, blosum50, blosum62,
 * blosum80, blosum90, pam30, pam70, pam250; any case), gap penalties -1 = the matrix's defaults. Fills matrix8, the penalties and
 * the Gumbel constants of that pair and leaves db_letters / max_evalue as they are. Replaces ScoreMatrix::ScoreMatrix
 * (src/stats/score_matrix.cpp:49-72) + StandardMatrix::get / constants (src/stats/stats.cpp:59-75), with their errors:
 * DMND_E_ARG "Unknown scoring matrix" / "Gap penalty settings are outside the supported range for this scoring matrix." */
int dmnd_matrix_params(const char* name, int gap_open, int gap_extend, dmnd_params* params);
/* Number of usable gfx950 devices (0 if none / HIP not available): what `--gpus N` is checked against. */
int dmnd_device_count(void);
/* device < 0: use the current HIP device. Fails (NULL) when no gfx950 device is present. */
dmnd_ctx* dmnd_create(int device, const dmnd_params* params);
void dmnd_destroy(dmnd_ctx* ctx);
/* ScoreMatrix::set_db_letters (src/run/double_indexed.cpp:900) */
int dmnd_set_db_letters(dmnd_ctx* ctx, double db_letters);

/* -- data: SequenceSet / Block residency (src/data/string_set.h:27-318, sequence_set.h:25) ------- */
/* Uploads the flat letter store of a block (the reference's data_ vector, padding and 0x1F
 * delimiters included, exactly as laid out in host memory) and keeps it resident in HBM.
 * limits may be NULL when only the DP entry points are used. */
int dmnd_upload_block(dmnd_ctx* ctx, int which, const int8_t* data, int64_t data_len,
	const int64_t* limits, int64_t n_seqs);
/* Optional: starts the HIP runtime and loads this library's kernels onto `device` (-1: device 0). Every entry point does so on
 * first use; a driver calls it on a helper thread to overlap the start-up with its own file I/O. */
int dmnd_init(int device);
/* Page-locked host memory for a block's letters: dmnd_upload_block hands such a buffer to the DMA engine as it is, any other
 * (pageable) source is staged through two page-locked chunks of the context while the previous chunk is in flight. A driver that
 * reads a database block from disk into this memory (the reference's loader: data/sequence_file.cpp:113-150 load_seqs,
 * legacy/dmnd/dmnd.cpp:224-340) uploads it at the PCIe rate. NULL on failure. */
void* dmnd_host_alloc(size_t bytes);
void dmnd_host_free(void* p);
/* Makes block `which` of ctx an alias of the block `src` holds (same device): no copy, no second resident copy. For a driver that
 * keeps several reference blocks resident (one context each) and lets ONE context search them in turn, e.g. to keep that context's
 * query seed index over the blocks (dmnd_set_query_index_reuse). `src` must outlive every use; the alias is read-only
 * (dmnd_mask_block on it is refused) and is dropped by the next dmnd_upload_block / dmnd_share_block of that block. */
int dmnd_share_block(dmnd_ctx* ctx, int which, const dmnd_ctx* src);
/* Overwrites the letters of block `which` of ctx with those of the block of the same shape (same sequence limits) that `src` holds
 * on the same device: a device-to-device copy on ctx's stream. For a driver that keeps a block as loaded in one context and masks
 * a working copy in another (dmnd_mask_block works in place) -- the reference re-reads or keeps unmasked letters the same way when
 * a block is searched by several query blocks (Block::soft_mask / remove_soft_masking, src/data/block/block.cpp:164-177). */
int dmnd_copy_block(dmnd_ctx* ctx, int which, const dmnd_ctx* src);
/* Uploads the per-query Hauser composition-bias vectors (HauserCorrection::int8,
 * src/stats/hauser_correction.cpp:107), concatenated; dmnd_dp_target::cbs_off indexes this buffer. */
int dmnd_upload_cbs(dmnd_ctx* ctx, const int8_t* cbs, int64_t len);

/* Uploads n composition-adjusted scoring matrices (32 x 32 int8 each, [target letter * 32 + query letter], as
 * dmnd_cbs_target_matrix writes them = Stats::TargetMatrix::scores, stats/cbs.h:50-62) for the work items that name one in
 * dmnd_dp_target::cbs_off; replaces what was uploaded before. --comp-based-stats 2..5 only. */
int dmnd_upload_matrices(dmnd_ctx* ctx, const int8_t* matrices, int64_t n);

/* -- banded Smith-Waterman: replaces DP::BandedSwipe::swipe (src/dp/dp.h:287;
 *    dispatcher src/dp/swipe/swipe_wrapper.cpp:446-470,487) -------------------------------------- */
/* Computes n work items in one batched launch (any mix of queries).  mode is DMND_SWIPE_*;
 * hsp_values (DMND_HSP_* bits) selects the cell types of DMND_SWIPE_STATS as dispatch_swipe() does.
 * out[n] receives one dmnd_hsp per item (input order).  transcript is a caller-owned host arena of
 * transcript_cap bytes receiving the packed edit transcripts of mode TRACEBACK (PackedOperation codes,
 * src/basic/packed_transcript.h:30-90), each followed by a 0 terminator; *transcript_used returns the
 * bytes written. With transcript == NULL, TRACEBACK still reports coordinates and statistics from the
 * traceback walk (transcript_len set, transcript_off = -1) but returns no transcript bytes.
 * Unlike the SIMD reference there is no 8/16/32-bit escalation: scores are exact int32. */
int dmnd_banded_swipe(dmnd_ctx* ctx, const dmnd_dp_target* items, int64_t n, int mode, uint32_t hsp_values,
	dmnd_hsp* out, uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used);

/* -- three-frame banded sweep of frameshift alignment (blastx -F): replaces the dispatch point banded_3frame_swipe
 *    (src/dp/dp.h:296, src/dp/swipe/banded_3frame_swipe.cpp:597-647) ------------------------------------------------ */
/* One work item = one DpTarget of a banded_3frame_swipe call together with the query strand of that call. */
typedef struct {
	int64_t frame_off[3];   /* the strand's frames 0, 1, 2 (TranslatedSequence::get_strand): offsets into the DMND_QUERY block */
	int64_t target_off;     /* first target letter, offset into the DMND_TARGET block */
	int32_t frame_len[3];
	int32_t target_len;
	int32_t d_begin, d_end; /* the target's band [d_begin, d_end) in query positions, diagonal = i - j */
	int32_t cols;           /* DpTarget::cols as the caller's constructor left it (the legacy pipeline passes qlen = 0: dp.h:47-52,
	                           legacy/banded_swipe_pipeline.cpp:70-76): only the batching order of the score-only pass reads it */
	int32_t strand;         /* 0 = forward, 1 = reverse */
	int32_t dna_len;        /* length of the read (coordinates of the results) */
	int32_t group;          /* items of one banded_3frame_swipe call (one query strand) carry one id and are consecutive */
} dmnd_fs_target;
/* The Hsp fields a call fills (banded_3frame_swipe.cpp:345-414). Score-only: score, frame (0 / 3), q_begin, q_end, read_begin,
 * read_end (the reference's estimate from the end column), max_col. Traceback: everything; transcript = PackedOperation codes with
 * the frameshift operations (basic/packed_transcript.h:26,81-88), 0-terminated, at transcript_off of the caller's arena. */
typedef struct {
	int32_t score, frame;                        /* frame 0 - 5: strand * 3 + frame of the first aligned query position */
	int32_t q_begin, q_end, s_begin, s_end;      /* query_range (positions in their frames) / subject_range, end exclusive */
	int32_t read_begin, read_end;                /* query_source_range: the alignment's interval of the read */
	int32_t length, identities, mismatches, positives, gap_openings, gaps;
	int32_t transcript_len, max_col;
	int64_t transcript_off;                      /* -1: none */
} dmnd_fs_hsp;
/* Sweeps n items. score_only != 0: the items of a group are ordered and swept `channels` at a time on one band geometry, as the
 * reference's int16 vectors do (16 channels with AVX2, 8 with SSE4.1: its results depend on that width; pass what the reference
 * build you compare with uses); a score of 65535 or more is repeated alone. score_only == 0: every item on its own band, with the
 * walk back. frame_shift = the -F penalty. Results in input order; no e-value cut is applied (the caller's report_cutoff). */
int dmnd_frameshift_swipe(dmnd_ctx* ctx, const dmnd_fs_target* items, int64_t n, int score_only, int frame_shift, int channels,
	dmnd_fs_hsp* out, uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used);

/* Same computation on caller-owned HOST sequences -- the literal shape of the reference call
 * (one query, its DpTargets as pointer+length): stages the letters into HBM, then runs the batch. */
typedef struct {
	const int8_t* seq;     /* DpTarget::seq.data() */
	int32_t len;           /* DpTarget::seq.length() */
	int32_t d_begin, d_end;
	const int8_t* matrix;  /* DpTarget::matrix->scores.data() (26 rows of 32 int8, [target letter][query letter]: Stats::TargetMatrix,
	                          stats/cbs.h:50-62) for a target with a composition-adjusted matrix -- it is then scored without cbs --, else NULL */
} dmnd_host_target;
int dmnd_banded_swipe_host(dmnd_ctx* ctx, const int8_t* query, int32_t query_len, const int8_t* cbs,
	const dmnd_host_target* targets, int64_t n, int mode, uint32_t hsp_values,
	dmnd_hsp* out, uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used);

/* DpTarget::banded_cols (src/dp/dp.h:47-52) and DpTarget::cells (:121-124): the cell count the
 * GCUPS metric is defined on. */
int32_t dmnd_banded_cols(int32_t qlen, int32_t tlen, int32_t d_begin, int32_t d_end);

/* ScoreMatrix::evalue / bitscore (src/stats/score_matrix.cpp:217,250), host double precision. */
double dmnd_evalue(const dmnd_ctx* ctx, int32_t raw_score, uint32_t query_len, uint32_t subject_len);
double dmnd_bitscore(const dmnd_ctx* ctx, double raw_score);
/* context-free forms (pure host arithmetic on a parameter block) */
double dmnd_evalue_p(const dmnd_params* params, int32_t raw_score, uint32_t query_len, uint32_t subject_len);
double dmnd_bitscore_p(const dmnd_params* params, double raw_score);
int dmnd_evalue_batch(const dmnd_params* params, const int32_t* raw_score, const int32_t* query_len,
	const int32_t* subject_len, int64_t n, double* out);

/* -- seed stage: replaces Search::search_shape for all shapes and index chunks of one (query block, reference
 *    block) pair (src/search/search.h:83, src/search/stage0.cpp:101-228) -------------------------------------- */
#define DMND_MAX_SHAPES 64        /* --ultra-sensitive: 64 shapes of weight 7 (search/setup.cpp:135-200) */
#define DMND_MAX_SHAPE_WEIGHT 32
/* The globals the reference reads across the seam: `shapes` (basic/shape_config.h), Reduction::instance
 * (basic/reduction.h), Search::Config::{seedp_bits,index_chunks,hamming_filter_id,seed_complexity_cut}
 * (run/config.h:102-115), config.ungapped_window / left_most_interval (basic/config.cpp:558,582). */
typedef struct {
	int32_t n_shapes;
	int32_t shape_len[DMND_MAX_SHAPES], shape_weight[DMND_MAX_SHAPES];
	uint32_t shape_mask[DMND_MAX_SHAPES];                         /* Shape::mask_ */
	int8_t shape_pos[DMND_MAX_SHAPES][DMND_MAX_SHAPE_WEIGHT];     /* Shape::positions_ */
	int8_t reduction[32];                                         /* Reduction::map_ for letters 0..31 (23 = mask) */
	int32_t reduction_size;
	int32_t seedp_bits, index_chunks, hamming_filter_id;
	int32_t ungapped_window, left_most_interval;
	double seed_complexity_cut;
	/* stage-2 ungapped window filter (src/search/stage2.h:43-63,107-113); use_ungapped = 0 <=> ungapped e-value 0 (--fast) */
	int32_t use_ungapped, short_query_max_len, short_query_cutoff;
	int32_t cutoff_table[32];     /* CutoffTable::data_ (src/util/scores/cutoff_table.h:26-47), index = bit length of the query length */
	int32_t tile_size, simd_lanes; /* config.tile_size (1024); int8 lanes of the reference's SIMD build (32 = AVX2): the batch rule
	                                  that decides whether a stage-2 score saturates at 255 (src/dp/ungapped_simd.cpp:69-87) */
	int32_t query_translated;      /* align_mode.query_translated: 1 for blastx blocks (six frames per read); enables the short-frame
	                                  rules of src/search/stage2.h:51,58-63 (window = frame length for frames of <= 85 letters) */
	int32_t seed_encoding;         /* 0 = spaced-factor seeds of the double-indexed algorithm (--algo 0); 1 = the hashed seeds of the
	                                  query-indexed algorithm (--algo 1, what AUTO picks for small query sets against databases of
	                                  >= 256 MB: run/double_indexed.cpp:267-300), set by dmnd_seed_params_set_query_indexed */
	int32_t cutoff_table_short[32]; /* CutoffTable(ungapped_evalue_short), used for translated frames of 61..85 letters (stage2.h:51) */
} dmnd_seed_params;

/* One stage-2 seed hit = Search::Hit (src/search/hit.h:30-47): query context index, reference location
 * (offset into the DMND_TARGET block data), seed offset inside the query, stage-1 score
 * (0xFFFF when the ungapped filter is off, as the reference writes: stage2.h:86,112,139). */
typedef struct {
	uint32_t query;
	int32_t seed_offset;
	int64_t subject;
	int32_t score;
	int32_t pad;
} dmnd_seed_hit;

/* Fills the --fast configuration (one shape 1101110101101111, murphy10 reduction, Hamming id 11, seed cut 0.9,
 * 4 index chunks; search/setup.cpp:43,211-212) for `threads` reference threads (seedp_bits depends on it,
 * setup.cpp:306-309). */
int dmnd_seed_params_fast(dmnd_seed_params* p, int threads);
/* Default sensitivity (two shapes of weight 10: 111101110111, 111011010010111; ungapped e-value 10000, seed cut 0.8;
 * search/setup.cpp:46,82-84); the cutoff table is derived from the scoring parameters (lambda, K). */
int dmnd_seed_params_default(dmnd_seed_params* p, int threads, const dmnd_params* scoring);
/* Runs the whole seed stage on the uploaded blocks (both must have been uploaded WITH limits). Supported:
 * spaced seeds, ungapped e-value filter off (the --fast family). Hits stay in device memory;
 * *n_hits returns their number. */
int dmnd_seed_search(dmnd_ctx* ctx, const dmnd_seed_params* params, int64_t* n_hits);
/* Optional: allocates the device buffers of a seed search with these parameters for a query block of query_block_len bytes
 * (SequenceSet layout) ahead of time. May run on another host thread while the caller uploads and masks the blocks of the same
 * context: the first search then starts without tens of milliseconds of allocations. Changes no result. */
int dmnd_seed_reserve(dmnd_ctx* ctx, const dmnd_seed_params* params, int64_t query_block_len);
/* One query block against many reference blocks: with reuse on, dmnd_seed_search keeps the query side of every shape (seed table,
 * position lists, bitmaps) in HBM and a later call on the same query block and parameters only resets the per-reference-block
 * marks instead of indexing the queries again (the reference rebuilds its query seed arrays for every block pair). Off by default;
 * results are identical either way. */
int dmnd_set_query_index_reuse(dmnd_ctx* ctx, int on);
/* Copies the hits of the last dmnd_seed_search to the caller, sorted by (query, subject, seed_offset)
 * (the reference sorts by query before extension, align/align.cpp:233). cap < n -> DMND_E_CAP. */
int dmnd_seed_hits(dmnd_ctx* ctx, dmnd_seed_hit* out, int64_t cap);
/* device milliseconds of the last dmnd_seed_search: [0] index queries [1] stream reference [2] mask [3] pair filter [4] total */
int dmnd_seed_kernel_ms(const dmnd_ctx* ctx, double ms[5]);

/* -- extension stage: replaces Extension::extend for every query of a block (src/align/extend.cpp:226-420;
 *    called per query from align_worker, src/align/align.cpp:157) as one block-wide batch ---------------------- */
/* One round-1 DpTarget as the extension stage builds it from seed hits (x-drop ungapped extension, chaining,
 * Extension::band, add_dp_targets: src/align/ungapped.cpp:62, chaining/greedy_align.cpp:482, align/gapped_score.cpp:107) */
typedef struct {
	uint32_t query, target;       /* block sequence ids; query = query id * query_contexts + frame */
	int32_t d_begin, d_end;
	int32_t ungapped_score;       /* WorkTarget::ungapped_score = max stage-1 score of the target's seed hits */
} dmnd_plan_target;

/* One reported alignment = one Hsp of an Extension::Match (src/align/extend.h:34-66); with max_hsps = 1 (the default) a match has
 * one record, else its HSPs are consecutive records (dmnd_set_max_hsps) */
typedef struct {
	uint32_t query, target;       /* query id (= block sequence id / query_contexts), target block id */
	int32_t ungapped_score, d_begin, d_end;
	int32_t frame;                /* query context of the HSP: 0 for blastp; 0-2 forward, 3-5 reverse frames for blastx (Hsp::frame) */
	int32_t read_begin, read_end; /* frameshift alignment (blastx -F): Hsp::query_source_range, the alignment's interval of the read -- the HSP
	                                 changes frame, so frame (that of its first position) and q_begin / q_end alone do not give it; else 0, 0 */
	double evalue, bit_score;
	dmnd_hsp hsp;
} dmnd_match;

/* Host-only part (no GPU needed): per-query Hauser composition bias (cbs_out: int8 array parallel to qdata, may be
 * NULL) and the round-1 DpTargets for seed hits sorted by query. */
int dmnd_extend_plan(const dmnd_params* params, const int8_t* qdata, const int64_t* qlimits, int64_t nq,
	const int8_t* tdata, const int64_t* tlimits, int64_t nt, const dmnd_seed_hit* hits, int64_t n_hits, int threads, int query_contexts,
	int8_t* cbs_out, dmnd_plan_target* out, int64_t cap, int64_t* n_out);
/* Whole extension stage on the uploaded blocks (qdata/tdata: the caller's host copies of the same blocks). hits must be
 * sorted by query. Matches come out ordered by query, then as the reference orders them (e-value, score, target).
 * transcript may be NULL (then dmnd_hsp::transcript_off = -1). Blastp defaults: max_target_seqs 25, max_hsps 1.
 * threads bounds the host threads: blocks with >= 2048 queries run as up to 8 runners (own HIP stream each), worker threads
 * under a runner only when its share of the seed hits is large. Without transcripts round 1 keeps its trace rows in HBM and
 * round 2 only walks them (dmnd_extend_stats[10] = 0 then); the records are the same either way. */
int dmnd_extend(dmnd_ctx* ctx, const int8_t* qdata, const int8_t* tdata, const dmnd_seed_hit* hits, int64_t n_hits,
	int threads, uint32_t hsp_values, dmnd_match* out, int64_t cap, int64_t* n_out,
	uint8_t* transcript, int64_t transcript_cap, int64_t* transcript_used);
/* Gapped filter (replaces Extension::gapped_filter, src/align/gapped_filter.cpp:80-109, with
 * scan_diags64/128 + diag_alignment, src/dp/scan_diags.cpp:30,128,277, and make_profile8, src/dp/score_profile.cpp:33).
 * dmnd_set_gapped_filter: Search::Config::gapped_filter_evalue (1.0 for --sensitive and above, 0 = off = default and
 * --fast; search/setup.cpp:40-53); builds the two CutoffTable2D tables (evalue1 = 2000 and this one).
 * dmnd_gapped_filter: per seed hit (any order) flags[h] = 1 iff the hit passes stage 1 (64 diagonals, +-100 columns)
 * and stage 2 (128 diagonals, +-200 columns) against the cutoffs of its (query length, target length); a target
 * survives iff any of its hits is flagged. scores (optional, 2 per hit) receives the two filter values (f2 = -1 when
 * stage 2 did not run). use_cbs: add the uploaded Hauser bias (dmnd_upload_cbs) to the profile, as the reference does
 * with composition based statistics on. Blocks must be uploaded with limits. dmnd_extend applies it by itself. */
int dmnd_set_gapped_filter(dmnd_ctx* ctx, double gapped_filter_evalue);
int dmnd_gapped_filter(dmnd_ctx* ctx, const dmnd_seed_hit* hits, int64_t n_hits, int use_cbs, uint8_t* flags, int32_t* scores);
/* device time (ms) of the gapped filter kernel of the last dmnd_gapped_filter / dmnd_extend */
double dmnd_gapped_filter_ms(const dmnd_ctx* ctx);
/* Sensitivity presets (Sensitivity enum + sensitivity_traits, src/search/setup.cpp:40-53): shapes, seed cut, ungapped
 * e-value, and through *gapped_filter_evalue (may be NULL) the value to pass to dmnd_set_gapped_filter. */
enum { DMND_SENS_FAST = 0, DMND_SENS_DEFAULT = 1, DMND_SENS_MID_SENSITIVE = 2, DMND_SENS_SENSITIVE = 3, DMND_SENS_MORE_SENSITIVE = 4,
       DMND_SENS_VERY_SENSITIVE = 5, DMND_SENS_ULTRA_SENSITIVE = 6 };
int dmnd_seed_params_preset(dmnd_seed_params* p, int sensitivity, int threads, const dmnd_params* scoring, double* gapped_filter_evalue);
/* -c / --index-chunks on a preset (config.lowmem_, src/run/double_indexed.cpp:297): sets index_chunks and recomputes
 * seedp_bits as Search::seedp_bits does (src/search/setup.cpp:306-309). The chunk of a seed decides which shape/chunk
 * pass sees it first, i.e. the left-most filter, so results depend on it exactly as in the reference. */
int dmnd_seed_params_set_index_chunks(dmnd_seed_params* p, int index_chunks, int threads);
/* Switches a preset to the reference's query-indexed algorithm (config.algo == QUERY_INDEXED: run/double_indexed.cpp:276-300,
 * search/seed_array/seed_iterator.h:161-198, enum_seeds.h:125-153): hashed seed encoding -- every window that ends in an amino
 * acid is a seed, mask and stop letters inside it read as class 0 --, low-complexity seeds dropped and masked on the query side
 * when its seeds are enumerated instead of per joined group, one index chunk. The reference additionally masks the reference
 * block lazily, i.e. AFTER the seed stage (extend.cpp:168-181): callers run dmnd_mask_block(DMND_TARGET) between
 * dmnd_seed_search and dmnd_extend in this mode. */
int dmnd_seed_params_set_query_indexed(dmnd_seed_params* p, int threads);
/* The reference's choice under --algo auto (run/double_indexed.cpp:267-288) for a query block (host copy, SequenceSet layout)
 * and a database of db_bytes (the size the reference looks at: the .dmnd file on disk): *query_indexed = 1 iff the query
 * block has at most 32 Mi letters, the database at least 256 MiB, and the largest per-shape hash set of the query seeds
 * (next_pow2(1.25 x distinct hashed seeds), HashedSeedSet, data/seed_set.cpp:91-119,144) stays within 32 Mi entries.
 * Host arithmetic only. */
int dmnd_auto_query_indexed(const dmnd_seed_params* p, const int8_t* qdata, const int64_t* qlimits, int64_t nq, int64_t db_bytes, int* query_indexed);
/* Sensitive mode seed configuration (16 shapes of weight 8, search/setup.cpp:86-102; ungapped e-value 10000, seed cut 1.0) */
int dmnd_seed_params_sensitive(dmnd_seed_params* p, int threads, const dmnd_params* scoring);

/* align_mode.query_contexts (src/basic/basic.cpp:40-60): 1 = blastp; 6 = blastx, the DMND_QUERY block then holds the six
 * translated frames of every read consecutively (Block::push_back, src/data/block/block.cpp:82-100; frame order of
 * Translator::translate, src/util/sequence/translate.h:62-108) and seed hits carry the frame's block sequence id. */
int dmnd_set_query_contexts(dmnd_ctx* ctx, int contexts);
/* Six-frame translation of one DNA read (letters 0-4 = ACGTN) exactly as the reference loads a blastx query:
 * standard genetic code, stop codons = letter 24, ORFs shorter than config.min_orf_len masked to 23 (find_orfs,
 * src/util/sequence/sequence.cpp:180-197). out[f] must hold len/3 letters each; lens[f] receives the frame lengths. */
int dmnd_translate(const int8_t* dna, int32_t len, int8_t* out[6], int32_t lens[6]);
/* The same with the options of a translated search: gencode = --query-gencode (NCBI table number; Translator::init,
 * src/basic/basic.cpp:116-139, "Invalid genetic code id." for a number it does not have), strands = --strand as a mask (1 plus,
 * 2 minus, 3 both: the frames of a strand that is not searched are filled with mask letters, frame_mask
 * src/data/sequence_file.cpp:286-294), min_orf = --min-orf (0 = by read length). dmnd_translate = (1, 3, 0). */
int dmnd_translate_opts(const int8_t* dna, int32_t len, int gencode, int strands, int min_orf, int8_t* out[6], int32_t lens[6]);
/* BLAST tabular line of a translated match: qstart/qend in DNA coordinates of the read (TranslatedPosition,
 * src/basic/translated_position.h:125-175; reverse frames print qstart > qend). */
int dmnd_format_tab_translated(const dmnd_match* m, const char* qseqid, const char* sseqid, int32_t source_len, char* buf, int64_t cap);

/* tantan repeat masking of an uploaded block, in place in HBM (hard mask: letter 23), as the reference masks the
 * reference block and the query block with its default --masking tantan (mask_seqs, src/masking/masking.cpp:225-251;
 * Util::tantan::mask, src/masking/tantan.cpp:112; likelihood ratios exp(lambda * score), masking.cpp:134-155).
 * Results are bit-identical to the reference's AVX2 build (float operation order restated, see csrc/mask_core.h).
 * host_data (may be NULL): the caller's host copy of the block as it stands in HBM (the bytes it was uploaded from, block raw
 * length); the mask letter is written over the masked positions -- from a list of those positions, not by copying the block
 * back -- so that the extension stage's host part reads the same letters. *n_masked (may be NULL) = number of positions at or
 * above the mask probability.
 * Calls on ONE context are serialized by the caller: the masking calls, dmnd_seed_search and the block joins share the context's
 * sort scratch (a helper thread may run dmnd_upload_block of the other block beside them, nothing else). The tantan work space
 * (4.25 B per letter) stays with the context for the next block; DMND_MASK_SCRATCH_KEEP_MB (default 16384) bounds what is kept. */
int dmnd_mask_block(dmnd_ctx* ctx, int which, int8_t* host_data, int64_t* n_masked);
/* The same for a subset of the block's sequences (block sequence ids, any order, no duplicates): what the reference's LAZY masking
 * does -- with the query-indexed algorithm it masks a target only when the extension stage loads it (align/extend.cpp:168-181), i.e.
 * the targets that have seed hits. tantan works sequence by sequence, so the letters of those targets come out as dmnd_mask_block
 * leaves them; the scratch and the kernel shrink with the subset (C2: 15 k of 10^6 sequences). */
int dmnd_mask_sequences(dmnd_ctx* ctx, int which, int8_t* host_data, const int32_t* seq_ids, int64_t n, int64_t* n_masked);
double dmnd_mask_kernel_ms(const dmnd_ctx* ctx);
/* The lambda of those likelihood ratios (host only): the scale at which the matrix's implied letter probabilities are valid
 * (cbrc::LambdaCalculator::calculate, src/lib/tantan/LambdaCalculator.cc), or -1 where the matrix has none (PAM250) -- the value
 * the reference then uses too. */
double dmnd_masking_lambda(const dmnd_params* params);
/* SEG low-complexity masking (`--masking seg`: the reference then hard-masks the reference block with NCBI's SEG and leaves the queries
 * alone, src/run/config.cpp:125-134, src/masking/masking.cpp:172-192, src/lib/blast/blast_seg.cpp). Host code, as in the reference:
 * the driver masks the block it is about to upload. dmnd_seg_ranges: the segments [begin, end] (inclusive) of one sequence (letters
 * 0-25), ascending; *n = their number (DMND_E_CAP if above cap). dmnd_seg_mask_block: letter 23 over every segment of every sequence
 * of a SequenceSet block (data, limits as for dmnd_upload_block), `threads` host threads. dmnd_seg_lnfact: the ln(n!) SEG computes
 * with (six-decimal table up to 10000, Stirling above). */
int dmnd_seg_ranges(const int8_t* seq, int32_t len, int32_t* ranges, int32_t cap, int32_t* n);
int dmnd_seg_mask_block(int8_t* data, const int64_t* limits, int64_t n_seqs, int threads, int64_t* n_masked);
double dmnd_seg_lnfact(uint32_t n);
/* Motif soft masking (default on up to --sensitive: sensitivity_traits.motif_masking, search/setup.cpp:40-53,322-335): while seeds
 * are enumerated the reference masks stretches covered by abundant 8-mer motifs (mask_motifs, masking/masking.cpp:110-131; the
 * letters come back before the filters and the extension run, Block::soft_mask / remove_soft_masking, data/block/block.cpp:164-177),
 * and query seed positions whose shape window touches such a stretch keep a SEED_MASK bit (MaskingTable::remove, masking.cpp:89-102).
 * dmnd_set_motif_table: the process-wide motif table as Kmer<8> codes (base-20 polynomial of the 8 letters; the reference's
 * table is src/masking/motifs.cpp -- tools/make_motif_table.py extracts it into diamond_amd/motifs.bin at build time, it is not
 * part of this repository). dmnd_soft_mask_block: prepares the masked view of an uploaded (and, if wanted, tantan-masked) block
 * in HBM; dmnd_seed_search then generates seeds from it -- query side always, reference side for spaced seeds only, as the
 * reference does (search/stage0.cpp:125-127). Uploading or masking a block drops its view. *n_covered = letters under motifs. */
int dmnd_set_motif_table(const uint64_t* codes, int64_t n);
int64_t dmnd_motif_table_size(void);
/* dmnd_set_motif_table is process-wide (the reference's table is a global too); a context takes a snapshot of it, under a lock, when
 * it soft-masks a block. A process that runs contexts with DIFFERENT tables gives each its own: n motifs for this context alone
 * (n = 0: none, i.e. no soft masking on this context; n = -1: back to the process-wide table). */
int dmnd_set_context_motif_table(dmnd_ctx* ctx, const uint64_t* codes, int64_t n);
int dmnd_soft_mask_block(dmnd_ctx* ctx, int which, int64_t* n_covered);

/* Frameshift alignment of translated queries (-F PENALTY, config.frame_shift; --range-culling, --range-cover): dmnd_extend then runs
 * the reference's legacy pipeline (align/legacy/banded_swipe_pipeline.cpp, query_mapper.cpp; entered at align/align.cpp:168) over
 * the three-frame sweep (dmnd_frameshift_swipe) instead of Extension::extend. penalty 0 = off (the default). range_culling != 0:
 * targets are culled per read range (RangeCulling, output/target_culling.h:112-160: a target is dropped when range_cover per cent
 * of its alignments' read range is covered by -k better ones, or by better-scoring ones with --top) instead of per query.
 * channels = the int16 vector width of the reference build whose score-only pass is to be reproduced (16: AVX2). Needs
 * dmnd_set_query_contexts(6) and the read lengths (dmnd_set_query_source_lengths). */
int dmnd_set_frameshift(dmnd_ctx* ctx, int penalty, int range_culling, double range_cover, int channels);
/* --comp-based-stats (Stats::CBS, stats/cbs.h:112-196): 0 = none; 1 = Hauser composition bias (default; HauserCorrection,
 * stats/hauser_correction.cpp); 2 / 3 = Hauser bias, and a composition-adjusted scoring matrix for every target that NCBI's
 * conditional test selects; 4 = no bias, every target gets an adjusted matrix; 5 = no bias, every target gets either the full
 * adjustment or the standard matrix rescaled to the pair's lambda. An adjusted target is swept with ITS matrix and without
 * the bias (dp/swipe/target_iterator.h:124-134, swipe.h:43-54); seed stage, gapped filter, x-drop stage and chaining keep the
 * standard matrix. Modes above 1 need one of the eight standard matrices (basic/config.cpp:837) and untranslated queries
 * (config.cpp:700): DMND_E_ARG otherwise, from dmnd_extend. */
int dmnd_set_comp_based_stats(dmnd_ctx* ctx, int mode);
/* Composition-based matrix adjustment of one (query, target) pair on the host (double precision, no device): what
 * WorkTarget::WorkTarget computes per target (align/ungapped.cpp:44-58). dmnd_extend calls the same code for every target it
 * plans; the entry points exist so that a caller -- and the CPU tests -- can reproduce a single matrix.
 *   dmnd_cbs_composition   Stats::composition + count_true_aa (stats/cbs.cpp:52-77): frequencies of the 20 residues
 *   dmnd_cbs_rule          Stats::adjust_matrix (cbs.cpp:94-112): *rule = -1 (keep the standard matrix), 0 (rescale it to the
 *                          pair's lambda, mode 5 only) or 4 (relative-entropy adjustment); query_true_aa = residues of the query
 *   dmnd_cbs_target_matrix Stats::TargetMatrix::TargetMatrix (cbs.cpp:114-173): matrix_out[target letter * 32 + query letter],
 *                          32 x 32 int8 (rows / columns above letter 25 at -128), for rule 0 or 4
 *   dmnd_cbs_ideal_lambda  ScoreMatrix::ideal_lambda (score_matrix.cpp:61): < 0 if params does not hold a standard matrix */
int dmnd_cbs_composition(const int8_t* seq, int32_t len, double* comp20, int32_t* true_aa);
int dmnd_cbs_rule(const dmnd_params* params, int mode, const double* query_comp20, int32_t query_true_aa, const int8_t* target, int32_t target_len, int32_t* rule);
int dmnd_cbs_target_matrix(const dmnd_params* params, int rule, const double* query_comp20, int32_t query_true_aa, const int8_t* target, int32_t target_len, int8_t* matrix_out);
double dmnd_cbs_ideal_lambda(const dmnd_params* params);
/* The part of the sensitivity that the extension stage reads: the band widths around a chain (Extension::Mode::BANDED_FAST up to
 * --sensitive, BANDED_SLOW from --more-sensitive up: src/align/extend.cpp:62-75, gapped_score.cpp:41-73), and ranking_chunk_size's
 * unit of reference-block letters (2e9, or 8e8 from --very-sensitive up: extend.cpp:79-92). */
int dmnd_set_sensitivity(dmnd_ctx* ctx, int sensitivity);
/* --ext banded-fast | banded-slow | full (Extension::Mode, src/align/extend.cpp:51-58): overrides the band mode of the sensitivity;
 * FULL skips chaining and aligns every target that has a seed hit over its whole matrix in both rounds (align/ungapped.cpp:70-75,
 * gapped_score.cpp:123-130,204-205, gapped_final.cpp:97-98) -- query length + target length must stay below DMND_MAX_BAND. */
enum { DMND_EXT_DEFAULT = -1, DMND_EXT_BANDED_FAST = 0, DMND_EXT_BANDED_SLOW = 1, DMND_EXT_FULL = 2 };
int dmnd_set_extension_mode(dmnd_ctx* ctx, int mode);
/* --top PERCENT (config.toppercent): report the targets whose bit score is within PERCENT of the best one instead of the first
 * -k ones (output_range / append_hits / ranking_chunk_size with toppercent, align/culling.cpp:97-145, align/extend.cpp:88-89,336);
 * percent < 0 switches it off. dmnd_join_blocks_top is the block join for such a run (JoinRecord::cmp_score + GlobalCulling,
 * output/join_blocks.cpp:133-136, output/target_culling.h:62-63). */
int dmnd_set_top_percent(dmnd_ctx* ctx, double percent);
/* --id, --query-cover, --subject-cover (percentages; 0 = off): HSPs below are removed after round 2 and the extension takes more
 * of the ranked targets, 16 or more at a time, until -k matches pass (filter_hsp, align/culling.cpp:147-170; the stepping of
 * align(), align/gapped_final.cpp:105-152; first_round_culling = false, align/extend.cpp:272). --min-score BITS (0 = off)
 * replaces the e-value cutoff (ScoreMatrix::report_cutoff, stats/score_matrix.cpp:234-239). Query cover is only available for
 * untranslated queries. */
int dmnd_set_filters(dmnd_ctx* ctx, double min_id, double query_cover, double subject_cover, double min_bit_score);
/* Translated queries: the lengths of the DNA reads of the uploaded query block (one per query = per six contexts), which the
 * query cover of an HSP is measured against (Hsp::query_cover_percent over query_source_range). Cleared by the next upload of
 * the query block. */
int dmnd_set_query_source_lengths(dmnd_ctx* ctx, const int32_t* lengths, int64_t n_queries);
/* --no-self-hits (filter_hsp, align/culling.cpp:166-169): an HSP is removed when the query and the target have the same letters
 * AND the same title. The library compares the letters; `same_title` (called only for such pairs, with the block-local query and
 * target ids of the uploaded blocks) answers for the titles. NULL switches the filter off. */
typedef int (*dmnd_same_title_fn)(void* user, uint32_t query, uint32_t target);
int dmnd_set_no_self_hits(dmnd_ctx* ctx, dmnd_same_title_fn same_title, void* user);
int dmnd_join_blocks_top(dmnd_match* records, int64_t n, double top_percent, int64_t* n_out);
/* -k / --max-target-seqs (default 25, src/basic/config.h:55) */
int dmnd_set_max_target_seqs(dmnd_ctx* ctx, int k);
/* --max-hsps N (config.max_hsps, default 1; 0 = no limit): HSPs reported per target. With N != 1 dmnd_extend
 *  - sends EVERY reported round-1 band of a target through round 2 (add_dp_targets, align/gapped_final.cpp:62-76) and culls the
 *    target's HSP list as Match::inner_culling does (align/culling.cpp:40-57: Hsp::operator< order, an HSP dropped when half of
 *    its query or subject range lies inside a better one, the list cut at N);
 *  - then searches alternative HSPs (recompute_alt_hsps, align/alt_hsp.cpp:86-142): every reported target is copied per query
 *    context with the subject ranges of its HSPs overwritten by letter 25 and swept over the whole matrix, round after round,
 *    until a sweep finds nothing above the report cutoff, the copy is masked through, or the target has N HSPs;
 *  - returns one dmnd_match record per HSP, the records of a (query, target) pair consecutive and in the list's order: the first
 *    one carries the target's place among the query's targets (Match::filter_evalue / filter_score).
 * Translated queries need dmnd_set_query_source_lengths (the envelope test works on the read's coordinates). */
int dmnd_set_max_hsps(dmnd_ctx* ctx, int n);
/* xdrop_ungapped for every seed hit of the resident block pair in one launch (the optional entry;
 * src/dp/ungapped_align.cpp:151-199): from the hit's position the diagonal is walked left and right until a sequence end or until
 * the running score has fallen xdrop below the best. out[k] = DiagonalSegment(qa - delta, sa - delta, len + delta, score) of hit k:
 * i = position in the query sequence, j = block offset in the reference block. use_bias != 0: the queries' Hauser bias is added to
 * every letter score (Extension::extend's call); xdrop <= 0: config.raw_ungapped_xdrop of the context's matrix (12.3 bits). */
typedef struct { int32_t i; int64_t j; int32_t len, score; } dmnd_diagonal_segment;
int dmnd_xdrop_ungapped(dmnd_ctx* ctx, const dmnd_seed_hit* hits, int64_t n_hits, int use_bias, int xdrop, dmnd_diagonal_segment* out);
/* --global-ranking N (config.global_ranking_targets; align/global_ranking/): instead of extending every reference block's seed
 * hits, the search keeps per query the N targets of the WHOLE database with the best ungapped score and extends only those, over
 * the full matrix, after the last block. Three calls:
 *  - dmnd_rank_targets after dmnd_seed_search / dmnd_seed_hits of a block pair (get_query_hits_reextend, table.cpp:108-121): one
 *    record per (query, target) of the hits = the best x-drop ungapped score over the target's seed hits (no composition bias) and
 *    the query context it was found in (target_score, table.cpp:88-106); target = block sequence id. Host code, as in the reference.
 *  - dmnd_rank_update (merge_hits, table.cpp:135-151): merges such records (target already turned into a database ordinal by the
 *    caller, records grouped by query) into a table of n entries per query, zero-initialised by the caller: per target its best
 *    score, rows ordered by (score descending, target ascending); an entry with score 0 is empty.
 *  - the final extension (global_ranking/extend.cpp:133-233): the caller loads the targets the table names as ONE reference block,
 *    masks and uploads it, calls dmnd_set_global_ranking(ctx, N) and dmnd_set_extension_mode(ctx, DMND_EXT_FULL), and hands
 *    dmnd_extend one seed hit per table entry { query = query id * contexts + context, seed_offset 0, subject = first letter of the
 *    target, score }: no ranking chunks (extend.cpp:80) and no gapped filter (extend.cpp:206) then. */
typedef struct { uint32_t query, target; uint16_t score; uint8_t context, pad; } dmnd_ranked_target;
int dmnd_rank_targets(dmnd_ctx* ctx, const int8_t* qdata, const int8_t* tdata, const dmnd_seed_hit* hits, int64_t n_hits, int threads,
	dmnd_ranked_target* out, int64_t cap, int64_t* n_out);
int dmnd_rank_update(dmnd_ranked_target* table, int64_t n_queries, int n, const dmnd_ranked_target* records, int64_t n_records);
int dmnd_set_global_ranking(dmnd_ctx* ctx, int n);
/* Multi-block databases (-b / --block-size): the records of one query block against several reference
 * blocks (dmnd_match::target already offset to database ordinals by the caller, blocks in any order) are merged per query
 * the way join_query does it: ascending by (e-value, score descending, target ordinal) = JoinRecord::cmp_evalue
 * (src/output/join_blocks.cpp:129-142), then the first max_target_seqs of every query are kept (GlobalCulling,
 * src/output/target_culling.h:70-88). In place; records stay grouped by ascending query. The outcome equals the
 * reference run with the same block boundaries, not the single-block run (ranking and culling happen per block). */
int dmnd_join_blocks(dmnd_match* records, int64_t n, int max_target_seqs, int64_t* n_out);
/* The same join with --range-culling (blastx -F n --range-culling / --long-reads): the reference's join culler is RangeCulling
 * then (TargetCulling::get, src/output/target_culling.cpp:22-28; src/output/target_culling.h:107-163): in the merged order
 * (cmp_evalue, or cmp_score when top_percent >= 0) a target is skipped when range_cover per cent of its HSPs' intervals of the
 * read (dmnd_match::read_begin / read_end) are covered already -- by max_target_seqs kept alignments, or with --top by one kept
 * alignment scoring at least score / (1 - top / 100) -- and every kept target adds its intervals. top_percent < 0: no --top. */
int dmnd_join_blocks_range(dmnd_match* records, int64_t n, int max_target_seqs, double top_percent, double range_cover, int64_t* n_out);
/* The block join ON THE DEVICE (round 5; SURVEY.md 8(e): the final top-k merge), for records that are in HBM already -- received
 * from the other ranks over RCCL, or produced block after block on this GPU: the union of the per-block lists sorted into
 * join_query's order (three stable radix sorts of a permutation: (score descending, target), e-value, query =
 * JoinRecord::cmp_evalue, src/output/join_blocks.cpp:129-142; top_percent >= 0: cmp_score and GlobalCulling's top-per-cent rule,
 * src/output/target_culling.h:56-64), the first max_target_seqs of every query kept and written to out_dev (HBM, >= n records, no
 * overlap with records_dev); *n_out = their number. Every (query, target) pair must have ONE record (--max-hsps 1, the default;
 * the host forms above carry HSP groups and range culling). max_query: largest query id among the records, 0 = unknown.
 * dmnd_join_blocks_device_host: the same for records in host memory (upload, join, download of the survivors), in place. */
int dmnd_join_blocks_device(dmnd_ctx* ctx, const dmnd_match* records_dev, int64_t n, int max_target_seqs, double top_percent, uint32_t max_query,
	dmnd_match* out_dev, int64_t* n_out);
int dmnd_join_blocks_device_host(dmnd_ctx* ctx, dmnd_match* records, int64_t n, int max_target_seqs, double top_percent, int64_t* n_out);
/* The final merge of a search that ran on several GPUs of one node, over RCCL (what `diamond-hip --gpus N` calls
 * per query block): n_ctx contexts, one per GPU, each with the records its GPU produced (host memory, database-wide target ordinals,
 * every (query, target) once). GPU j becomes the owner of the query range [j Q/N, (j + 1) Q/N), Q = n_queries: the records are
 * ordered by owner and uploaded, ONE grouped RCCL exchange (ncclGroupStart; ncclSend / ncclRecv per (source, owner); ncclGroupEnd)
 * moves them to their owners device to device, every owner merges its range with dmnd_join_blocks_device, the survivors are
 * written to `out` owner after owner = in query order: the records dmnd_join_blocks (top_percent < 0) / dmnd_join_blocks_top give for
 * the concatenated input. RCCL is loaded at run time (librccl.so). Contexts that all share one device exchange with device-to-device
 * copies instead (RCCL refuses two ranks on a device; the test hook of a 1-GPU box); one context alone goes through RCCL with itself.
 * *transport_used (may be NULL): 1 = RCCL, 2 = copies. */
int dmnd_join_ranks(dmnd_ctx* const* ctx, int n_ctx, const dmnd_match* const* records, const int64_t* counts, int64_t n_queries, int max_target_seqs,
	double top_percent, dmnd_match* out, int64_t cap, int64_t* n_out, int* transport_used);
/* The exchange plan of dmnd_join_ranks as plain arithmetic, no device needed (the code the RCCL path runs; tests/test_rank_join_plan.py
 * moves bytes by it for 2 - 8 ranks): queries[g][i] = query of record i of source g. Outputs (n x n, row-major): cnt[g][j] = records
 * source g sends to owner j, send_off[g][j] = their offset in g's owner-ordered copy, recv_off[j][g] = their offset in owner j's
 * receive buffer; n_recv[j] (n entries); place[g][i] (may be NULL) = position of record i in g's owner-ordered copy. */
int dmnd_join_ranks_plan(int n, const int64_t* counts, const uint32_t* const* queries, int64_t n_queries, int64_t* cnt, int64_t* send_off, int64_t* recv_off,
	int64_t* n_recv, int64_t* const* place);
/* Touches every HIP stream the context owns (its own and those of the extension stage's runners) with an empty marker and
 * waits for them. A driver that calls hipDeviceSynchronize between batches (bench.py must, by its timing contract) lets the
 * runtime release idle hardware queues; re-acquiring them costs the next dmnd_extend several milliseconds (measured: +6.5 ms
 * for eight runner streams). Not needed by callers that only use the library's own stream-scoped waits. */
int dmnd_touch_streams(dmnd_ctx* ctx);
/* Statistics of the last dmnd_extend: [0] round-1 DpTargets [1] round-2 DpTargets [2] round-1 cells [3] round-2 cells
 * (DpTarget::cells, src/dp/dp.h:121-124: the GCUPS denominator); host wall ms [4] Hauser+upload [5] chaining [6] round-1
 * call [7] culling [8] round-2 call; device ms [9] round-1 swipe [10] round-2 swipe [11] traceback. */
int dmnd_extend_stats(const dmnd_ctx* ctx, double out[12]);
/* Round 6: the extension stage plans on the device -- grouping the seed hits by target (load_hits, src/align/load_hits.h:44),
 * diagonal segments (src/align/ungapped.cpp:62-126), chaining (src/chaining/greedy_align.cpp:482-497) and band construction
 * (src/align/gapped_score.cpp:107-180) of ALL (query, target) pairs of a block pair in a few launches (one query context, banded
 * modes). Of the last dmnd_extend: [0] (query, target) groups the device found, [1] groups it left to the host (more than 32 hits
 * or 16 segments), [2] round-1 bands it planned; all 0 = the host planned (several contexts, --ext full, hits not in
 * (query, location) order). Lets a test tell which path produced the records. */
int dmnd_extend_plan_stats(const dmnd_ctx* ctx, double out[3]);
/* Round 6: for the queries whose targets fit one ranking chunk (src/align/extend.cpp:79-92) the rest of the extension stage runs in
 * HBM as well -- DpTargets and their launch order from the bands (DP::BandedSwipe::bin, src/dp/swipe/swipe_wrapper.cpp:75-102), best
 * HSP per target and report cutoff (src/align/gapped_score.cpp:182-268), culling (src/align/culling.cpp:97-113,189-203), round 2 as a
 * walk of the kept traces (src/align/gapped_final.cpp:66-160), match records in output order (src/align/extend.h:51-56). The host
 * writes its own e-value and bit score into the records. Of the last dmnd_extend: [0] queries extended that way, [1] of them handed
 * back to the host path (two device e-values too close to order safely, or a saturated 16-bit sweep), [2] round-1 DpTargets,
 * [3] records, [4] sum over the round-1 DpTargets of band diagonals x anti-diagonal steps and [5] of the 128 P diagonals their wavefront
 * holds x steps ([4] / [5] = lane use of the sweeps), [6] DP cells of the device half's round-2 targets, [7] of those swept again in
 * round 2 (their round-1 sweep kept no trace rows), [8] device ms of those sweeps, [9] reserved; all 0 = every query took the host path (other modes: --max-hsps != 1, --top,
 * filters, matrix adjustment, --ext full, transcripts wanted, translated queries). */
int dmnd_extend_device_stats(const dmnd_ctx* ctx, double out[10]);
/* Round 6: the records of the last dmnd_extend where they lie in HBM, complete (the host's e-values and bit scores are written back
 * into them): *records_dev is valid until the context's next dmnd_extend; *n = -1 (and NULL) when part of the records only exists on the
 * host (queries that took the host path: other modes, or a query handed back). The records dmnd_extend returned are the same. */
int dmnd_extend_records_device(const dmnd_ctx* ctx, const dmnd_match** records_dev, int64_t* n);
/* The block join (dmnd_join_blocks_device: join_query of src/output/join_blocks.cpp:129-256, one record per (query, target)) over the
 * device-resident records of several contexts' last dmnd_extend -- one context per reference block, all on join_ctx's device; the
 * block-local target ids become database ordinals on the way (target_offset[k] = first sequence of context k's block). The
 * records never leave HBM between round 2 and the join; the survivors come back in `out` (query order). A context without a complete
 * device copy (dmnd_extend_records_device: n = -1) is refused: join the host records then (dmnd_join_blocks_device_host). */
int dmnd_join_contexts_device(dmnd_ctx* join_ctx, dmnd_ctx* const* ctx, const uint32_t* target_offset, int n_ctx, int max_target_seqs, double top_percent,
	uint32_t max_query, dmnd_match* out, int64_t cap, int64_t* n_out);
/* Optional: the first-call allocations of dmnd_extend made ahead of it, for about n_hits_hint seed hits (device work arrays of the
 * x-drop stage, planner and device half; the page-locked result buffer). A driver calls it beside its upload / masking phase, as
 * dmnd_seed_reserve; a hint that is too small costs nothing but the growth inside the call. The query block must be uploaded. */
int dmnd_extend_reserve(dmnd_ctx* ctx, int64_t n_hits_hint);
/* BLAST tabular (-f 6 default fields) line of one match, as the reference prints it; returns the length written. */
int dmnd_format_tab(const dmnd_match* m, const char* qseqid, const char* sseqid, char* buf, int64_t cap);

/* -- output formats over the records + packed transcripts (host only) ---------------------------------------------------
 * `-f 6 FIELD...` (TabularFormat, src/output/blast_tab_format.cpp:46-620) and `-f 0` (PairwiseFormat,
 * src/output/blast_pairwise_format.cpp:24-85) of the reference, printed from what dmnd_extend returns: the record, the HSP's
 * PackedOperation bytes (basic/packed_transcript.h: op << 6 | count for matches / insertions, op << 6 | letter for deletions /
 * substitutions) and the sequences. Fields that need the taxonomy or FASTQ qualities are not part of this build. */
enum {
	DMND_F_QSEQID = 0, DMND_F_QLEN, DMND_F_SSEQID, DMND_F_SALLSEQID, DMND_F_SLEN, DMND_F_QSTART, DMND_F_QEND, DMND_F_SSTART, DMND_F_SEND,
	DMND_F_QSEQ, DMND_F_SSEQ, DMND_F_EVALUE, DMND_F_BITSCORE, DMND_F_SCORE, DMND_F_LENGTH, DMND_F_PIDENT, DMND_F_NIDENT, DMND_F_MISMATCH,
	DMND_F_POSITIVE, DMND_F_GAPOPEN, DMND_F_GAPS, DMND_F_PPOS, DMND_F_QFRAME, DMND_F_BTOP, DMND_F_STITLE, DMND_F_SALLTITLES, DMND_F_QCOVHSP,
	DMND_F_QTITLE, DMND_F_FULL_SSEQ, DMND_F_QNUM, DMND_F_SNUM, DMND_F_SCOVHSP, DMND_F_FULL_QSEQ, DMND_F_QSEQ_GAPPED, DMND_F_SSEQ_GAPPED,
	DMND_F_QSTRAND, DMND_F_CIGAR, DMND_F_QSEQ_TRANSLATED, DMND_F_HSPNUM, DMND_F_COUNT
};
/* One HSP with everything a format reads (HspContext, src/basic/match.h:281-440) */
typedef struct {
	const dmnd_match* match;
	const uint8_t* transcript;     /* the HSP's packed operations (match->hsp.transcript_len bytes), NULL if dmnd_extend ran without an arena */
	const char* qtitle;            /* full FASTA titles */
	const char* stitle;
	const int8_t* qseq;            /* letters of the aligned query context (blastx: the frame match->frame), as the extension stage saw them */
	int32_t qlen;
	int32_t slen;                  /* target length */
	const int8_t* full_sseq;       /* target letters before masking (full_sseq; may be NULL if the field is not printed) */
	const int8_t* source_seq;      /* blastx: the DNA read (A C G T N = 0..4); blastp: NULL */
	int32_t source_len;            /* blastx: its length; blastp: ignored */
	int64_t qnum, snum;            /* ordinal ids in the query file / database */
	const int8_t* qframes[3];      /* blastx: the letters of the three reading frames (offset 0, 1, 2) of the alignment's strand -- a frameshift
	                                  alignment (-F) leaves the frame of qseq, and the fields that walk the alignment (btop, cigar, sseq, qseq_gapped,
	                                  sseq_gapped, qseq_translated) then read these; all NULL: one frame (qseq) */
} dmnd_hsp_view;
/* Field names of --outfmt 6 (blast_tab_format.cpp:46-102) -> DMND_F_*; DMND_E_ARG with the reference's message for an unknown
 * or unavailable field. *needs_transcript = 1 if a field reads the transcript (HspValues::TRANSCRIPT). */
int dmnd_output_fields(const char* const* names, int n, int32_t* ids, int* needs_transcript);
/* One tabular line (fields separated by tabs, newline at the end); returns the length written or DMND_E_CAP. */
int64_t dmnd_format_fields(const dmnd_hsp_view* v, const int32_t* ids, int n, char* buf, int64_t cap);
/* The line of a query without alignments (`--unal 1`; TabularFormat::print_query_intro, blast_tab_format.cpp:776-787): its id,
 * length, title and sequence where a field asks for them, '*' / -1 / 0 elsewhere. full_qseq: the query letters (blastp) or the
 * read's nucleotides (source_seq != NULL). */
int64_t dmnd_format_fields_unaligned(const char* qtitle, const int8_t* qseq, int32_t qlen, const int8_t* source_seq, int32_t source_len,
	const int32_t* ids, int n, char* buf, int64_t cap);
/* `--header simple`: the field keys, tab-separated (TabularFormat::output_header). */
int64_t dmnd_format_fields_header(const int32_t* ids, int n, char* buf, int64_t cap);
/* BLAST pairwise: "Query= ..." block of a query (print_query_intro) and one alignment (print_match). matrix8 = the 32x32
 * substitution matrix of the scoring parameters (midline '+'). */
int64_t dmnd_format_pairwise_intro(const char* qtitle, int32_t qlen, int unaligned, char* buf, int64_t cap);
int64_t dmnd_format_pairwise(const dmnd_hsp_view* v, const int8_t* matrix8, char* buf, int64_t cap);
/* PAF (`-f paf` / 103, src/output/paf_format.cpp:24-66): one line per HSP; v == NULL prints the line of an unaligned query
 * (qtitle given), which this format reports by default. */
int64_t dmnd_format_paf(const dmnd_hsp_view* v, const char* unaligned_qtitle, char* buf, int64_t cap);
/* SAM (`-f sam` / 101, src/output/sam_format.cpp:30-133): one alignment line per HSP (CIGAR, MD:Z and the reference's Z? tags);
 * v == NULL prints the line of an unaligned query (flag 4), which this format reports by default. The @HD/@PG header is the caller's. */
int64_t dmnd_format_sam(const dmnd_hsp_view* v, const char* unaligned_qtitle, char* buf, int64_t cap);
/* BLAST XML (`-f 5` / xml, src/output/xml_format.cpp): header of the file (print_header; program = "blastp" / "blastx", version and
 * database name are the caller's), the <Iteration> opening of a query (print_query_intro; qnum = ordinal of the query in the file),
 * one <Hsp> (print_match: hsp_num == 0 also opens the <Hit> and, for hit_num > 0, closes the one before), the closing of a query
 * (print_query_epilog; db_seqs / db_letters < 0 = not printed) -- the footer is "</BlastOutput_iterations>\n</BlastOutput>". The format
 * reports unaligned queries by default (intro + epilog with unaligned = 1). */
/* Process-wide switches of the writers, as the reference reads them from its global config: --xml-blord-format (Hit_id = gnl|BL_ORD_ID|<snum>,
 * Hit_def = all titles; xml_format.cpp:43-48), --no-parse-seqids (Hit_accession = the id as it is), --sam-query-len (ZQ:i: tag). */
enum { DMND_FMT_XML_BLORD = 1, DMND_FMT_NO_PARSE_SEQIDS = 2, DMND_FMT_SAM_QUERY_LEN = 4,
	DMND_FMT_FRAMESHIFT = 8 };     /* config.frame_shift != 0: qseq_translated follows the alignment (blast_tab_format.cpp:565-574) */
int dmnd_set_format_flags(uint32_t flags);
int64_t dmnd_format_xml_header(const char* program, const char* version, const char* database, const char* first_qtitle, int32_t first_qlen,
	const char* matrix, int gap_open, int gap_extend, double max_evalue, char* buf, int64_t cap);
int64_t dmnd_format_xml_query_intro(const char* qtitle, int64_t qnum, int32_t qlen, char* buf, int64_t cap);
int64_t dmnd_format_xml(const dmnd_hsp_view* v, int32_t hit_num, int32_t hsp_num, const int8_t* matrix8, char* buf, int64_t cap);
int64_t dmnd_format_xml_query_epilog(int unaligned, int64_t db_seqs, int64_t db_letters, double K, double lambda, char* buf, int64_t cap);
/* DAA (`-f 100` / daa: DIAMOND's alignment archive, src/legacy/daa/daa_write.cpp + daa_file.h:30-90). File layout: header
 * (dmnd_format_daa_header: DAA_header1 + DAA_header2, 2448 bytes -- written first with finished = 0 and rewritten at the end), one
 * record per aligned query (dmnd_format_daa_query: size placeholder, length, id, packed sequence -- the block's letters, a read as its
 * DNA -- followed by its dmnd_format_daa_match records; the caller stores the record's byte count after the placeholder in its first 4
 * bytes), a zero uint32, the titles of the targets used (one C string each, in dictionary order: the order of first appearance),
 * their lengths as uint32. dict_id = index of the target in that dictionary. */
typedef struct {
	int64_t build;                 /* DAA_header2::diamond_build */
	int64_t db_seqs, db_letters;   /* of the whole database */
	int64_t db_seqs_used;          /* dictionary size */
	int64_t query_records;
	int32_t mode;                  /* AlignMode: 2 = blastp, 3 = blastx */
	int32_t gap_open, gap_extend;
	double K, lambda, max_evalue;
	const char* matrix;            /* written in lower case */
	int32_t finished;              /* 1: block sizes / types are filled in */
	int64_t alignment_bytes;       /* bytes between the header and the reference names (records + the zero uint32) */
	int64_t ref_name_bytes;
} dmnd_daa_header;
int64_t dmnd_format_daa_header(const dmnd_daa_header* h, char* buf, int64_t cap);
int64_t dmnd_format_daa_query(const char* qtitle, const int8_t* seq, int32_t len, int dna, char* buf, int64_t cap);
int64_t dmnd_format_daa_match(const dmnd_hsp_view* v, uint32_t dict_id, char* buf, int64_t cap);
/* Reading an archive back (`view`): dmnd_daa_match_read parses one match record at p (DAA_query_record::Match::read,
 * src/legacy/daa/daa_record.cpp:52-83: dictionary id, score, frame, begin coordinates -- of a translated query from its DNA position
 * and the read length source_len --, the transcript's offset inside p and its length in hsp.transcript_len; *used = bytes consumed);
 * dmnd_hsp_from_transcript fills the rest of the record as HspContext::parse does (src/basic/hssp.cpp:48-105) from the aligned query
 * context qseq and the transcript, with the e-value of a query of evalue_qlen letters (the reference's view passes the length of
 * frame 0 there) and the scoring parameters of the archive's header (params->db_letters = its database letters). */
int dmnd_daa_match_read(const uint8_t* p, int64_t avail, int translated, int32_t source_len, uint32_t* dict_id, dmnd_match* m, int64_t* transcript_off, int64_t* used);
int dmnd_hsp_from_transcript(const dmnd_params* params, const int8_t* qseq, int32_t qlen, int32_t evalue_qlen, int32_t slen, const uint8_t* transcript, dmnd_match* m);
/* The same for a translated query given as the three reading frames of the alignment's strand (frame offsets 0, 1, 2; m->frame names the
 * first column's): a frameshift alignment (blastx -F) changes frame along its transcript, and its range on the read
 * (m->read_begin / read_end) is that of its first and last column (TranslatedPosition::absolute_interval). */
int dmnd_hsp_from_transcript_frames(const dmnd_params* params, const int8_t* const qframes[3], const int32_t qframe_len[3], int32_t source_len, int32_t evalue_qlen,
	int32_t slen, const uint8_t* transcript, dmnd_match* m);

/* -- timing hooks for bench.py: device time of the DP kernels of the last dmnd_banded_swipe call,
 *    measured with HIP events on the stream the kernels ran on ------------------------------------ */
int dmnd_last_kernel_ms(const dmnd_ctx* ctx, double* swipe_ms, double* traceback_ms);

#ifdef __cplusplus
}
#endif
#endif
