/* oracle/oracle.h -- TEST INFRASTRUCTURE ONLY: C interface of the CPU restatement of the
 * reference's hot path (see each .c file for the reference file:line it follows). */
#ifndef DIAMOND_ORACLE_H
#define DIAMOND_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_OK = 0, ORACLE_ERR_ARG = -1, ORACLE_ERR_TRACEBACK = -2, ORACLE_ERR_CAP = -3 };
enum { ORACLE_SCORE_ONLY = 0, ORACLE_COORDS = 1, ORACLE_TRACEBACK = 2, ORACLE_STATS_FWD = 3, ORACLE_STATS_BWD = 4 };

typedef struct {
	int32_t score, max_col, max_band_row, cols;
	int32_t q_begin, q_end, s_begin, s_end;
	int32_t length, identities, mismatches, positives, gap_openings, gaps;
	int32_t transcript_len;
} oracle_hsp;

int oracle_banded_cols(int qlen, int tlen, int d_begin, int d_end);

/* channel model of the reference's 8-bit vector pass, see banded_swipe.c; penalty 0 = off (the default) */
void oracle_set_channel_band(int own_d_begin, int own_d_end, int penalty);
int oracle_banded_swipe(const int8_t* query, int qlen, const int8_t* cbs,
	const int8_t* target, int tlen, int d_begin, int d_end,
	const int8_t* matrix8, int gap_open, int gap_extend, int mode,
	oracle_hsp* out, uint8_t* transcript, int transcript_cap);

int oracle_swipe_stats(const int8_t* query, int qlen, const int8_t* cbs,
	const int8_t* target, int tlen, int d_begin, int d_end,
	const int8_t* matrix8, int gap_open, int gap_extend, unsigned hsp_values, oracle_hsp* out);

typedef struct {
	double lambda, K, a_I, b_I, a_J, b_J, alpha_I, beta_I, alpha_J, beta_J, sigma, tau;
	double vi_y_thr, vj_y_thr, c_y_thr, db_letters, ln_k;
} oracle_evaluer;

void oracle_evalue_init(oracle_evaluer* e, double lambda, double K, double alpha, double alpha_v, double sigma,
	double u_alpha, double u_alpha_v, int gap_open, int gap_extend, double db_letters);
double oracle_area(const oracle_evaluer* e, double y, double seqlen1, double seqlen2);
double oracle_evalue(const oracle_evaluer* e, int raw_score, unsigned query_len, unsigned subject_len);
double oracle_bitscore(const oracle_evaluer* e, double raw_score);

/* three-frame banded sweep of frameshift alignment, blastx -F (frameshift_swipe.c) */
typedef struct {
	int32_t score, frame, q_begin, q_end, s_begin, s_end, qs_begin, qs_end;
	int32_t length, identities, mismatches, positives, gap_openings, gaps, transcript_len;
} oracle_hsp3;
int oracle_3frame_score(const int8_t* const frames[3], const int32_t lens[3], const int8_t* target, int tlen,
	int band, int i0, int i1, int pos0, const int8_t* matrix8, int gap_open, int gap_extend, int frame_shift, int* max_col, int* overflow);
void oracle_3frame_score_range(int strand, int dna_len, int qlen, int band, int i0, int pos0, int max_col, oracle_hsp3* out);
int oracle_3frame_traceback(const int8_t* const frames[3], const int32_t lens[3], int strand, int dna_len, const int8_t* target, int tlen,
	int d_begin, int d_end, const int8_t* matrix8, int gap_open, int gap_extend, int frame_shift,
	oracle_hsp3* out, uint8_t* transcript, int transcript_cap);

/* gapped filter (gapped_filter.c) */
void oracle_scan_diags(const int8_t* matrix8, const int8_t* query, int qlen, const int8_t* cbs, const int8_t* target,
	int d_begin, int j_begin, int j_end, int band, int* out);
int oracle_diag_alignment(const int* s, int count, int diag_score, int gap_open, int gap_extend);
int oracle_gapped_filter_hit(const int8_t* matrix8, const int8_t* query, int qlen, const int8_t* cbs, const int8_t* target, int slen,
	int hit_i, int hit_j, int band, int window, int diag_score, int gap_open, int gap_extend);
int oracle_gapped_filter_target(const int8_t* matrix8, const int8_t* query, int qlen, const int8_t* cbs, const int8_t* target, int slen,
	const int32_t* hit_i, const int32_t* hit_j, int n_hits, int cutoff1, int cutoff2, int window2, int diag_score, int gap_open, int gap_extend);
void oracle_cutoff_table2d(const oracle_evaluer* e, double evalue, int32_t* table);

/* tantan repeat masking (tantan.c) */
int oracle_tantan_mask(int8_t* seq, int len, const float* lr, float p_repeat, float p_repeat_end, float repeat_growth, float p_mask);
double oracle_tantan_lambda(const int8_t* matrix8);
void oracle_tantan_matrix(const int8_t* matrix8, float* lr);

/* motif soft masking (motif_mask.c) */
int oracle_motif_mask(int8_t* seq, int len, const uint64_t* table, int n_table, int max_motif_len);

/* ---- seed stage (oracle/seed_search.c) ---- */
typedef struct {
	int32_t seedp_bits, index_chunks, hamming_filter_id, n_shapes;
	int32_t shape_len[16], shape_weight[16];
	uint32_t shape_mask[16];
	int32_t shape_pos[16][32];
	int32_t reduction[32], reduction_size;
	int32_t ungapped_window, left_most_interval;     /* config.ungapped_window (48), config.left_most_interval (32) */
	double seed_complexity_cut;
	/* stage-2 ungapped window filter (stage2.h:43-63, 107-113); use_ungapped = 0 when ungapped_evalue == 0 */
	int32_t use_ungapped, short_query_max_len, short_query_cutoff;
	int32_t cutoff_table[32];        /* CutoffTable::data_[bit_length(query_len)], util/scores/cutoff_table.h:26-47 */
	int32_t tile_size, simd_lanes;   /* config.tile_size (1024), int8 lanes of the reference's SIMD build (32 for AVX2) */
	int8_t matrix[32 * 32];          /* ScoreMatrix::matrix8 (= matrix32 values) for ungapped_window */
	int32_t query_translated;        /* align_mode.query_translated: short-frame rules of stage2.h:51,58-63 */
	int32_t cutoff_table_short[32];  /* CutoffTable(ungapped_evalue_short), stage2.h:51 */
	int32_t seed_encoding;           /* 0 = SeedEncoding::SPACED_FACTOR (double-indexed), 1 = HASHED (query-indexed algorithm) */
} oracle_seed_cfg;

typedef struct { uint32_t query; int32_t seed_offset; int64_t subject; int32_t score; int32_t pad; } oracle_hit;

int64_t oracle_seed_search(const oracle_seed_cfg* c, int8_t* qdata, const int64_t* qlimits, int64_t nq,
	const int8_t* tdata, const int64_t* tlimits, int64_t nt, oracle_hit* hits, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
