// cbs_adjust.h -- composition-based matrix adjustment (--comp-based-stats 2..5), host double precision.
//
// What the reference computes per (query, target) pair in WorkTarget::WorkTarget (/root/reference/src/align/ungapped.cpp:44-58):
//   Stats::adjust_matrix              src/stats/cbs.cpp:94-112        which rule applies (or none)
//   Stats::TargetMatrix::TargetMatrix src/stats/cbs.cpp:114-173       the target's own 26 x 26 integer score table
//   s_TestToApplyREAdjustmentConditional, CompositionMatrixAdjust    src/stats/matrix_adjust.cpp:354-478
//   Blast_OptimizeTargetFrequencies (Newton's method on the KKT system of Yu, Wootton & Altschul 2003) src/stats/blast/ncbi.cpp
//   CompositionBasedStats (lambda rescaling, mode 5)                  src/stats/comp_based_stats.cpp:402-460
// The rounded tables must equal the reference's bit for bit, so every floating-point expression keeps the reference's order
// of operations and the file is compiled with -ffp-contract=off (Makefile); libm's log / exp / acos / sqrt are the same
// functions the reference calls. Pinned on matrices tapped from the reference (tests/test_cbs_adjust.py).
#pragma once
#include <stdint.h>
#include "../../include/diamond_hip.h"

namespace dmnd {

enum { CBS_RULE_NONE = -1, CBS_RULE_SCALE_OLD = 0, CBS_RULE_REL_ENTROPY = 4 };      // EMatrixAdjustRule (stats/cbs.h:41-48): the three the path uses

// what `--comp-based-stats N` switches on (Stats::CBS, stats/cbs.h:112-196)
inline bool cbs_hauser(int mode) { return mode == 1 || mode == 2 || mode == 3; }
inline bool cbs_matrix_adjust(int mode) { return mode >= 2 && mode <= 6; }
inline bool cbs_conditioned(int mode) { return mode == 2 || mode == 3 || mode == 5 || mode == 6; }

// Everything of a scoring matrix that the adjustment reads: ScoreMatrix::joint_probs / background_freqs / freq_ratios /
// ungapped_lambda / ideal_lambda (stats/score_matrix.h:196-210) and the thresholds of Stats::comp_based_stats.
struct CbsModel {
	bool valid = false;
	const double* joint_probs = nullptr;      // 20 x 20
	const double* background = nullptr;       // 20
	const double* freq_ratios = nullptr;      // 28 x 28, NCBIstdaa order
	double ungapped_lambda = 0, ideal_lambda = 0;
	int8_t matrix8[32 * 32];                  // the standard matrix (row = letter, 32 columns)
	int scaled20[20 * 20];                    // its residue block as round(log(frequency ratio) / ungapped lambda * scale): ScoreMatrix::matrix32_scaled_
	// config.cbs_matrix_scale / cbs_err_tolerance / cbs_it_limit / cbs_angle / query_match_distance_threshold / length_ratio_threshold
	int scale = 1, it_limit = 2000;
	double err_tolerance = 0.00000001, angle = 50.0, distance_threshold = -1.0, length_ratio_threshold = -1.0;
};

// model of the standard matrix whose scores are params.matrix8; valid = false for a matrix that is not one of the eight
// (the reference refuses --comp-based-stats > 1 with a custom matrix, basic/config.cpp:837)
void cbs_model_init(CbsModel& m, const dmnd_params& params);

// Stats::composition + count_true_aa (cbs.cpp:52-77): frequencies of the 20 residues among the residues of seq
void cbs_composition(const int8_t* seq, int len, double comp[20], int* true_aa);

// Stats::adjust_matrix: CBS_RULE_* for this pair. query_true_aa = count_true_aa(query context 0)
int cbs_rule(const CbsModel& m, int mode, const double query_comp[20], int query_true_aa, const int8_t* target, int target_len);

// Stats::TargetMatrix::TargetMatrix: out[target letter * 32 + query letter], 32 x 32 with the rows and columns of the
// letters above 25 at -128 like the standard table; rule = what cbs_rule returned (not CBS_RULE_NONE)
void cbs_target_matrix(const CbsModel& m, int rule, const double query_comp[20], int query_true_aa, const int8_t* target, int target_len, int8_t* out);

}  // namespace dmnd
