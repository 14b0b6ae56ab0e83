/* oracle/evalue.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Restatement of the reference's e-value / bit-score arithmetic:
 *   ScoreMatrix::evalue / bitscore            src/stats/score_matrix.cpp:217-220,250-254
 *   alp_params()                              src/stats/score_matrix.cpp:43-47
 *   Sls::AlignmentEvaluer::evalue / area      src/lib/alp/sls_alignment_evaluer.hpp:135-161, .cpp:988-1028
 *   pvalues::get_appr_tail_prob_with_cov_without_errors (area only)   src/lib/alp/sls_pvalues.cpp:367-530
 *   pvalues::compute_tmp_values               src/lib/alp/sls_pvalues.cpp:343-365
 *   sls_basic::normal_probability(x)          src/lib/alp/sls_basic.hpp:191-194
 * The ALP library is vendored in the reference tree (src/lib/alp); the finite-size-correction
 * formula is restated here in double precision with the same operation order.
 * Pinned against the e-values/bit scores of every HSP in tests/golden/swipe_*.tap.
 */
#include <math.h>
#include "oracle.h"

/* gapped (lambda,K,alpha,alpha_v,sigma) + ungapped (alpha_u, alpha_v_u) constants, e.g. BLOSUM62 11/1:
 * src/stats/matrices/blosum62.h rows 0 (ungapped) and {11,1}. */
void oracle_evalue_init(oracle_evaluer* e, double lambda, double K, double alpha, double alpha_v, double sigma,
	double u_alpha, double u_alpha_v, int gap_open, int gap_extend, double db_letters)
{
	const double G = gap_open + gap_extend;
	const double b = 2.0 * G * (u_alpha - alpha), beta = 2.0 * G * (u_alpha_v - alpha_v);
	e->lambda = lambda; e->K = K;
	e->a_I = alpha; e->b_I = b; e->a_J = alpha; e->b_J = b;
	e->alpha_I = alpha_v; e->beta_I = beta; e->alpha_J = alpha_v; e->beta_J = beta;
	e->sigma = sigma; e->tau = 2.0 * G * (u_alpha_v - sigma);
	const double nat_cut_off_in_max = 2.0;
	e->vi_y_thr = fmax(nat_cut_off_in_max * e->alpha_I / lambda, 0.0);
	e->vj_y_thr = fmax(nat_cut_off_in_max * e->alpha_J / lambda, 0.0);
	e->c_y_thr = fmax(nat_cut_off_in_max * e->sigma / lambda, 0.0);
	e->db_letters = db_letters;
	e->ln_k = log(K);
}

static double normal_probability(double x) { return 0.5 * erfc(-0.70710678118654752440 * x); }

double oracle_area(const oracle_evaluer* e, double y, double seqlen1, double seqlen2)
{
	const double const_val = 1.0 / sqrt(2.0 * 3.1415926535897932384626433832795);
	const double m_ = seqlen2, n_ = seqlen1;
	const double m_li_y = m_ - (e->a_I * y + e->b_I);
	const double vi_y = fmax(e->vi_y_thr, e->alpha_I * y + e->beta_I);
	const double sqrt_vi_y = sqrt(vi_y);
	const double m_F = sqrt_vi_y == 0.0 ? 1e100 : m_li_y / sqrt_vi_y;
	const double P_m_F = normal_probability(m_F);
	const double E_m_F = -const_val * exp(-0.5 * m_F * m_F);
	const double p1 = m_li_y * P_m_F - sqrt_vi_y * E_m_F;
	const double n_lj_y = n_ - (e->a_J * y + e->b_J);
	const double vj_y = fmax(e->vj_y_thr, e->alpha_J * y + e->beta_J);
	const double sqrt_vj_y = sqrt(vj_y);
	const double n_F = sqrt_vj_y == 0.0 ? 1e100 : n_lj_y / sqrt_vj_y;
	const double P_n_F = normal_probability(n_F);
	const double E_n_F = -const_val * exp(-0.5 * n_F * n_F);
	const double p2 = n_lj_y * P_n_F - sqrt_vj_y * E_n_F;
	const double c_y = fmax(e->c_y_thr, e->sigma * y + e->tau);
	return p1 * p2 + c_y * (P_m_F * P_n_F);
}

double oracle_evalue(const oracle_evaluer* e, int raw_score, unsigned query_len, unsigned subject_len)
{
	const double s = (double)raw_score;
	return oracle_area(e, s, query_len, subject_len) * (e->K * exp(-e->lambda * s)) * e->db_letters / (double)subject_len;
}

double oracle_bitscore(const oracle_evaluer* e, double raw_score)
{
	return (e->lambda * round(raw_score) - e->ln_k) / 0.69314718055994530941723212145818;
}
