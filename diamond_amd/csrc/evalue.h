// evalue.h -- host-side (double precision) alignment statistics of the product path.
// Computes what ScoreMatrix::evalue / bitscore compute in the reference
// (/root/reference/src/stats/score_matrix.cpp:43-47,217-220,250-254), i.e. the ALP finite-size
// corrected e-value (vendored src/lib/alp: sls_alignment_evaluer.hpp:135-161,
// sls_pvalues.cpp:343-365 thresholds, :367-530 area) -- never evaluated on the GPU.
#pragma once
#include <cmath>
#include "../../include/diamond_hip.h"

namespace dmnd {

struct Evaluer {
	double lambda, K, ln_k, db_letters;
	double a, b, alpha, beta, sigma, tau;     // symmetric: the I and J parameter sets are equal
	double v_thr, c_thr;

	void init(const dmnd_params& p)
	{
		const double G = p.gap_open + p.gap_extend;
		lambda = p.lambda; K = p.K; ln_k = std::log(K); db_letters = p.db_letters;
		a = p.alpha;
		b = 2.0 * G * (p.u_alpha - p.alpha);
		alpha = p.alpha_v;
		beta = 2.0 * G * (p.u_alpha_v - p.alpha_v);
		sigma = p.sigma;
		tau = 2.0 * G * (p.u_alpha_v - p.sigma);
		v_thr = std::fmax(2.0 * alpha / lambda, 0.0);
		c_thr = std::fmax(2.0 * sigma / lambda, 0.0);
	}

	static double normal_cdf(double x) { return 0.5 * std::erfc(-0.70710678118654752440 * x); }

	double area(double y, double len1, double len2) const
	{
		const double inv_sqrt_2pi = 1.0 / std::sqrt(2.0 * 3.1415926535897932384626433832795);
		const double m_l = len2 - (a * y + b), n_l = len1 - (a * y + b);
		const double v = std::fmax(v_thr, alpha * y + beta), sv = std::sqrt(v);
		const double mF = sv == 0.0 ? 1e100 : m_l / sv, nF = sv == 0.0 ? 1e100 : n_l / sv;
		const double PmF = normal_cdf(mF), PnF = normal_cdf(nF);
		const double EmF = -inv_sqrt_2pi * std::exp(-0.5 * mF * mF), EnF = -inv_sqrt_2pi * std::exp(-0.5 * nF * nF);
		const double p1 = m_l * PmF - sv * EmF, p2 = n_l * PnF - sv * EnF;
		const double c = std::fmax(c_thr, sigma * y + tau);
		return p1 * p2 + c * (PmF * PnF);
	}

	double evalue(int raw_score, unsigned qlen, unsigned slen) const
	{
		const double s = (double)raw_score;
		return area(s, qlen, slen) * (K * std::exp(-lambda * s)) * db_letters / (double)slen;
	}

	double bitscore(double raw_score) const { return (lambda * std::round(raw_score) - ln_k) / 0.69314718055994530941723212145818; }
};

}  // namespace dmnd
