// cbs_adjust.cpp -- composition-based matrix adjustment on the host (see cbs_adjust.h for what it replaces).
//
// Formulation of this file: everything is sized at compile time for the 20-letter problem (400 target frequencies, 39 linear
// constraints + the relative-entropy constraint), lives on the stack of the calling thread and is addressed through flat
// arrays -- no allocation per (query, target) pair, which is what the extension stage's host threads call this for, tens of
// thousands of times per block. The arithmetic is the published algorithm (Yu, Wootton & Altschul 2003; Altschul et al. 2005:
// minimise the Kullback-Leibler distance to the standard target frequencies under the marginal constraints of both sequences'
// compositions and a fixed relative entropy, by Newton's method on the block-reduced KKT system) in the operation order of the
// reference, because the rounded integer tables have to come out identical.
#include <climits>
#include <cmath>
#include <cstring>
#include <algorithm>
#include "cbs_adjust.h"
#include "cbs_tables.h"
#include "score_matrices.h"

namespace dmnd {

namespace {

enum { N = 20, NN = N * N, MA = 2 * N - 1, M = 2 * N, STD = 26, X_LETTER = 23 };
const double PSEUDOCOUNTS = 20.0;              // kReMatrixAdjustmentPseudocounts
const double FIXED_RE = 0.44;                  // kFixedReBlosum62: the relative entropy every adjusted matrix is held to
const double MAX_X_SCORE = -1.0;               // kMaximumXscore
const double SCORE_MIN = -128.0;               // COMPO_SCORE_MIN

// ---- the constraint matrix A (39 x 400, never formed): row j < 20 sums column j of x, row 19 + i sums row i >= 1 -----------

// y += alpha * A x                                              (MultiplyByA, ncbi.cpp:153-180, after its beta handling)
void add_A(double* y, double alpha, const double* x)
{
	for (int i = 0; i < N; ++i)
		for (int j = 0; j < N; ++j)
			y[j] += alpha * x[i * N + j];
	for (int i = 1; i < N; ++i)
		for (int j = 0; j < N; ++j)
			y[i + N - 1] += alpha * x[i * N + j];
}

// y += A^T z                                                    (MultiplyByAtranspose with alpha = beta = 1, ncbi.cpp:196-227)
void add_At(double* y, const double* z)
{
	for (int i = 0; i < N; ++i)
		for (int j = 0; j < N; ++j) {
			const int k = i * N + j;
			y[k] += 1.0 * z[j];
			if (i > 0) y[k] += 1.0 * z[i + N - 1];
		}
}

// scaled overflow-safe 2-norm                                   (Nlm_EuclideanNorm, linear_algebra_ncbi.h:190-208)
double norm2(const double* v, int n)
{
	double sum = 1.0, scale = 0.0;
	for (int i = 0; i < n; ++i) {
		if (v[i] != 0.0) {
			const double a = std::fabs(v[i]);
			if (scale < a) { sum = 1.0 + sum * (scale / a) * (scale / a); scale = a; }
			else sum += (a / scale) * (a / scale);
		}
	}
	return scale * std::sqrt(sum);
}

// The Newton system of one iteration, block-reduced: D = diag(x) / (1 - eta) inverted, and the Cholesky factor of J D^-1 J^T
// (40 x 40, lower triangle) with J = [A; grad of the relative entropy]          (ReNewtonSystem, ncbi.cpp:330-352)
struct Kkt {
	double L[M][M];
	double Dinv[NN];
	double grad_re[NN];
};

// (FactorReNewtonSystem, ncbi.cpp:440-510; ScaledSymmetricProductA :106-135; Nlm_FactorLtriangPosDef)
void kkt_factor(Kkt& s, const double* x, const double* z, const double* grad1, double* work)
{
	const double eta = z[M - 1];
	for (int i = 0; i < NN; ++i) s.Dinv[i] = x[i] / (1 - eta);
	for (int r = 0; r < MA; ++r)
		for (int c = 0; c <= r; ++c) s.L[r][c] = 0.0;
	for (int i = 0; i < N; ++i)
		for (int j = 0; j < N; ++j) {
			const double dd = s.Dinv[i * N + j];
			s.L[j][j] += dd;
			if (i > 0) { s.L[i + N - 1][j] += dd; s.L[i + N - 1][i + N - 1] += dd; }
		}
	std::memcpy(s.grad_re, grad1, sizeof s.grad_re);
	s.L[M - 1][M - 1] = 0.0;
	for (int i = 0; i < NN; ++i) {
		work[i] = s.Dinv[i] * s.grad_re[i];
		s.L[M - 1][M - 1] += s.grad_re[i] * work[i];
	}
	for (int i = 0; i < MA; ++i) s.L[M - 1][i] = 0.0;
	add_A(s.L[M - 1], 1.0, work);
	for (int i = 0; i < M; ++i) {
		for (int j = 0; j < i; ++j) {
			double t = s.L[i][j];
			for (int k = 0; k < j; ++k) t -= s.L[i][k] * s.L[j][k];
			s.L[i][j] = t / s.L[j][j];
		}
		double t = s.L[i][i];
		for (int k = 0; k < i; ++k) t -= s.L[i][k] * s.L[i][k];
		s.L[i][i] = std::sqrt(t);
	}
}

// in: dual residuals x, primal residuals z; out: the Newton step in both     (SolveReNewtonSystem, ncbi.cpp:526-579)
void kkt_solve(const Kkt& s, double* x, double* z, double* work)
{
	for (int i = 0; i < NN; ++i) work[i] = x[i] * s.Dinv[i];
	add_A(z, -1.0, work);
	for (int i = 0; i < NN; ++i) z[M - 1] -= s.grad_re[i] * work[i];
	for (int i = 0; i < M; ++i) {                     // L y = b, then L^T z = y (Nlm_SolveLtriangPosDef)
		double t = z[i];
		for (int j = 0; j < i; ++j) t -= s.L[i][j] * z[j];
		z[i] = t / s.L[i][i];
	}
	for (int j = M - 1; j >= 0; --j) {
		z[j] /= s.L[j][j];
		for (int i = 0; i < j; ++i) z[i] -= s.L[j][i] * z[j];
	}
	for (int i = 0; i < NN; ++i) x[i] += s.grad_re[i] * z[M - 1];
	add_At(x, z);
	for (int i = 0; i < NN; ++i) x[i] *= s.Dinv[i];
}

// Target frequencies x closest (in relative entropy) to q whose row sums are row[], column sums col[] and whose own relative
// entropy is `re`. Returns 0 = converged to a minimiser, 1 = not.          (Blast_OptimizeTargetFrequencies, ncbi.cpp:660-790)
int optimise_target_freqs(double* x, const double* q, const double* row, const double* col, double re, double tol, int maxits)
{
	double old_scores[NN], grad0[NN], grad1[NN], rx[NN], work[NN], z[M], rz[M], rnorm = 0.0;
	Kkt kkt;
	for (int i = 0; i < N; ++i)
		for (int j = 0; j < N; ++j)
			old_scores[i * N + j] = std::log(q[i * N + j] / (row[i] * col[j]));
	std::memcpy(x, q, NN * sizeof(double));
	for (int i = 0; i < M; ++i) z[i] = 0.0;
	int its = 0;
	while (its <= maxits) {
		// objective, relative entropy and their gradients at x                (EvaluateReFunctions)
		double v0 = 0.0, v1 = 0.0;
		for (int k = 0; k < NN; ++k) {
			double t = std::log(x[k] / q[k]);
			v0 += x[k] * t;
			grad0[k] = t + 1;
			t += old_scores[k];
			v1 += x[k] * t;
			grad1[k] = t + 1;
		}
		(void)v0;
		// residuals of the optimality conditions                               (CalculateResiduals)
		const double eta = z[M - 1];
		for (int i = 0; i < NN; ++i) rx[i] = -grad0[i] + eta * grad1[i];
		add_At(rx, z);
		const double norm_x = norm2(rx, NN);
		for (int i = 0; i < N; ++i) rz[i] = col[i];
		for (int i = 1; i < N; ++i) rz[i + N - 1] = row[i];
		add_A(rz, -1.0, x);
		rz[M - 1] = re - v1;
		const double norm_z = norm2(rz, M);
		rnorm = std::sqrt(norm_x * norm_x + norm_z * norm_z);
		if (!(rnorm > tol)) break;                    // also leaves on NaN
		if (++its <= maxits) {
			kkt_factor(kkt, x, z, grad1, work);
			kkt_solve(kkt, rx, rz, work);
			double alpha = 1.0 / .95;                 // the longest step that keeps x positive (Nlm_StepBound), damped
			for (int i = 0; i < NN; ++i) {
				const double a = -x[i] / rx[i];
				if (a >= 0 && a < alpha) alpha = a;
			}
			alpha *= 0.95;
			for (int i = 0; i < NN; ++i) x[i] += alpha * rx[i];
			for (int i = 0; i < M; ++i) z[i] += alpha * rz[i];
		}
	}
	return (its <= maxits && rnorm <= tol && z[M - 1] < 1) ? 0 : 1;
}

// probs <- mix of the observed frequencies with the background, weight 20 / (observations + 20)   (Blast_ApplyPseudocounts)
void apply_pseudocounts(double* probs, int observations, const double* background)
{
	double sum = 0.0;
	for (int i = 0; i < N; ++i) sum += probs[i];
	if (sum == 0.0) sum = 1.0;
	const double weight = PSEUDOCOUNTS / (observations + PSEUDOCOUNTS);
	for (int i = 0; i < N; ++i)
		probs[i] = (1.0 - weight) * probs[i] / sum + weight * background[i];
}

// scores of the mask letter against everything: the expected score of the row / column, at most -1   (s_SetXUOScores,
// comp_based_stats.cpp:330-352). S is 26 x 26 with the 20 x 20 residue block filled in.
void set_x_scores(double (*S)[STD], const double* row_probs, const double* col_probs)
{
	double score_xx = 0.0;
	for (int i = 0; i < N; ++i) {
		double avg = 0.0;
		for (int j = 0; j < N; ++j) avg += S[i][j] * col_probs[j];
		S[i][X_LETTER] = std::min(avg, MAX_X_SCORE);
		score_xx += avg * row_probs[i];
		double c = 0.0;
		for (int j = 0; j < N; ++j) c += S[j][i] * row_probs[j];
		S[X_LETTER][i] = std::min(c, MAX_X_SCORE);
	}
	S[X_LETTER][X_LETTER] = std::min(score_xx, MAX_X_SCORE);
}

int round_score(double f) { return f < INT_MIN ? INT_MIN : (int)std::round(f); }

// entries [i][j], i and j residues or the mask letter, of the integer table the reference hands to TargetMatrix
// (array `s`, [query letter][target letter]); the other entries are never read there
struct IntTable { int v[STD][STD]; };

// target frequencies -> rounded scores on scale lambda          (s_ScoresStdAlphabet, matrix_adjust.cpp:170-208)
void scores_from_freqs(IntTable& out, const double* freq, const double* row_prob, const double* col_prob, double lambda)
{
	double S[STD][STD];
	double sum = 0.0;
	for (int a = 0; a < N; ++a)
		for (int b = 0; b < N; ++b) sum += freq[a * N + b];
	for (int a = 0; a < STD; ++a)
		for (int b = 0; b < STD; ++b) S[a][b] = a < N && b < N ? freq[a * N + b] / sum : 0.0;
	for (int i = 0; i < N; ++i)
		if (row_prob[i] > 0)
			for (int j = 0; j < N; ++j)
				if (col_prob[j] > 0) S[i][j] /= (row_prob[i] * col_prob[j]);
	for (int i = 0; i < STD; ++i)
		for (int j = 0; j < STD; ++j) S[i][j] = 0.0 == S[i][j] ? SCORE_MIN : std::log(S[i][j]) / lambda;
	set_x_scores(S, row_prob, col_prob);
	for (int i = 0; i < STD; ++i)
		for (int j = 0; j < STD; ++j) out.v[i][j] = round_score(S[i][j]);
}

// CompositionMatrixAdjust (matrix_adjust.cpp:455-476) = Blast_CompositionMatrixAdj with the fixed relative entropy
bool relative_entropy_adjust(const CbsModel& m, IntTable& out, int query_len, int target_len, const double* query_comp, const double* target_comp)
{
	double row[N], col[N], x[NN];
	std::copy(query_comp, query_comp + N, row);
	std::copy(target_comp, target_comp + N, col);
	apply_pseudocounts(row, query_len, m.background);
	apply_pseudocounts(col, target_len, m.background);
	if (optimise_target_freqs(x, m.joint_probs, row, col, FIXED_RE, m.err_tolerance, m.it_limit) != 0) return false;
	scores_from_freqs(out, x, row, col, m.ideal_lambda / m.scale);
	return true;
}

// ---- Karlin-Altschul lambda of a score distribution (mode 5 and ScoreMatrix::ideal_lambda) ---------------------------------

int gcd(int a, int b)
{
	b = std::abs(b);
	if (b > a) std::swap(a, b);
	while (b != 0) { const int c = a % b; a = b; b = c; }
	return a;
}

// root in (0, 1) of sum_s p(s) x^-s = 1 by safeguarded Newton iterations on the polynomial in x = exp(-lambda);
// p = probabilities centred on score 0, scores low .. high             (NlmKarlinLambdaNR, comp_based_stats.cpp:96-176)
double karlin_lambda_nr(const double* p, int d, int low, int high, double lambda0, double tolx, int itmax, int max_newton)
{
	double x0 = std::exp(-lambda0), x = (0 < x0 && x0 < 1) ? x0 : .5, a = 0, b = 1, f = 4;
	bool is_newton = false;
	for (int k = 0; k < itmax; ++k) {
		const double fold = f;
		const bool was_newton = is_newton;
		is_newton = false;
		double g = 0;
		f = p[low];
		int i;
		for (i = low + d; i < 0; i += d) { g = x * g + f; f = f * x + p[i]; }
		g = x * g + f;
		f = f * x + p[0] - 1;
		for (i = d; i <= high; i += d) { g = x * g + f; f = f * x + p[i]; }
		if (f > 0) a = x;
		else if (f < 0) b = x;
		else break;
		if (b - a < 2 * a * (1 - b) * tolx) { x = (a + b) / 2; break; }
		if (k >= max_newton || (was_newton && std::fabs(f) > .9 * std::fabs(fold)) || g >= 0) x = (a + b) / 2;
		else {
			const double step = -f / g, y = x + step;
			if (y <= a || y >= b) x = (a + b) / 2;
			else {
				is_newton = true;
				x = y;
				if (std::fabs(step) < tolx * x * (1 - x)) break;
			}
		}
	}
	return -std::log(x) / d;
}

// distribution of the scores of the 20 x 20 block of `scores` (row stride `stride`) under the two letter distributions, then
// its lambda; < 0: the expected score is not negative   (s_GetMatrixScoreProbs + s_CalcLambda + Blast_KarlinLambdaNR)
template<typename T>
double lambda_of(const T* scores, int stride, const double* col_prob, const double* row_prob, double lambda0)
{
	int lo = 0, hi = 0;
	for (int r = 0; r < N; ++r)
		for (int c = 0; c < N; ++c) { const int s = scores[r * stride + c]; if (s < lo) lo = s; if (s > hi) hi = s; }
	double prob[512] = { 0 };                        // int8 scores: the range never exceeds 256
	double* centred = prob - lo;
	for (int r = 0; r < N; ++r)
		for (int c = 0; c < N; ++c) centred[scores[r * stride + c]] += (row_prob[r] * col_prob[c]);
	double avg = 0.0;
	for (int i = 0; i < hi - lo + 1; ++i) avg += (lo + i) * prob[i];
	if (avg >= 0.) return -1.0;
	int d = -lo;
	for (int i = 1; i <= hi - lo && d > 1; ++i)
		if (centred[i + lo] != 0.0) d = gcd(d, i);
	return karlin_lambda_nr(centred, d, lo, hi, lambda0, 1.e-5, 20, 20 + 17);
}

// amino acid background frequencies of Robinson & Robinson (1991), per thousand, in the letter order ARNDCQEGHILKMFPSTWYV
// (ideal_lambda, comp_based_stats.cpp:473-523, sums them in alphabetical order of the one-letter codes)
const int ROBINSON_ORDER[N] = { 0, 4, 3, 6, 13, 7, 8, 9, 11, 10, 12, 2, 14, 5, 1, 15, 16, 19, 17, 18 };      // A C D E F G H I K L M N P Q R S T V W Y
const double ROBINSON[N] = { 78.05, 19.25, 53.64, 62.95, 38.56, 73.77, 21.99, 51.42, 57.44, 90.19, 22.43, 44.87, 52.03, 42.64, 51.29, 71.20, 58.41, 64.41, 13.30, 32.16 };

const int TO_NCBISTDAA[N] = { 1, 16, 13, 4, 3, 15, 5, 7, 8, 9, 11, 10, 12, 6, 14, 17, 18, 20, 22, 19 };      // ALPH_TO_NCBI

// Blast_CompositionBasedStats (comp_based_stats.cpp:402-451): the standard matrix rescaled to the lambda its scores have
// under the two compositions, ratio clamped to [0.5, 1]
bool lambda_rescale(const CbsModel& m, IntTable& out, const double* query_comp, const double* target_comp)
{
	const double ungapped = m.ungapped_lambda / m.scale;
	const double correct = lambda_of(m.scaled20, N, target_comp, query_comp, ungapped);
	if (correct < 0.0) return false;
	double ratio = correct / ungapped;
	ratio = std::min(1.0, ratio);
	ratio = std::max(ratio, 0.5);
	const double scaled = ungapped / ratio;
	double S[STD][STD];
	for (int i = 0; i < N; ++i)
		for (int j = 0; j < N; ++j) {
			const double r = m.freq_ratios[TO_NCBISTDAA[i] * 28 + TO_NCBISTDAA[j]];
			S[i][j] = 0.0 == r ? SCORE_MIN : std::log(r) / scaled;
		}
	set_x_scores(S, query_comp, target_comp);
	for (int i = 0; i < STD; ++i)
		for (int j = 0; j < STD; ++j)
			out.v[i][j] = (i < N || i == X_LETTER) && (j < N || j == X_LETTER) ? round_score(S[i][j]) : 0;
	return true;
}

// relative entropy "distance" of two compositions               (Blast_GetRelativeEntropy, matrix_adjust.cpp:322-343)
double composition_distance(const double* A, const double* B)
{
	double value = 0.0;
	for (int i = 0; i < N; ++i) {
		const double t = (A[i] + B[i]) / 2;
		if (t > 0) {
			if (A[i] > 0) value += A[i] * std::log(A[i] / t) / 2;
			if (B[i] > 0) value += B[i] * std::log(B[i] / t) / 2;
		}
	}
	if (value < 0) value = 0;
	return std::sqrt(value);
}

// more than 50 letters, the two most frequent of them above 40 %     (s_HighPairFrequencies)
bool high_pair(const double* p, int length)
{
	if (length <= 50) return false;
	double max = 0, second = 0;
	for (int i = 0; i < N; ++i)
		if (p[i] > second) {
			second = p[i];
			if (p[i] > max) { second = max; max = p[i]; }
		}
	return (max + second) > 0.4;
}

}  // namespace

void cbs_model_init(CbsModel& m, const dmnd_params& params)
{
	m = CbsModel();
	std::memcpy(m.matrix8, params.matrix8, sizeof m.matrix8);
	for (int t = 0; t < N_STANDARD_MATRICES; ++t) {
		const StandardMatrixTable& s = STANDARD_MATRICES[t];
		bool same = true;
		for (int i = 0; i < 26 && same; ++i)
			for (int j = 0; j < 26 && same; ++j) {
				const char ch = s.scores[i * 26 + j];
				same = params.matrix8[i * 32 + j] == (int8_t)(ch >= 'a' ? ch - 'a' : -(ch - 'A' + 1));
			}
		if (!same) continue;
		for (int k = 0; k < N_CBS_TABLES; ++k)
			if (std::strcmp(CBS_TABLES[k].name, s.name) == 0) {
				m.joint_probs = CBS_TABLES[k].joint_probs; m.background = CBS_TABLES[k].background; m.freq_ratios = CBS_TABLES[k].freq_ratios;
			}
		if (!m.joint_probs) return;
		m.ungapped_lambda = s.rows[0].lambda;
		// ScoreMatrix::matrix32_scaled_ (score_matrix.cpp:59,194-205): the residue block re-derived from the frequency ratios
		for (int i = 0; i < N; ++i)
			for (int j = 0; j < N; ++j)
				m.scaled20[i * N + j] = (int)std::round(std::log(m.freq_ratios[TO_NCBISTDAA[i] * 28 + TO_NCBISTDAA[j]]) / m.ungapped_lambda * m.scale);
		// ScoreMatrix::ideal_lambda_ (score_matrix.cpp:61): lambda of the matrix under the Robinson & Robinson frequencies
		double bg[N], sum = 0.0;
		for (int i = 0; i < N; ++i) { bg[ROBINSON_ORDER[i]] = ROBINSON[i]; sum += ROBINSON[i]; }
		for (int i = 0; i < N; ++i) bg[i] /= sum;
		m.ideal_lambda = lambda_of(m.matrix8, 32, bg, bg, 0.5);
		m.valid = m.ideal_lambda > 0.0;
		return;
	}
}

void cbs_composition(const int8_t* seq, int len, double comp[20], int* true_aa)
{
	for (int i = 0; i < N; ++i) comp[i] = 0.0;
	int n = 0;
	for (int i = 0; i < len; ++i) {
		const int l = seq[i] & 31;
		if (l < N) { ++comp[l]; ++n; }
	}
	if (true_aa) *true_aa = n;
	if (n == 0) return;
	for (int i = 0; i < N; ++i) comp[i] /= n;
}

int cbs_rule(const CbsModel& m, int mode, const double query_comp[20], int query_true_aa, const int8_t* target, int target_len)
{
	if (!cbs_matrix_adjust(mode) || target_len == 0 || query_true_aa == 0) return CBS_RULE_NONE;
	if (!cbs_conditioned(mode)) return CBS_RULE_REL_ENTROPY;
	// s_TestToApplyREAdjustmentConditional (matrix_adjust.cpp:354-447): the lambda rescaling for pairs whose compositions deviate
	// from the background in different directions (angle above the threshold), the full adjustment otherwise
	double tc[N];
	cbs_composition(target, target_len, tc, nullptr);
	const double d_m_mat = composition_distance(tc, m.background), d_q_mat = composition_distance(query_comp, m.background),
		d_m_q = composition_distance(tc, query_comp);
	double angle = std::acos((d_m_mat * d_m_mat + d_q_mat * d_q_mat - d_m_q * d_m_q) / 2.0 / d_m_mat / d_q_mat);
	angle = angle * 180 / 3.1415926543;
	const double len_q = 1.0 * query_true_aa, len_m = 1.0 * target_len;
	const double len_large = len_q > len_m ? len_q : len_m, len_small = len_q > len_m ? len_m : len_q;
	int rule;
	if (high_pair(query_comp, query_true_aa) || high_pair(tc, target_len)) rule = CBS_RULE_REL_ENTROPY;
	else if (d_m_q > m.distance_threshold && len_large / len_small > m.length_ratio_threshold && angle > m.angle) rule = CBS_RULE_SCALE_OLD;
	else rule = CBS_RULE_REL_ENTROPY;
	if (mode == 5) return rule;
	return rule == CBS_RULE_REL_ENTROPY ? CBS_RULE_REL_ENTROPY : CBS_RULE_NONE;
}

void cbs_target_matrix(const CbsModel& m, int rule, const double query_comp[20], int query_true_aa, const int8_t* target, int target_len, int8_t* out)
{
	double tc[N];
	int target_true_aa = 0;
	cbs_composition(target, target_len, tc, &target_true_aa);
	IntTable t;
	bool ok = false;
	if (rule == CBS_RULE_SCALE_OLD) ok = lambda_rescale(m, t, query_comp, tc);
	if (!ok) ok = relative_entropy_adjust(m, t, query_true_aa, target_true_aa, query_comp, tc);
	if (!ok)                                          // the optimisation did not converge: the standard scores (matrix_adjust.cpp:470-474)
		for (int i = 0; i < STD; ++i)
			for (int j = 0; j < STD; ++j) t.v[i][j] = m.matrix8[i * 32 + j] * m.scale;
	// TargetMatrix::TargetMatrix (cbs.cpp:151-165): residues and the mask letter take the adjusted score -- stored as int8 without
	// a clamp, exactly as the reference's assignment converts it --, every other letter keeps the standard score
	for (int i = 0; i < 32; ++i)
		for (int j = 0; j < 32; ++j) {
			int8_t v = -128;
			if (i < STD && j < STD) {
				if ((i < N || i == X_LETTER) && (j < N || j == X_LETTER)) v = (int8_t)t.v[j][i];
				else v = (int8_t)std::max(m.matrix8[i * 32 + j] * m.scale, (int)SCHAR_MIN);
			}
			out[i * 32 + j] = v;
		}
}

}  // namespace dmnd

// ---- C ABI (host only: no device is touched) -----------------------------------------------------------------------------------
using namespace dmnd;

extern "C" int dmnd_cbs_composition(const int8_t* seq, int32_t len, double* comp, int32_t* true_aa)
{
	if (!seq || len < 0 || !comp) return DMND_E_ARG;
	int n = 0;
	cbs_composition(seq, len, comp, &n);
	if (true_aa) *true_aa = n;
	return DMND_OK;
}

extern "C" int dmnd_cbs_rule(const dmnd_params* params, int mode, const double* query_comp, int32_t query_true_aa, const int8_t* target, int32_t target_len, int32_t* rule)
{
	if (!params || !query_comp || !target || !rule) return DMND_E_ARG;
	CbsModel m;
	cbs_model_init(m, *params);
	if (!m.valid) return DMND_E_ARG;
	*rule = cbs_rule(m, mode, query_comp, query_true_aa, target, target_len);
	return DMND_OK;
}

extern "C" int dmnd_cbs_target_matrix(const dmnd_params* params, int rule, const double* query_comp, int32_t query_true_aa, const int8_t* target, int32_t target_len, int8_t* matrix_out)
{
	if (!params || !query_comp || !target || !matrix_out || (rule != CBS_RULE_SCALE_OLD && rule != CBS_RULE_REL_ENTROPY)) return DMND_E_ARG;
	CbsModel m;
	cbs_model_init(m, *params);
	if (!m.valid) return DMND_E_ARG;
	cbs_target_matrix(m, rule, query_comp, query_true_aa, target, target_len, matrix_out);
	return DMND_OK;
}

extern "C" double dmnd_cbs_ideal_lambda(const dmnd_params* params)
{
	if (!params) return -1.0;
	CbsModel m;
	cbs_model_init(m, *params);
	return m.valid ? m.ideal_lambda : -1.0;
}
