/* TEST INFRASTRUCTURE ONLY (see oracle.h): CPU restatement of DIAMOND's tantan repeat masking (SURVEY 8f "masking").
 *
 *   Util::tantan::mask (forward-backward over a 50-offset repeat HMM)   src/masking/tantan.cpp:112-215
 *   forward_step / backward_step                                           src/masking/tantan.cpp:45-110
 *   SIMD::sum / scale / hsum (8 float lanes of the AVX2 build)              src/util/simd/vector.h:37-67, vector8_avx2.h:132-139
 *   Masking::Masking (likelihood ratios exp(lambda * score))               src/masking/masking.cpp:134-155
 *   LambdaCalculator (sum(inv(exp(lambda S))) = 1)                         src/masking/lambda.cpp
 *
 * Single-precision results depend on the order of the float operations; this file restates the order of the reference's
 * AVX2 build without FMA (the build flags of oracle/Makefile and of the reference's CMakeLists.txt:240: -mavx2, no
 * -mfma): products and sums are separately rounded, horizontal sums are taken per group of 8 consecutive offsets as
 * ((v0+v4)+(v1+v5)) + ((v2+v6)+(v3+v7)), groups are accumulated in order, offsets 48 and 49 are scalar.
 * Compile without floating-point contraction (-ffp-contract=off, set in oracle/Makefile).
 * Pinned by tests/golden/tantan.tap (tap at Util::tantan::mask: every sequence before/after + the float matrix). */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define WINDOW 50

static float hsum8(const float* v)
{
	const float s0 = v[0] + v[4], s1 = v[1] + v[5], s2 = v[2] + v[6], s3 = v[3] + v[7];
	return (s0 + s1) + (s2 + s3);
}

static float sum50(const float* x)            /* SIMD::sum(f, 50) */
{
	float acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	for (int g = 0; g < 6; ++g) for (int l = 0; l < 8; ++l) acc[l] = acc[l] + x[8 * g + l];
	float s = hsum8(acc);
	s += x[48];
	s += x[49];
	return s;
}

/* lr: 32x32 floats, lr[a*32+b] = likelihood ratio of letters a, b. Returns the number of masked letters; seq is hard-masked
 * in place (letter 23) where the repeat probability >= p_mask (mask_mode 1 of the reference). */
int oracle_tantan_mask(int8_t* seq, int len, const float* lr, float p_repeat, float p_repeat_end, float repeat_growth, float p_mask)
{
	if (len <= 0) return 0;
	float f[WINDOW], d[WINDOW];
	const float b2b = 1.0f - p_repeat, f2f = 1.0f - p_repeat_end;
	const float b2f0 = p_repeat * (1.0f - repeat_growth) / (1.0f - powf(repeat_growth, (float)WINDOW));
	d[WINDOW - 1] = b2f0;
	for (int i = WINDOW - 2; i >= 0; --i) d[i] = d[i + 1] * repeat_growth;
	for (int i = 0; i < WINDOW; ++i) f[i] = 0.0f;
	float* pb = (float*)malloc(sizeof(float) * (size_t)len);
	float* scale = (float*)malloc(sizeof(float) * (size_t)(len / 16 + 1));
	float e[WINDOW];
	float b = 1.0f, f_sum = 0.0f;
	for (int i = 0; i < len; ++i) {
		const int ltr = seq[i] & 31;
		for (int off = 0; off < WINDOW; ++off) e[off] = i - 1 - off >= 0 ? lr[ltr * 32 + (seq[i - 1 - off] & 31)] : 0.0f;
		const float b_old = b;
		float f_sum_new = 0.0f;
		for (int g = 0; g < 6; ++g) {
			float v[8];
			for (int l = 0; l < 8; ++l) {
				const int off = 8 * g + l;
				const float t1 = f[off] * f2f, t2 = b_old * d[off];
				const float tmp = t1 + t2;
				v[l] = tmp * e[off];
				f[off] = v[l];
			}
			f_sum_new += hsum8(v);
		}
		for (int off = 48; off < 50; ++off) {
			const float t1 = f[off] * f2f, t2 = b_old * d[off];
			float vf = t1 + t2;
			vf = vf * e[off];
			f[off] = vf;
			f_sum_new += vf;
		}
		{ const float t1 = b_old * b2b, t2 = f_sum * p_repeat_end; b = t1 + t2; }
		f_sum = f_sum_new;
		if ((i & 15) == 15) {
			const float s = 1.0f / b;
			scale[i / 16] = s;
			b *= s;
			for (int k = 0; k < WINDOW; ++k) f[k] *= s;
			f_sum *= s;
		}
		pb[i] = b;
	}
	float z;
	{ const float t1 = b * b2b, t2 = sum50(f) * p_repeat_end; z = t1 + t2; }
	const float zinv = 1.0f / z;
	b = b2b;
	for (int k = 0; k < WINDOW; ++k) f[k] = p_repeat_end;
	int n_masked = 0;
	int8_t* orig = (int8_t*)malloc((size_t)len);
	memcpy(orig, seq, (size_t)len);
	for (int i = len - 1; i >= 0; --i) {
		float pf;
		{ const float t = pb[i] * b; const float u = t * zinv; pf = 1.0f - u; }
		if ((i & 15) == 15) {
			const float s = scale[i / 16];
			b *= s;
			for (int k = 0; k < WINDOW; ++k) f[k] *= s;
		}
		const int ltr = orig[i] & 31;
		for (int off = 0; off < WINDOW; ++off) e[off] = i - 1 - off >= 0 ? lr[ltr * 32 + (orig[i - 1 - off] & 31)] : 0.0f;
		const float C = p_repeat_end * b;
		float tsum = 0.0f;
		for (int g = 0; g < 6; ++g) {
			float vt[8];
			for (int l = 0; l < 8; ++l) {
				const int off = 8 * g + l;
				float vf = f[off] * e[off];
				vt[l] = vf * d[off];
				const float t1 = vf * f2f;
				f[off] = t1 + C;
			}
			tsum += hsum8(vt);
		}
		for (int off = 48; off < 50; ++off) {
			float vf = f[off] * e[off];
			{ const float t = vf * d[off]; tsum += t; }
			{ const float t1 = vf * f2f, t2 = p_repeat_end * b; f[off] = t1 + t2; }
		}
		{ const float t1 = b2b * b; b = t1 + tsum; }
		if (pf >= p_mask) { seq[i] = 23; ++n_masked; }
	}
	free(pb); free(scale); free(orig);
	return n_masked;
}

/* ---- likelihood-ratio matrix: lambda solves sum(inverse(exp(lambda * S))) = 1 over the 20 standard amino acids ---- */
static int inv_sum(const int* S, int n, double lambda, double* f)
{
	double A[20][40];
	for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { A[i][j] = exp(lambda * S[i * n + j]); A[i][n + j] = i == j ? 1.0 : 0.0; }
	for (int k = 0; k < n; ++k) {                                       /* Gauss-Jordan with partial pivoting */
		int p = k;
		for (int i = k + 1; i < n; ++i) if (fabs(A[i][k]) > fabs(A[p][k])) p = i;
		if (fabs(A[p][k]) < 1e-12) return 0;
		if (p != k) for (int j = 0; j < 2 * n; ++j) { const double t = A[k][j]; A[k][j] = A[p][j]; A[p][j] = t; }
		const double piv = A[k][k];
		for (int j = 0; j < 2 * n; ++j) A[k][j] /= piv;
		for (int i = 0; i < n; ++i) if (i != k) { const double m = A[i][k]; if (m != 0.0) for (int j = 0; j < 2 * n; ++j) A[i][j] -= m * A[k][j]; }
	}
	long double acc = 0;
	for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) acc += A[i][n + j];
	*f = (double)(acc - 1.0L);
	return 1;
}

double oracle_tantan_lambda(const int8_t* matrix8)
{
	int S[400];
	for (int i = 0; i < 20; ++i) for (int j = 0; j < 20; ++j) S[i * 20 + j] = matrix8[i * 32 + j];
	double lo = 1e-3, hi = 1.0, flo, fhi;
	if (!inv_sum(S, 20, lo, &flo) || !inv_sum(S, 20, hi, &fhi) || flo * fhi > 0) return -1.0;
	for (int it = 0; it < 200; ++it) {
		const double mid = 0.5 * (lo + hi);
		double fm;
		if (!inv_sum(S, 20, mid, &fm)) return -1.0;
		if ((fm < 0) == (flo < 0)) { lo = mid; flo = fm; } else { hi = mid; fhi = fm; }
		if (hi - lo < 1e-15) break;
	}
	return 0.5 * (lo + hi);
}

/* Masking::Masking: lr[i][j] = (float)exp(lambda * score(i, j)) for the 26 alphabet letters, 0 elsewhere */
void oracle_tantan_matrix(const int8_t* matrix8, float* lr)
{
	const double lambda = oracle_tantan_lambda(matrix8);
	for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j)
		lr[i * 32 + j] = (i < 26 && j < 26) ? (float)exp(lambda * (double)matrix8[i * 32 + j]) : 0.0f;
}
