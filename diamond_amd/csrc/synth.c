/* Deterministic synthetic protein workload generator.
 *
 * PRNG: xoshiro256** seeded through splitmix64. Residues i.i.d. from the Robinson-Robinson
 * background frequencies (the BLOSUM62 background the reference uses for its statistics,
 * /root/reference/src/stats/matrices/blosum62.h background_freqs), lengths
 * clamp(round(N(300,80^2)),50,2000). A database is F families x M members: the ancestor is random,
 * each member is the ancestor with a per-sequence substitution rate drawn U(sub_lo,sub_hi) and
 * indels (rate indel_rate per site, geometric length p=0.5). Queries are fresh mutants of uniformly
 * chosen ancestors (rate U(q_lo,q_hi)) with a fraction of pure-random decoys.
 *
 * q_family[q] = index of the family a query was derived from (its members are database sequences
 * family*members .. family*members+members-1), or -1 for a decoy.
 * Output is letters in the reference's amino-acid code (0..19 = "ARNDCQEGHILKMFPSTWYV",
 * src/basic/value.h:53) in one flat buffer + offsets, or FASTA text for the reference CLI.
 * Plain C, no dependencies: used by bench.py / tests through ctypes and by the C++ host driver.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>

typedef struct { uint64_t s[4]; } rng_t;

static uint64_t splitmix64(uint64_t* x) {
	uint64_t z = (*x += 0x9e3779b97f4a7c15ULL);
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
	return z ^ (z >> 31);
}
static void rng_seed(rng_t* r, uint64_t seed) { for (int i = 0; i < 4; ++i) r->s[i] = splitmix64(&seed); }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng_t* r) {
	uint64_t* s = r->s;
	const uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
	s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
	return result;
}
static inline double rng_u(rng_t* r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static double rng_normal(rng_t* r) {
	double u1 = rng_u(r), u2 = rng_u(r);
	if (u1 < 1e-300) u1 = 1e-300;
	return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

/* Robinson & Robinson (1991) amino-acid frequencies, order ARNDCQEGHILKMFPSTWYV */
static const double BG[20] = { 0.07805, 0.05129, 0.04487, 0.05364, 0.01925, 0.04264, 0.06295, 0.07377, 0.02199, 0.05142,
	0.09019, 0.05744, 0.02243, 0.03856, 0.05203, 0.07120, 0.05841, 0.01330, 0.03216, 0.06441 };
static uint16_t cdf16[20];
static void init_cdf(void) {
	double c = 0;
	for (int i = 0; i < 20; ++i) { c += BG[i]; cdf16[i] = (uint16_t)(c * 65535.0 + 0.5); }
	cdf16[19] = 65535;
}
static inline int8_t rand_letter(rng_t* r) {
	const uint16_t x = (uint16_t)(rng_next(r) >> 48);
	int i = 0;
	while (x > cdf16[i]) ++i;
	return (int8_t)i;
}
static int rand_len(rng_t* r, double mean, double sd, int lo, int hi) {
	long l = lround(mean + sd * rng_normal(r));
	if (l < lo) l = lo;
	if (l > hi) l = hi;
	return (int)l;
}
static int geometric(rng_t* r) { int n = 1; while (rng_u(r) < 0.5) ++n; return n; }

/* out must hold at least 2*n+64 letters; returns mutant length (>= 1) */
static int mutate(rng_t* r, const int8_t* anc, int n, double sub, double indel, int8_t* out, int cap) {
	int m = 0;
	for (int i = 0; i < n && m < cap - 40; ++i) {
		const double u = rng_u(r);
		if (u < indel * 0.5) {              /* deletion of a geometric run */
			i += geometric(r) - 1;
			continue;
		}
		if (u < indel) {                    /* insertion before this site */
			int k = geometric(r);
			while (k-- > 0 && m < cap - 40) out[m++] = rand_letter(r);
		}
		out[m++] = rng_u(r) < sub ? rand_letter(r) : anc[i];
	}
	if (m == 0) out[m++] = rand_letter(r);
	return m;
}

typedef struct {
	uint64_t seed;
	int64_t families; int members;            /* database = families * members sequences */
	int64_t queries; double decoy_frac;
	double len_mean, len_sd; int len_min, len_max;
	double sub_lo, sub_hi, q_lo, q_hi, indel;
} synth_cfg;

/* Generates database and queries into caller-visible malloc'ed buffers.
 * db_data/q_data: letters back to back (no delimiters); *_off: n+1 int64 offsets. Free with synth_free. */
int synth_generate(const synth_cfg* c, int8_t** db_data, int64_t** db_off, int64_t* db_n,
	int8_t** q_data, int64_t** q_off, int64_t* q_n, int64_t** q_family)
{
	init_cdf();
	rng_t r; rng_seed(&r, c->seed);
	const int64_t n_db = c->families * c->members;
	const int cap = 2 * c->len_max + 128;
	int64_t dcap = (int64_t)((double)n_db * (c->len_mean + 16) * 1.1) + 1024;
	int8_t* dd = (int8_t*)malloc((size_t)dcap);
	int64_t* doff = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_db + 1));
	int8_t* anc_all = (int8_t*)malloc((size_t)c->families * (size_t)c->len_max);
	int32_t* anc_len = (int32_t*)malloc(sizeof(int32_t) * (size_t)c->families);
	int8_t* tmp = (int8_t*)malloc((size_t)cap);
	if (!dd || !doff || !anc_all || !anc_len || !tmp) return -1;
	int64_t pos = 0, k = 0;
	doff[0] = 0;
	for (int64_t f = 0; f < c->families; ++f) {
		int8_t* anc = anc_all + f * c->len_max;
		const int n = rand_len(&r, c->len_mean, c->len_sd, c->len_min, c->len_max);
		anc_len[f] = n;
		for (int i = 0; i < n; ++i) anc[i] = rand_letter(&r);
		for (int m = 0; m < c->members; ++m) {
			const double sub = c->sub_lo + (c->sub_hi - c->sub_lo) * rng_u(&r);
			const int l = mutate(&r, anc, n, sub, c->indel, tmp, cap);
			if (pos + l > dcap) {
				dcap = dcap * 3 / 2 + l;
				dd = (int8_t*)realloc(dd, (size_t)dcap);
				if (!dd) return -1;
			}
			memcpy(dd + pos, tmp, (size_t)l);
			pos += l;
			doff[++k] = pos;
		}
	}
	int64_t qcap = (int64_t)((double)c->queries * (c->len_mean + 16) * 1.1) + 1024, qpos = 0;
	int8_t* qd = (int8_t*)malloc((size_t)qcap);
	int64_t* qoff = (int64_t*)malloc(sizeof(int64_t) * (size_t)(c->queries + 1));
	int64_t* qfam = (int64_t*)malloc(sizeof(int64_t) * (size_t)(c->queries + 1));
	if (!qd || !qoff || !qfam) return -1;
	qoff[0] = 0;
	for (int64_t q = 0; q < c->queries; ++q) {
		int l;
		if (rng_u(&r) < c->decoy_frac) {
			l = rand_len(&r, c->len_mean, c->len_sd, c->len_min, c->len_max);
			for (int i = 0; i < l; ++i) tmp[i] = rand_letter(&r);
			qfam[q] = -1;
		}
		else {
			const int64_t f = (int64_t)(rng_u(&r) * (double)c->families) % c->families;
			const double sub = c->q_lo + (c->q_hi - c->q_lo) * rng_u(&r);
			l = mutate(&r, anc_all + f * c->len_max, anc_len[f], sub, c->indel, tmp, cap);
			qfam[q] = f;
		}
		if (qpos + l > qcap) {
			qcap = qcap * 3 / 2 + l;
			qd = (int8_t*)realloc(qd, (size_t)qcap);
			if (!qd) return -1;
		}
		memcpy(qd + qpos, tmp, (size_t)l);
		qpos += l;
		qoff[q + 1] = qpos;
	}
	free(anc_all); free(anc_len); free(tmp);
	*db_data = dd; *db_off = doff; *db_n = n_db;
	*q_data = qd; *q_off = qoff; *q_n = c->queries;
	if (q_family) *q_family = qfam; else free(qfam);
	return 0;
}

void synth_free(void* p) { free(p); }

int synth_write_fasta(const char* path, const char* prefix, const int8_t* data, const int64_t* off, int64_t n)
{
	static const char AA[] = "ARNDCQEGHILKMFPSTWYVBJZX*_";
	FILE* f = fopen(path, "w");
	if (!f) return -1;
	char* line = (char*)malloc(1 << 16);
	for (int64_t i = 0; i < n; ++i) {
		fprintf(f, ">%s%lld\n", prefix, (long long)i);
		const int64_t l = off[i + 1] - off[i];
		for (int64_t a = 0; a < l; a += 60000) {
			const int64_t m = l - a < 60000 ? l - a : 60000;
			for (int64_t j = 0; j < m; ++j) line[j] = AA[(int)data[off[i] + a + j]];
			fwrite(line, 1, (size_t)m, f);
		}
		fputc('\n', f);
	}
	free(line);
	fclose(f);
	return 0;
}
