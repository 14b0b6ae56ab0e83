/* oracle/motif_mask.c -- TEST INFRASTRUCTURE ONLY (parity checker of the motif soft masking; never used by the product).
 * Plain-C restatement of mask_motifs (/root/reference/src/masking/masking.cpp:110-131) with KmerIterator<8>
 * (src/util/kmer/kmer.h:61-105: only 8-mers of letters < 20; a non-standard letter restarts the k-mer) and Mask::Ranges
 * (src/masking/def.h:70-85: a range that begins at or before the end of the last one extends it).
 * Returns the number of letters under motifs (0 when they make up half of the sequence or more: nothing is masked then);
 * seq receives the mask letter over every range of at most max_motif_len letters. table: sorted Kmer<8> codes. */
#include <stdint.h>
#include <stdlib.h>
#include "oracle.h"

static int in_table(const uint64_t* table, int n, uint64_t code)
{
	int lo = 0, hi = n;
	while (lo < hi) { const int mid = (lo + hi) / 2; if (table[mid] < code) lo = mid + 1; else hi = mid; }
	return lo < n && table[lo] == code;
}

int oracle_motif_mask(int8_t* seq, int len, const uint64_t* table, int n_table, int max_motif_len)
{
	if (len < 8) return 0;
	int* rb = (int*)malloc(sizeof(int) * (size_t)len);
	int* re = (int*)malloc(sizeof(int) * (size_t)len);
	int nr = 0;
	uint64_t code = 0, mod = 1;
	for (int i = 0; i < 7; ++i) mod *= 20;              /* power(20, K - 1) */
	int run = 0;                                        /* standard letters accumulated in the k-mer */
	for (int p = 0; p < len; ++p) {
		const int l = seq[p] & 31;
		if (l < 20) {
			if (run >= 8) code %= mod;                  /* operator++: drop the oldest letter */
			code = code * 20 + (uint64_t)l;
			++run;
		}
		else { code = 0; run = 0; }
		if (run >= 8) {
			const int begin = p - 7;
			if (in_table(table, n_table, code)) {
				if (nr == 0 || begin > re[nr - 1]) { rb[nr] = begin; re[nr] = begin + 8; ++nr; }
				else re[nr - 1] = begin + 8;
			}
		}
	}
	long n = 0;
	for (int i = 0; i < nr; ++i) n += re[i] - rb[i];
	if ((double)n / len >= 0.5) { free(rb); free(re); return 0; }
	for (int i = 0; i < nr; ++i)
		if (re[i] - rb[i] <= max_motif_len)
			for (int x = rb[i]; x < re[i]; ++x) seq[x] = 23;
	free(rb); free(re);
	return (int)n;
}
