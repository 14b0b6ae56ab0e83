// CPU lane emulator of the gapped filter kernel (diamond_amd/csrc/gapped_kernels.hip): same per-diagonal code
// (gapped_core.h), lanes run one after the other, v_readlane becomes an array read.
#include <cstdint>
#include "../../diamond_amd/csrc/gapped_core.h"

using namespace dmnd;

// returns flag; f[0] = stage-1 value, f[1] = stage-2 value or -1
extern "C" int emu_gapped_filter_hit(const GfParams* p, const int8_t* M, const int8_t* q, int qlen, const int8_t* cbs,
	const int8_t* t, int slen, int hit_i, int hit_j, int cutoff1, int cutoff2, int* f)
{
	int s[128], d, jb, je, j0, j1;
	hit_window(hit_i, hit_j, slen, 64, 100, d, jb, je);
	scan_range(qlen, d, 64, jb, je, j0, j1);
	for (int lane = 0; lane < 64; ++lane) s[lane] = scan_diag(M, q, qlen, p->use_cbs ? cbs : nullptr, t, d + lane, j0, j1);
	DiagAln al;
	al.init(*p);
	for (int i = 0; i < 64; ++i) al.step(*p, s[i], i);
	f[0] = al.best; f[1] = -1;
	if (f[0] > cutoff1) {
		hit_window(hit_i, hit_j, slen, 128, p->window2, d, jb, je);
		scan_range(qlen, d, 128, jb, je, j0, j1);
		for (int lane = 0; lane < 64; ++lane) {
			s[lane] = scan_diag(M, q, qlen, p->use_cbs ? cbs : nullptr, t, d + lane, j0, j1);
			s[64 + lane] = scan_diag(M, q, qlen, p->use_cbs ? cbs : nullptr, t, d + 64 + lane, j0, j1);
		}
		al.init(*p);
		for (int i = 0; i < 128; ++i) al.step(*p, s[i], i);
		f[1] = al.best;
	}
	return f[1] >= 0 && f[1] > cutoff2;
}
