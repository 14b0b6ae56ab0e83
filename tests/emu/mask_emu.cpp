// CPU lane emulator of the tantan kernel (diamond_amd/csrc/mask_kernels.hip): the same per-lane arithmetic
// (mask_core.h), 64 lanes held in arrays, ds_swizzle / v_readlane become array reads. Compile without FP contraction.
#include <cstddef>
#include <cstdint>
#include <vector>
#include "../../diamond_amd/csrc/mask_core.h"

using namespace dmnd;

namespace {

struct Wave { float v[64]; };

float ordered_sum(const Wave& w)
{
	// tantan_group_sum's butterfly, stage by stage over the whole wave (lane l exchanges with lane l ^ mask)
	float a[64], b[64], c[64];
	for (int l = 0; l < 64; ++l) a[l] = w.v[l] + w.v[l ^ 4];
	for (int l = 0; l < 64; ++l) b[l] = a[l] + a[l ^ 1];
	for (int l = 0; l < 64; ++l) c[l] = b[l] + b[l ^ 2];
	float s = 0.0f;
	for (int k = 0; k < 6; ++k) s = s + c[8 * k];
	s = s + w.v[48];
	s = s + w.v[49];
	return s;
}

}

extern "C" int emu_tantan_mask(const TantanParams* p, const float* L, int8_t* seq, int len)
{
	if (len <= 0) return 0;
	std::vector<float> pb((std::size_t)len), scale((std::size_t)len / 16 + 1);
	Wave f;
	for (int l = 0; l < 64; ++l) f.v[l] = 0.0f;
	float b = 1.0f, f_sum = 0.0f;
	for (int i = 0; i < len; ++i) {
		const int ltr = seq[i] & 31;
		const float b_old = b;
		for (int l = 0; l < 64; ++l) {
			const bool own = l < TANTAN_WINDOW;
			const int hp = i - 1 - l;
			const float e = (own && hp >= 0) ? L[ltr * 32 + (seq[hp] & 31)] : 0.0f;
			f.v[l] = tantan_fwd_cell(f.v[l], p->f2f, b_old, own ? p->d[l] : 0.0f, e);
		}
		const float f_sum_new = ordered_sum(f);
		{ const float t1 = b_old * p->b2b, t2 = f_sum * p->p_repeat_end; b = t1 + t2; }
		f_sum = f_sum_new;
		if ((i & 15) == 15) {
			const float s = 1.0f / b;
			scale[(std::size_t)i >> 4] = s;
			b = b * s;
			for (int l = 0; l < 64; ++l) f.v[l] = f.v[l] * s;
			f_sum = f_sum * s;
		}
		pb[(std::size_t)i] = b;
	}
	float acc[64];
	for (int l = 0; l < 64; ++l) { acc[l] = 0.0f; for (int g = 0; g < 6; ++g) acc[l] = acc[l] + f.v[(l & 7) + 8 * g]; }
	float total;
	{ float a[8], bb[8]; for (int l = 0; l < 8; ++l) a[l] = acc[l] + acc[l ^ 4]; for (int l = 0; l < 8; ++l) bb[l] = a[l] + a[l ^ 1]; total = bb[0] + bb[2]; }
	total = total + f.v[48];
	total = total + f.v[49];
	float z;
	{ const float t1 = b * p->b2b, t2 = total * p->p_repeat_end; z = t1 + t2; }
	const float zinv = 1.0f / z;
	b = p->b2b;
	for (int l = 0; l < 64; ++l) f.v[l] = l < TANTAN_WINDOW ? p->p_repeat_end : 0.0f;
	std::vector<char> mask((std::size_t)len, 0);
	int n_masked = 0;
	for (int i = len - 1; i >= 0; --i) {
		float pf;
		{ const float t = pb[(std::size_t)i] * b; const float u = t * zinv; pf = 1.0f - u; }
		if ((i & 15) == 15) {
			const float s = scale[(std::size_t)i >> 4];
			b = b * s;
			for (int l = 0; l < 64; ++l) f.v[l] = f.v[l] * s;
		}
		const int ltr = seq[i] & 31;
		const float C = p->p_repeat_end * b;
		Wave vt;
		for (int l = 0; l < 64; ++l) {
			const bool own = l < TANTAN_WINDOW;
			const int hp = i - 1 - l;
			const float e = (own && hp >= 0) ? L[ltr * 32 + (seq[hp] & 31)] : 0.0f;
			f.v[l] = tantan_bwd_cell(f.v[l], e, own ? p->d[l] : 0.0f, p->f2f, C, vt.v[l]);
			if (!own) { f.v[l] = 0.0f; vt.v[l] = 0.0f; }
		}
		const float tsum = ordered_sum(vt);
		{ const float t1 = p->b2b * b; b = t1 + tsum; }
		if (pf >= p->p_mask) { mask[(std::size_t)i] = 1; ++n_masked; }
	}
	for (int i = 0; i < len; ++i) if (mask[(std::size_t)i]) seq[i] = 23;
	return n_masked;
}

// motif soft masking of one sequence through the product's per-thread code (mask_core.h): motif_hit_kernel's test per
// position, motif_apply_kernel's per-sequence rule
extern "C" int emu_motif_mask(int8_t* seq, int len, const uint64_t* table, int n_table, int max_range)
{
	std::vector<uint8_t> hit((size_t)len + 8, 0);
	for (int p = 0; p + dmnd::MOTIF_LEN <= len; ++p) {
		uint64_t code;
		hit[(size_t)p] = dmnd::motif_code_at(seq + p, code) && dmnd::motif_in_table(table, n_table, code) ? 1 : 0;
	}
	return dmnd::motif_mask_sequence(seq, hit.data(), len, max_range);
}
