// valu_probe.hip -- issue rate of the instruction kinds the packed 16-bit sweeps are made of (round 6): what is the VALU peak a
// kernel of v_pk_*_i16 / DPP / v_perm / 32-bit integer operations can reach on gfx950? Every kernel runs ITER x 64 operations per
// lane on 16 independent registers (no dependency stalls at any occupancy), 8 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
template<int KIND> __global__ __launch_bounds__(256) void probe(uint32_t* out, int iter, uint32_t seed)
{
	uint32_t r[16];
	for (int k = 0; k < 16; ++k) r[k] = seed * (threadIdx.x + 1) + k;
	const uint32_t c = seed | 1;
	for (int it = 0; it < iter; ++it) {
#pragma unroll
		for (int rep = 0; rep < 4; ++rep)
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				if (KIND == 0) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r[k]) : "v"(c));
				else if (KIND == 1) asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(r[k]) : "v"(c));
				else if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[k]) : "v"(c));
				else if (KIND == 3) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r[k]) : "v"(c));
				else if (KIND == 4) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[k]) : "v"(r[(k + 1) & 15]));
				else if (KIND == 5) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(c), "v"(0x06020400u));
				else if (KIND == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[k]) : "v"(c));
				else if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<f2*>(&r[k & 14])) : "v"(*reinterpret_cast<const f2*>(&r[14])));
				else if (KIND == 8) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[k]) : "v"(c));
				else if (KIND == 9) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(r[k]) : "v"(c));
				else if (KIND == 10) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(r[k]) : "v"(c));
				else if (KIND == 11) asm volatile("v_pk_mad_u16 %0, %0, 2, %1 op_sel_hi:[1,0,1]" : "+v"(r[k]) : "v"(c));
			}
	}
	uint32_t x = 0;
	for (int k = 0; k < 16; ++k) x ^= r[k];
	if (x == 0x12345678u) out[0] = x;
}
template<int KIND> static void run(const char* name, uint32_t* d)
{
	const int iter = 2000, blocks = 256 * 8;          // 8 workgroups of 4 wavefronts per CU = 8 wavefronts per SIMD
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10, 3u);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d, iter, 3u);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	const double inst = (double)blocks * 4 * iter * 64;      // wave-instructions
	std::printf("%-28s %8.3f ms  %8.1f G wave-instructions/s  = %.2f cycles per instruction and SIMD at 2.4 GHz\n", name, ms, inst / ms * 1e-6, 1024 * 2.4e9 / (inst / (ms * 1e-3)));
}
int main()
{
	uint32_t* d; hipMalloc(&d, 64);
	run<0>("v_pk_max_i16", d); run<1>("v_pk_add_i16 clamp", d); run<9>("v_pk_sub_u16 clamp", d); run<11>("v_pk_mad_u16", d);
	run<2>("v_add_u32", d); run<3>("v_max_i32", d); run<8>("v_and_b32", d); run<10>("v_lshl_or_b32", d);
	run<4>("v_mov_b32_dpp row_shr:1", d); run<5>("v_perm_b32", d); run<6>("v_fma_f32", d); run<7>("v_pk_fma_f32", d);
	return 0;
}
