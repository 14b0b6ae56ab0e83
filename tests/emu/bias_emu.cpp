// tests/emu/bias_emu.cpp -- TEST INFRASTRUCTURE ONLY. Runs the per-position closed form of the Hauser bias
// (diamond_amd/csrc/bias_core.h, the code of hauser_bias_kernel) on the CPU for one sequence.
#include "../../diamond_amd/csrc/bias_core.h"

extern "C" void emu_hauser_bias(const int8_t* seq, int l, const int8_t* M, const float* bg, int window, int8_t* out)
{
	for (int m = 0; m < l; ++m) out[m] = dmnd::hauser_at(seq, l, m, M, bg, window);
}
