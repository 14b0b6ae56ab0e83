// probe (round 4): start-up costs of the HIP runtime on this box -- runtime up, code-object loads (serial and from two threads),
// stream / pinned / device allocations
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static std::vector<char> slurp(const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
static double load(const std::vector<char>& b) { const double t = now(); hipModule_t m; if (hipModuleLoadData(&m, b.data()) != hipSuccess) return -1; return now() - t; }
int main(int argc, char** argv)
{
	double t = now();
	hipInit(0);
	std::printf("hipInit %.2f ms\n", now() - t); t = now();
	int n = 0; hipGetDeviceCount(&n);
	std::printf("hipGetDeviceCount %.2f ms\n", now() - t); t = now();
	hipSetDevice(0); hipFree(nullptr);
	std::printf("hipSetDevice + hipFree(0) %.2f ms\n", now() - t); t = now();
	hipStream_t s; hipStreamCreate(&s);
	std::printf("first hipStreamCreate %.2f ms\n", now() - t); t = now();
	hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
	std::printf("second hipStreamCreate %.2f ms\n", now() - t);
	for (size_t mb : { 8, 8, 64, 300 }) { t = now(); void* p; hipHostMalloc(&p, mb << 20, hipHostMallocDefault); std::printf("hipHostMalloc %zu MB %.2f ms\n", mb, now() - t); }
	for (size_t mb : { 1, 16, 16, 376, 1700 }) { t = now(); void* p; hipMalloc(&p, mb << 20); std::printf("hipMalloc %zu MB %.2f ms\n", mb, now() - t); }
	{ std::vector<char> h((size_t)300 << 20, 1); t = now(); hipHostRegister(h.data(), h.size(), hipHostRegisterDefault); std::printf("hipHostRegister 300 MB %.2f ms\n", now() - t); }
	if (argc >= 3) {
		const std::vector<char> a = slurp(argv[1]), b = slurp(argv[2]);
		std::printf("serial: %s %.2f ms, %s %.2f ms\n", argv[1], load(a), argv[2], load(b));
		double ta = 0, tb = 0;
		t = now();
		std::thread x([&] { ta = load(a); }), y([&] { tb = load(b); });
		x.join(); y.join();
		std::printf("two threads: %.2f / %.2f ms, wall %.2f ms\n", ta, tb, now() - t);
	}
	return 0;
}
