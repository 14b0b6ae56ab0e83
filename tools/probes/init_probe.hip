// init_probe.hip -- the floor of a HIP process's start-up on this box: runtime up, first kernel of a one-kernel binary (round 6:
// diamond-hip's dmnd_init reports 53 ms + 39 ms for the same two steps with 21 MB of device code registered)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(int* p) { if (p) *p = 1; }
int main()
{
	const auto t0 = std::chrono::steady_clock::now();
	auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
	int n = 0;
	(void)hipGetDeviceCount(&n);
	std::printf("hipGetDeviceCount %.2f ms\n", ms());
	(void)hipSetDevice(0);
	std::printf("hipSetDevice %.2f ms\n", ms());
	hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, nullptr, (int*)nullptr);
	std::printf("first launch returned %.2f ms\n", ms());
	(void)hipDeviceSynchronize();
	std::printf("first kernel done %.2f ms\n", ms());
	hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	std::printf("stream created %.2f ms\n", ms());
	void* p; (void)hipMalloc(&p, 1 << 20);
	std::printf("hipMalloc 1 MB %.2f ms\n", ms());
	void* q; (void)hipMalloc(&q, (size_t)1 << 30);
	std::printf("hipMalloc 1 GB %.2f ms\n", ms());
	void* h; (void)hipHostMalloc(&h, 16 << 20, hipHostMallocDefault);
	std::printf("hipHostMalloc 16 MB %.2f ms\n", ms());
	return 0;
}
