// Probe: does a small device-to-host copy on stream B wait for a long kernel running on stream A?
// pageable vs pinned destination, hipMemcpyAsync + hipStreamSynchronize. Build: hipcc --offload-arch=gfx950 -O2 -o copy_probe copy_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long cycles, int* sink) { const long long t0 = clock64(); while (clock64() - t0 < cycles) {} if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1; }
__global__ void tiny(int* p) { p[threadIdx.x] = threadIdx.x; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	hipStream_t a, b;
	int least, greatest;
	hipDeviceGetStreamPriorityRange(&least, &greatest);
	hipStreamCreateWithPriority(&a, hipStreamNonBlocking, least);
	hipStreamCreateWithPriority(&b, hipStreamNonBlocking, greatest);
	int *d, *sink, *pinned;
	hipMalloc(&d, 4096); hipMalloc(&sink, 4); hipHostMalloc(&pinned, 4096);
	std::vector<int> pageable(1024);
	for (int mode = 0; mode < 4; ++mode) {
		for (int rep = 0; rep < 3; ++rep) {
			const bool busy = mode & 1, pin = mode & 2;
			if (busy) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, 5000000LL, sink);     // ~2 ms at 2.4 GHz on a quarter of the CUs
			const double t0 = now();
			hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, d);
			const double t1 = now();
			hipMemcpyAsync(pin ? (void*)pinned : (void*)pageable.data(), d, 256, hipMemcpyDeviceToHost, b);
			hipStreamSynchronize(b);
			const double t2 = now();
			hipMemcpyAsync(d, pin ? (void*)pinned : (void*)pageable.data(), 256, hipMemcpyHostToDevice, b);
			hipStreamSynchronize(b);
			const double t3 = now();
			hipStreamSynchronize(a);
			const double t4 = now();
			std::printf("long kernel on other stream: %d  pinned: %d | launch %.3f ms, D2H+sync %.3f ms, H2D+sync %.3f ms, other stream done after %.3f ms\n", busy, pin, t1 - t0, t2 - t1, t3 - t2, t4 - t0);
		}
	}
	return 0;
}
