// wait_probe.hip -- CPU time a host thread spends inside the different ways of waiting for the GPU (round 6: the seed stage's thread
// was found to burn ~1 CPU-ms per ms of kernel time although it waits on hipEventBlockingSync events)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <ctime>
#include <chrono>
__global__ void busy(long long cycles, int* out) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) {} if (out) *out = 1; }
static double cpu_ms() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static double proc_ms() { timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
	if (argc > 1 && argv[1][0] == 'b') std::printf("hipSetDeviceFlags(hipDeviceScheduleBlockingSync) -> %d\n", (int)hipSetDeviceFlags(hipDeviceScheduleBlockingSync));
	if (argc > 1 && argv[1][0] == 'y') std::printf("hipSetDeviceFlags(hipDeviceScheduleYield) -> %d\n", (int)hipSetDeviceFlags(hipDeviceScheduleYield));
	hipStream_t st, lo;
	hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	int least, greatest; hipDeviceGetStreamPriorityRange(&least, &greatest);
	hipStreamCreateWithPriority(&lo, hipStreamNonBlocking, least);
	hipEvent_t blk, dflt, blk_t;
	hipEventCreateWithFlags(&blk, hipEventBlockingSync | hipEventDisableTiming);
	hipEventCreateWithFlags(&blk_t, hipEventBlockingSync);
	hipEventCreateWithFlags(&dflt, hipEventDefault);
	const long long cyc = 100000000LL / 2;      // wall_clock64 ticks at 100 MHz: 0.5 s ... scaled below
	int* d; hipMalloc(&d, 4);
	int* pinned; hipHostMalloc(&pinned, 4);
	for (int mode = 0; mode < 10; ++mode) {
		hipStream_t s = mode >= 6 ? lo : st;
		hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, s, 5000000LL, d);      // 50 ms
		const double c0 = cpu_ms(), p0 = proc_ms(), w0 = wall_ms();
		const char* name = "";
		switch (mode) {
		case 0: name = "hipStreamSynchronize"; hipStreamSynchronize(s); break;
		case 1: name = "record + hipEventSynchronize(BlockingSync|DisableTiming)"; hipEventRecord(blk, s); hipEventSynchronize(blk); break;
		case 2: name = "record + hipEventSynchronize(BlockingSync, timing)"; hipEventRecord(blk_t, s); hipEventSynchronize(blk_t); break;
		case 3: name = "record + hipEventSynchronize(Default)"; hipEventRecord(dflt, s); hipEventSynchronize(dflt); break;
		case 4: name = "hipMemcpyAsync D2H pinned + blocking event"; hipMemcpyAsync(pinned, d, 4, hipMemcpyDeviceToHost, s); hipEventRecord(blk, s); hipEventSynchronize(blk); break;
		case 5: { name = "hipMemcpyAsync D2H PAGEABLE + blocking event"; int x; hipMemcpyAsync(&x, d, 4, hipMemcpyDeviceToHost, s); hipEventRecord(blk, s); hipEventSynchronize(blk); break; }
		case 6: name = "low-priority stream: blocking event"; hipEventRecord(blk, s); hipEventSynchronize(blk); break;
		case 7: { name = "low-priority stream: D2H pageable + blocking event"; int x; hipMemcpyAsync(&x, d, 4, hipMemcpyDeviceToHost, s); hipEventRecord(blk, s); hipEventSynchronize(blk); break; }
		case 8: { name = "record + hipEventQuery / nanosleep(50 us) poll"; hipEventRecord(blk, s); timespec ts{ 0, 50000 }; while (hipEventQuery(blk) == hipErrorNotReady) nanosleep(&ts, nullptr); break; }
		case 9: { name = "hipStreamQuery / nanosleep(50 us) poll"; timespec ts{ 0, 50000 }; while (hipStreamQuery(s) == hipErrorNotReady) nanosleep(&ts, nullptr); break; }
		}
		std::printf("%-62s wall %7.2f ms  thread CPU %7.2f ms  process CPU %7.2f ms\n", name, wall_ms() - w0, cpu_ms() - c0, proc_ms() - p0);
	}
	(void)cyc;
	return 0;
}
