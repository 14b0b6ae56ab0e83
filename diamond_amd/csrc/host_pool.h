// host_pool.h -- persistent host worker pool shared by the host parts of libdiamond_hip.so (extension stage, swipe call
// preparation). One pool per process; a loop that finds it busy falls back to plain threads.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <thread>
#include <vector>
#include "tuning.h"

namespace dmnd {

// Persistent worker pool: dmnd_extend issues a handful of short parallel loops per call, and thread start-up would
// cost more than the loops themselves. Workers sleep on a condition variable between loops; the caller is worker 0.
inline void cpu_relax()
{
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
	__builtin_ia32_pause();
#endif
}

// Workers first spin on an atomic generation counter for a few tens of microseconds (dmnd_extend issues its short parallel
// loops back to back: a sleeping worker costs ~0.1 ms of wake-up per loop with 64 threads), then sleep on a condition variable.
class WorkerPool {
public:
	~WorkerPool()
	{
		{ std::lock_guard<std::mutex> g(m_); stop_ = true; stop_a_.store(true); }
		cv_.notify_all();
		for (auto& t : th_) t.join();
	}
	// f(t) once on each of `threads` participants, participant t always being the same OS thread (the caller is 0): work that
	// is cut into one fixed slice per participant stays in that core's caches from one parallel phase of a call to the next
	template<typename F>
	void run_each(int threads, F& f)
	{
		auto g = [&f](size_t, int t) { f(t); };
		run_impl((size_t)threads, threads, g, true);
	}
	template<typename F>
	void run(size_t n, int threads, F& f) { run_impl(n, threads, f, false); }
private:
	template<typename F>
	void run_impl(size_t n, int threads, F& f, bool each)
	{
		std::unique_lock<std::mutex> own(busy_, std::try_to_lock);
		if (!own.owns_lock()) {                              // another context is using the pool: plain threads for this loop
			std::vector<std::thread> th;
			std::atomic<size_t> next(0);
			for (int t = 0; t < threads; ++t) th.emplace_back([&, t] { if (each) { f((size_t)t, t); return; } size_t i; while ((i = next.fetch_add(1)) < n) f(i, t); });
			for (auto& x : th) x.join();
			return;
		}
		grow(threads - 1);
		std::function<void(size_t, int)> fn = [&f](size_t i, int t) { f(i, t); };
		{
			std::lock_guard<std::mutex> g(m_);
			fn_ = &fn; n_ = n; each_ = each; next_.store(0); want_ = threads - 1; want_a_.store(threads - 1);
			running_.store(threads - 1);
			++gen_;
			gen_a_.store(gen_, std::memory_order_release);      // publishes the job to the spinning workers
		}
		cv_.notify_all();
		if (each) f(0, 0);
		else { size_t i; while ((i = next_.fetch_add(1)) < n) f(i, 0); }
		for (int s = 0, n = spins() * 8; s < n && running_.load(std::memory_order_acquire) != 0; ++s) cpu_relax();
		if (running_.load(std::memory_order_acquire) != 0) {
			std::unique_lock<std::mutex> g(m_);
			done_.wait(g, [&] { return running_.load(std::memory_order_acquire) == 0; });
		}
		fn_ = nullptr;
	}
	// iterations of the spin phase (a pause instruction each, ~25 ns): long enough to catch the next of a run of back-to-back
	// loops, short enough not to eat a CPU quota with dozens of idle spinning workers (DMND_POOL_SPINS overrides)
	static int spins()
	{
		return tuning().pool_spins;
	}
	void grow(int workers)
	{
		while ((int)th_.size() < workers) {
			const int id = (int)th_.size() + 1;
			th_.emplace_back([this, id] {
				pthread_setname_np(pthread_self(), "dmnd-pool");      // (per-thread CPU accounting by name: bench.py host_cpu_ms_per_step_by_thread)
				uint64_t seen = 0;
				for (;;) {
					bool got = false;
					for (int s = 0, n = spins(); s < n; ++s) {
						if (stop_a_.load(std::memory_order_relaxed)) return;
						if (gen_a_.load(std::memory_order_acquire) != seen && id <= want_a_.load(std::memory_order_relaxed)) { got = true; break; }
						cpu_relax();
					}
					if (!got) {
						std::unique_lock<std::mutex> g(m_);
						cv_.wait(g, [&] { return stop_ || (gen_ != seen && id <= want_); });
						if (stop_) return;
					}
					// the job fields were written before gen_a_ was released (and under m_)
					seen = gen_a_.load(std::memory_order_acquire);
					const std::function<void(size_t, int)>* fn = fn_;
					const size_t n = n_;
					if (each_) (*fn)((size_t)id, id);
					else { size_t i; while ((i = next_.fetch_add(1)) < n) (*fn)(i, id); }
					if (running_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
						std::lock_guard<std::mutex> g(m_);           // pairs with the predicate check of the waiting caller
						done_.notify_one();
					}
				}
			});
		}
	}
	std::vector<std::thread> th_;
	std::mutex m_, busy_;
	std::condition_variable cv_, done_;
	const std::function<void(size_t, int)>* fn_ = nullptr;
	size_t n_ = 0;
	bool each_ = false;
	std::atomic<size_t> next_{ 0 };
	int want_ = 0;
	std::atomic<int> want_a_{ 0 }, running_{ 0 };
	uint64_t gen_ = 0;
	std::atomic<uint64_t> gen_a_{ 0 };
	bool stop_ = false;
	std::atomic<bool> stop_a_{ false };
};

enum { MAX_POOLS = 16 };
// pool(): the worker pool of the calling thread. Threads that drive one of the concurrent sub-batches of dmnd_extend select
// their own pool with set_thread_pool(k) (k < MAX_POOLS); every other thread shares the default pool. Defined in extend_host.hip.
WorkerPool& pool();
void set_thread_pool(int k);
int thread_pool();

// f(t) for t in [0, threads), participant t pinned to one thread of the calling thread's pool
template<typename F>
void parallel_each(int threads, F f)
{
	if (threads <= 1) { f(0); return; }
	pool().run_each(threads, f);
}

template<typename F>
void parallel_for(size_t n, int threads, F f)
{
	threads = std::max(1, std::min<int>(threads, (int)n));
	if (threads == 1) { for (size_t i = 0; i < n; ++i) f(i, 0); return; }
	pool().run(n, threads, f);
}


}  // namespace dmnd
