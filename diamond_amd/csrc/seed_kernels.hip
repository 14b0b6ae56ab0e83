// seed_kernels.hip -- gfx950 (MI355X) kernels of the seed stage: replaces
// Search::search_shape (/root/reference/src/search/stage0.cpp:101-217) and its SIMD stage-1/2 filters.
//
// Data flow (per-thread arithmetic in seed_core.h):
//   seed_qid_kernel     query position -> query id (for Hit::query_ and the seed offset)
//   seed_index_kernel   every query position: seed in registers -> open-addressing table in HBM
//                       (keys[], heads[]) + per-position `next` links = per-seed lists of query positions
//   seed_stream_kernel  the reference block is streamed ONCE, coalesced; every position's seed is computed in
//                       registers and probed; a probe hit marks the slot "joined" and appends (slot, position)
//                       -- no reference seed array, no radix passes, no 9-byte scatter (the reference's #1 cost)
//   seed_mask_kernel    per joined seed: low-complexity test (mask_seeds), per-letter mask times
//   seed_pair_kernel    per (joined reference position x query list entry): 48-byte Hamming filter, left-most rule,
//                       Hit append with one atomic per surviving pair
// The table for the 10k-query workload (3e6 seeds, 8M slots, 96 MB) lives in the 256 MB Infinity Cache, so the
// stream kernel's only HBM traffic is the reference letters themselves.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "seed_core.h"
#include "seed_kernels.h"

namespace dmnd {

__global__ void seed_qid_kernel(const int64_t* __restrict__ limits, int64_t n_seqs, uint32_t* __restrict__ qid_of)
{
	// one wavefront per sequence
	const int64_t seq = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	if (seq >= n_seqs) return;
	const int64_t b = limits[seq], e = limits[seq + 1];
	for (int64_t p = b + lane; p < e; p += 64)
		qid_of[p] = (uint32_t)seq;
}

__global__ void seed_index_kernel(SeedArgs a, int sid)
{
	const int64_t p = a.q_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= a.q_end) return;
	uint64_t seed;
	if (!seed_at(a.params, sid, a.qdata + p, seed)) return;
	uint64_t slot = seed_hash(seed) & a.slot_mask;
	for (;;) {
		const unsigned long long old = atomicCAS((unsigned long long*)&a.keys[slot], (unsigned long long)SEED_EMPTY, (unsigned long long)seed);
		if (old == SEED_EMPTY || old == seed) break;
		slot = (slot + 1) & a.slot_mask;
	}
	const uint32_t prev = atomicExch(&a.heads[slot], (uint32_t)(p - a.q_begin));
	a.next[p - a.q_begin] = prev;
}

__global__ void seed_stream_kernel(SeedArgs a, int sid)
{
	const int64_t p = a.t_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= a.t_end) return;
	uint64_t seed;
	if (!seed_at(a.params, sid, a.tdata + p, seed)) return;
	uint64_t slot = seed_hash(seed) & a.slot_mask;
	for (;;) {
		const uint64_t k = a.keys[slot];
		if (k == SEED_EMPTY) return;
		if (k == seed) break;
		slot = (slot + 1) & a.slot_mask;
	}
	a.flags[slot] = SLOT_JOINED;                       // benign race: every writer stores the same value
	const unsigned long long idx = atomicAdd(a.matched_count, 1ull);
	if (idx < (unsigned long long)a.matched_cap) {
		a.matched_slot[idx] = (uint32_t)slot;
		a.matched_loc[idx] = p;
	}
}

__global__ void seed_mask_kernel(SeedArgs a, int sid)
{
	const uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (slot > a.slot_mask || a.flags[slot] != SLOT_JOINED) return;
	// Search::mask_seeds evaluates the first query position of the joined group (seed_complexity.cpp:97-99);
	// "first" = smallest position here (and in the oracle)
	uint32_t first = 0xffffffffu;
	for (uint32_t x = a.heads[slot]; x != LIST_END; x = a.next[x])
		first = x < first ? x : first;
	if (seed_is_complex(a.params, sid, a.qdata + a.q_begin + first)) return;
	a.flags[slot] = SLOT_ERASED;
	const int t = sid * a.params.index_chunks + seed_chunk(a.params, a.keys[slot]);
	for (uint32_t x = a.heads[slot]; x != LIST_END; x = a.next[x]) {
		const uint8_t old = a.mask_time[a.q_begin + x];
		if (t < old) a.mask_time[a.q_begin + x] = (uint8_t)t;       // one group per position and shape: no race within a launch
	}
}

__global__ void seed_pair_kernel(SeedArgs a, int sid, int64_t n_matched)
{
	const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (m >= n_matched) return;
	const uint32_t slot = a.matched_slot[m];
	if (a.flags[slot] == SLOT_ERASED) return;
	const int64_t sloc = a.matched_loc[m];
	const int chunk = seed_chunk(a.params, a.keys[slot]);
	const int8_t* s = a.tdata + sloc;
	for (uint32_t x = a.heads[slot]; x != LIST_END; x = a.next[x]) {
		const int64_t qp = a.q_begin + x;
		const int8_t* q = a.qdata + qp;
		if (fingerprint_id(q, s) < a.params.hamming_filter_id) continue;
		const uint32_t qid = a.qid_of[qp];
		const int seed_offset = (int)(qp - a.qlimits[qid]);
		if (!left_most_pair(a.params, q, a.mask_time + qp, s, seed_offset, sid, chunk)) continue;
		const unsigned long long idx = atomicAdd(a.hit_count, 1ull);
		if (idx < (unsigned long long)a.hit_cap) {
			dmnd_seed_hit h;
			h.query = qid; h.seed_offset = seed_offset; h.subject = sloc; h.score = 0xFFFF; h.pad = 0;
			a.hits[idx] = h;
		}
	}
}

static unsigned blocks_for(int64_t n, int threads) { return (unsigned)((n + threads - 1) / threads); }

hipError_t launch_seed_qid(const int64_t* limits, int64_t n_seqs, uint32_t* qid_of, hipStream_t st)
{
	if (n_seqs == 0) return hipSuccess;
	hipLaunchKernelGGL(seed_qid_kernel, dim3(blocks_for(n_seqs * 64, 256)), dim3(256), 0, st, limits, n_seqs, qid_of);
	return hipGetLastError();
}

hipError_t launch_seed_index(const SeedArgs& a, int sid, hipStream_t st)
{
	hipLaunchKernelGGL(seed_index_kernel, dim3(blocks_for(a.q_end - a.q_begin, 256)), dim3(256), 0, st, a, sid);
	return hipGetLastError();
}

hipError_t launch_seed_stream(const SeedArgs& a, int sid, hipStream_t st)
{
	hipLaunchKernelGGL(seed_stream_kernel, dim3(blocks_for(a.t_end - a.t_begin, 256)), dim3(256), 0, st, a, sid);
	return hipGetLastError();
}

hipError_t launch_seed_mask(const SeedArgs& a, int sid, hipStream_t st)
{
	hipLaunchKernelGGL(seed_mask_kernel, dim3(blocks_for((int64_t)a.slot_mask + 1, 256)), dim3(256), 0, st, a, sid);
	return hipGetLastError();
}

hipError_t launch_seed_pairs(const SeedArgs& a, int sid, int64_t n_matched, hipStream_t st)
{
	if (n_matched == 0) return hipSuccess;
	hipLaunchKernelGGL(seed_pair_kernel, dim3(blocks_for(n_matched, 128)), dim3(128), 0, st, a, sid, n_matched);
	return hipGetLastError();
}

}  // namespace dmnd
