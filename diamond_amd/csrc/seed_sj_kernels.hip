// seed_sj_kernels.hip -- round 5 experiment, off by default (DMND_SEED_SJ=1): the short-seed join of the reference stream as a
// SCATTER + JOIN over 64 key partitions instead of the fused by-class stream kernel (seed_kernels.hip).
//
// Why: the fused kernel's time is random fabric reads -- ~1 miss per joined window whatever the probe
// structure, because a key class's query side (table eighth, lists, folded windows: 9-13 MB) does not fit an XCD's 4 MB L2. A join
// whose random accesses DO fit needs partitions eight times smaller, i.e. the windows of a partition brought together first:
//   K1 seed_sj_scatter_kernel: the reference block is streamed as the long-seed stream does it (16 window starts per thread, keys
//      from nibble windows, level-1 filter probes in batches); a level-1 positive -- 27 % real joins + the filter's false
//      positives -- becomes a 32-byte entry { compact key, position, the window's 48 folded letters } in the slab of its
//      (workgroup, partition). A slab is private to its workgroup: ranks come from an LDS counter per partition, entries of a slab
//      are written back to back, NO global cursor and no global atomic (round 3's partitioned join spent 6 ms per shape on 1e8
//      returning fabric atomics and partial-line writes). A full slab spills to a global overflow list (rare).
//   K2 seed_sj_join_kernel: partition p = (key class c, three top bits of the hash) owns a contiguous 1/64 of the table
//      (SeedArgs::home with parts = 64); workgroups of class c run on XCD c (blockIdx mod 8, the same affinity the by-class kernel
//      uses) and walk the sub-partitions in order, so that at any time an XCD probes ~1/64 of the query side: 2 MB of slots + the
//      1.5 MB of folded query windows. An entry carries everything the Hamming pre-filter needs from the reference side.
//   seed_sj_overflow_kernel: the spilled windows, one thread each.
// Output = the fused kernel's: JOINED marks, the joined-position list (slot, position) for the deferred pass, the survivor list.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "seed_core.h"
#include "seed_kernels.h"

namespace dmnd {

namespace {

__device__ __forceinline__ int sj_window_identity(const uint32_t* a, const uint32_t* b)
{
	int n = 0;
#pragma unroll
	for (int w = 0; w < 12; ++w) {
		const uint32_t d = (a[w] ^ b[w]) & 0x1f1f1f1fu;
		n += 4 - __builtin_popcount((d + 0x7f7f7f7fu) & 0x80808080u);
	}
	return n;
}

__device__ __forceinline__ uint32_t sj_reduce4(uint32_t letter, uint64_t map_lo, uint64_t map_hi)
{
	const uint64_t m = (letter & 16) ? map_hi : map_lo;
	return (uint32_t)(m >> ((letter & 15) * 4)) & 15u;
}

// the 48 folded letters [pos - 16, pos + 32) of a block's folded copy (4 bits per letter, two per byte)
__device__ __forceinline__ void sj_fold_window(const uint8_t* fold, int64_t pos, uint32_t (&tf)[6])
{
	const int64_t t0 = pos - 16;
	uint32_t raw[8];
	__builtin_memcpy(raw, fold + (t0 >> 1), 32);
	const uint32_t sh = (uint32_t)(t0 & 1) * 4;
#pragma unroll
	for (int w = 0; w < 6; ++w) tf[w] = __builtin_amdgcn_alignbit(raw[w + 1], raw[w], sh);
}

// key <-> its 4 bits per care position, packed (weight <= 8)
__device__ __forceinline__ uint32_t sj_compact(const SeedParams& c, int sid, uint64_t key)
{
	uint32_t k = 0;
	for (int j = 0; j < c.shape_weight[sid]; ++j) k |= ((uint32_t)(key >> (4 * c.shape_pos[sid][j])) & 15u) << (4 * j);
	return k;
}
__device__ __forceinline__ uint64_t sj_expand(const SeedParams& c, int sid, uint32_t k)
{
	uint64_t key = 0;
	for (int j = 0; j < c.shape_weight[sid]; ++j) key |= (uint64_t)((k >> (4 * j)) & 15u) << (4 * c.shape_pos[sid][j]);
	return key;
}

__device__ __forceinline__ uint32_t sj_part(uint64_t key, uint32_t h) { return (seed_class(key) << 3) | (h >> 29); }

// Everything behind a level-1 positive whose key, position and folded window are at hand: table probe, JOINED mark, and for the
// seed's query positions the Hamming pre-filter on folded letters, the exact count for the few that pass.
// on_join(slot, head, count) -> true: the caller filters the list itself (heavy lists); on_survivor(slot, x).
template<typename OnJoin, typename OnSurvivor>
__device__ __forceinline__ void sj_join_window(const SeedArgs& a, uint64_t key, uint32_t h, int64_t pos, const uint32_t (&tf)[6], OnJoin&& on_join, OnSurvivor&& on_survivor)
{
	uint64_t slot = a.home((uint64_t)h, key);
	SeedSlot sl = a.slot(slot);
	for (;;) {
		if (sl.key == SEED_EMPTY) return;
		if (sl.key == key) break;
		slot = (slot + 1) & a.slot_mask;
		sl = a.slot(slot);
	}
	if (!(sl.flags & SLOT_JOINED)) a.slot(slot).flags = sl.flags | SLOT_JOINED;      // benign race: every writer stores the same value
	if (sl.flags & SLOT_LOWC) return;
	const uint32_t count = sl.flags >> 8;
	if (on_join((uint32_t)slot, sl.head, count)) return;
	for (uint32_t i = 0; i < count; ++i) {
		const uint32_t x = count == 1 ? sl.head : a.qlist[sl.head + i];
		const int64_t x0 = a.q_begin + (int64_t)x - 16;
		uint32_t raw[8];
		__builtin_memcpy(raw, a.qfold + (x0 >> 1), 32);
		const uint32_t sh = (uint32_t)(x0 & 1) * 4;
		int mism = 0;
#pragma unroll
		for (int w = 0; w < 6; ++w) {
			const uint32_t d = tf[w] ^ __builtin_amdgcn_alignbit(raw[w + 1], raw[w], sh);
			mism += __builtin_popcount((((d & 0x77777777u) + 0x77777777u) | d) & 0x88888888u);
		}
		if (48 - mism < a.params.hamming_filter_id) continue;
		uint32_t qw[12], tw[12];
		__builtin_memcpy(qw, a.qdata + x0, 48);
		__builtin_memcpy(tw, a.tdata + pos - 16, 48);
		if (sj_window_identity(tw, qw) >= a.params.hamming_filter_id) on_survivor((uint32_t)slot, x);
	}
}

}  // namespace

// ---- K1 ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seed_sj_scatter_kernel(SeedArgs a, SeedSjArgs j, int sid, uint64_t map_lo, uint64_t map_hi, int64_t base, uint64_t care64)
{
	__shared__ unsigned cnt[SEED_SJ_PARTS];
	if (threadIdx.x < SEED_SJ_PARTS) cnt[threadIdx.x] = 0;
	__syncthreads();
	SeedSjEntry* const my_slabs = j.slabs + (size_t)blockIdx.x * SEED_SJ_PARTS * SEED_SJ_SLAB;
	const int len = a.params.shape_len[sid];
	const uint32_t care = a.params.shape_mask[sid], span = (1u << len) - 1;      // len <= 16
	auto emit = [&](uint64_t key, uint32_t h, int64_t pos) {
		const uint32_t part = sj_part(key, h);
		const unsigned r = atomicAdd(&cnt[part], 1u);
		if (r >= j.slab_limit) {                                 // slab full: the window goes to the overflow list
			const unsigned long long o = atomicAdd(j.overflow_count, 1ull);
			if (o < (unsigned long long)j.overflow_cap) { j.overflow[2 * o] = key; j.overflow[2 * o + 1] = (uint64_t)pos; }
			return;
		}
		uint32_t tf[6];
		sj_fold_window(a.tfold, pos, tf);
		uint4* e = reinterpret_cast<uint4*>(my_slabs + (size_t)part * SEED_SJ_SLAB + r);
		e[0] = make_uint4(sj_compact(a.params, sid, key), (uint32_t)(pos - base), tf[0], tf[1]);
		e[1] = make_uint4(tf[2], tf[3], tf[4], tf[5]);
	};
#pragma unroll 1
	for (int sub = 0; sub < SEED_SJ_TILES; ++sub) {
		const int64_t p0 = base + (((int64_t)blockIdx.x * SEED_SJ_TILES + sub) * 256 + threadIdx.x) * 16;
		if (p0 >= a.t_end) continue;
		typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
		const u32x4 v0 = *reinterpret_cast<const u32x4*>(a.tseed + p0), v1 = *reinterpret_cast<const u32x4*>(a.tseed + p0 + 16);
		const uint32_t w[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
		uint64_t codes[2] = { 0, 0 };
		uint32_t delim = 0, bad = 0;
#pragma unroll
		for (int k = 0; k < 32; ++k) {
			const uint32_t l = (w[k >> 2] >> ((k & 3) * 8)) & LETTER_MASK;
			const uint32_t c = sj_reduce4(l, map_lo, map_hi);
			codes[k >> 4] |= (uint64_t)c << ((k & 15) * 4);
			delim |= (l == L_DELIM ? 1u : 0u) << k;
			bad |= (c == 15u ? 1u : 0u) << k;
		}
		const int64_t first = a.t_begin - p0, last = a.t_end - p0;                 // valid window starts: first <= i < last
#pragma unroll 1
		for (int half = 0; half < 2; ++half) {
			uint64_t key[8];
			uint32_t hash[8], pos_mask = 0;
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				const int sh = (8 * half + i) * 4;
				key[i] = (sh == 0 ? codes[0] : (codes[0] >> sh) | (codes[1] << (64 - sh))) & care64;
			}
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				const int w0 = 8 * half + i;
				const bool ok = w0 >= first && w0 < last && ((delim >> w0) & span) == 0 && ((bad >> w0) & care) == 0;
				hash[i] = seed_hash_a(key[i]);
				const uint32_t bw = ok ? a.bitmap1[a.bm1_index(hash[i], key[i])] : 0u;
				const uint32_t need = bm1_bits(hash[i], a.bitmap1_k3);
				pos_mask |= ((bw & need) == need ? 1u : 0u) << i;
			}
			while (pos_mask) {
				const int i = __builtin_ctz(pos_mask);
				pos_mask &= pos_mask - 1;
				uint64_t k = 0;
				uint32_t h = 0;
#pragma unroll
				for (int x = 0; x < 8; ++x) if (x == i) { k = key[x]; h = hash[x]; }
				emit(k, h, p0 + 8 * half + i);
			}
		}
	}
	__syncthreads();
	if (threadIdx.x < SEED_SJ_PARTS) j.counts[(size_t)blockIdx.x * SEED_SJ_PARTS + threadIdx.x] = cnt[threadIdx.x] < j.slab_limit ? cnt[threadIdx.x] : j.slab_limit;
}

// ---- K2 ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seed_sj_join_kernel(SeedArgs a, SeedSjArgs j, int sid, int64_t base, int units_per_part)
{
	constexpr unsigned G = SEED_SJ_GROUP, STAGE = SEED_SJ_GROUP * SEED_SJ_SLAB, SURV = 256, HEAVY = 64, LIGHT = 8;
	__shared__ unsigned pre[G + 1];
	__shared__ uint32_t st_slot[STAGE], st_pos[STAGE];
	__shared__ uint32_t sv_slot[SURV], sv_x[SURV], sv_pos[SURV];
	__shared__ uint32_t hv_slot[HEAVY], hv_pos[HEAVY], hv_head[HEAVY], hv_count[HEAVY];
	__shared__ unsigned st_n, sv_n, hv_n;
	__shared__ unsigned long long st_base;
	const uint32_t cls = blockIdx.x & 7u;
	const int64_t jj = (int64_t)(blockIdx.x >> 3);
	const uint32_t sub = (uint32_t)(jj / units_per_part);
	const int64_t u = jj % units_per_part;
	if (sub >= 8) return;
	const uint32_t part = (cls << 3) | sub;
	const int64_t wg0 = u * G;
	if (threadIdx.x == 0) { st_n = 0; sv_n = 0; hv_n = 0; pre[0] = 0; }
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned acc = 0;
		for (unsigned g = 0; g < G; ++g) {
			const int64_t wg = wg0 + g;
			acc += wg < j.n_wg ? j.counts[(size_t)wg * SEED_SJ_PARTS + part] : 0u;
			pre[g + 1] = acc;
		}
	}
	__syncthreads();
	const unsigned E = pre[G];
	auto survive = [&](uint32_t slot, uint32_t x, uint32_t rel) {
		const unsigned k = atomicAdd(&sv_n, 1u);
		if (k < SURV) { sv_slot[k] = slot; sv_x[k] = x; sv_pos[k] = rel; }
		else {
			const unsigned long long idx = atomicAdd(a.survivor_count, 1ull);
			if (idx < (unsigned long long)a.survivor_cap) a.survivors[idx] = SeedSurvivor{ slot, x, base + (int64_t)rel };
		}
	};
	for (unsigned e = threadIdx.x; e < E; e += 256) {
		unsigned g = 0;
#pragma unroll
		for (unsigned step = G / 2; step > 0; step >>= 1) if (pre[g + step] <= e) g += step;
		const SeedSjEntry* src = j.slabs + ((size_t)(wg0 + g) * SEED_SJ_PARTS + part) * SEED_SJ_SLAB + (e - pre[g]);
		typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
		const u32x4 e0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src)), e1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src) + 1);      // read once
		const uint32_t tf[6] = { e0.z, e0.w, e1.x, e1.y, e1.z, e1.w };
		const uint64_t key = sj_expand(a.params, sid, e0.x);
		const uint32_t rel = e0.y;
		sj_join_window(a, key, seed_hash_a(key), base + (int64_t)rel, tf,
			[&](uint32_t slot, uint32_t head, uint32_t count) {
				const unsigned k = atomicAdd(&st_n, 1u);           // (never more joins than entries: k < STAGE)
				st_slot[k] = slot; st_pos[k] = rel;
				if (count <= LIGHT) return false;
				const unsigned hk = atomicAdd(&hv_n, 1u);
				if (hk >= HEAVY) return false;                       // no room: this thread walks the long list itself
				hv_slot[hk] = slot; hv_pos[hk] = rel; hv_head[hk] = head; hv_count[hk] = count;
				return true;
			},
			[&](uint32_t slot, uint32_t x) { survive(slot, x, rel); });
	}
	__syncthreads();
	// lists longer than LIGHT: the whole workgroup filters them, the reference window folded again from the block's folded copy
	const unsigned n_heavy = hv_n < HEAVY ? hv_n : HEAVY;
	for (unsigned h = 0; h < n_heavy; ++h) {
		const int64_t pos = base + (int64_t)hv_pos[h];
		uint32_t tf[6];
		sj_fold_window(a.tfold, pos, tf);
		for (uint32_t i = threadIdx.x; i < hv_count[h]; i += 256) {
			const uint32_t x = a.qlist[hv_head[h] + i];
			const int64_t x0 = a.q_begin + (int64_t)x - 16;
			uint32_t raw[8];
			__builtin_memcpy(raw, a.qfold + (x0 >> 1), 32);
			const uint32_t sh = (uint32_t)(x0 & 1) * 4;
			int mism = 0;
#pragma unroll
			for (int w = 0; w < 6; ++w) {
				const uint32_t d = tf[w] ^ __builtin_amdgcn_alignbit(raw[w + 1], raw[w], sh);
				mism += __builtin_popcount((((d & 0x77777777u) + 0x77777777u) | d) & 0x88888888u);
			}
			if (48 - mism < a.params.hamming_filter_id) continue;
			uint32_t qw[12], tw[12];
			__builtin_memcpy(qw, a.qdata + x0, 48);
			__builtin_memcpy(tw, a.tdata + pos - 16, 48);
			if (sj_window_identity(tw, qw) >= a.params.hamming_filter_id) survive(hv_slot[h], x, hv_pos[h]);
		}
	}
	__syncthreads();
	const unsigned n_sv = sv_n < SURV ? sv_n : SURV;
	if (n_sv) {
		if (threadIdx.x == 0) st_base = atomicAdd(a.survivor_count, (unsigned long long)n_sv);
		__syncthreads();
		for (unsigned k = threadIdx.x; k < n_sv; k += 256) {
			const unsigned long long idx = st_base + k;
			if (idx < (unsigned long long)a.survivor_cap) a.survivors[idx] = SeedSurvivor{ sv_slot[k], sv_x[k], base + (int64_t)sv_pos[k] };
		}
		__syncthreads();
	}
	const unsigned n_staged = st_n;
	if (n_staged == 0) return;
	if (threadIdx.x == 0) st_base = atomicAdd(a.matched_count, (unsigned long long)n_staged);
	__syncthreads();
	for (unsigned k = threadIdx.x; k < n_staged; k += 256) {
		const unsigned long long idx = st_base + k;
		if (idx < (unsigned long long)a.matched_cap) { a.matched_slot[idx] = st_slot[k]; a.matched_loc[idx] = base + (int64_t)st_pos[k]; }
	}
}

// the windows whose slab was full: one thread each, lists walked whole, appends straight to the global lists
__global__ __launch_bounds__(256) void seed_sj_overflow_kernel(SeedArgs a, SeedSjArgs j, int sid, int64_t n)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t key = j.overflow[2 * i];
	const int64_t pos = (int64_t)j.overflow[2 * i + 1];
	uint32_t tf[6];
	sj_fold_window(a.tfold, pos, tf);
	sj_join_window(a, key, seed_hash_a(key), pos, tf,
		[&](uint32_t slot, uint32_t, uint32_t) {
			const unsigned long long idx = atomicAdd(a.matched_count, 1ull);
			if (idx < (unsigned long long)a.matched_cap) { a.matched_slot[idx] = slot; a.matched_loc[idx] = pos; }
			return false;
		},
		[&](uint32_t slot, uint32_t x) {
			const unsigned long long idx = atomicAdd(a.survivor_count, 1ull);
			if (idx < (unsigned long long)a.survivor_cap) a.survivors[idx] = SeedSurvivor{ slot, x, pos };
		});
}

bool seed_sj_supported(const SeedParams& c)
{
	if (c.seed_encoding != SEED_SPACED) return false;
	for (int sid = 0; sid < c.n_shapes; ++sid)
		if (!seed_nibble_mode(c, sid) || c.shape_weight[sid] > 8) return false;
	return true;
}

int64_t seed_sj_workgroups(int64_t t_begin, int64_t t_end)
{
	const int64_t base = t_begin & ~(int64_t)15, threads = (t_end - base + 15) / 16;
	return (threads + 256 * SEED_SJ_TILES - 1) / (256 * SEED_SJ_TILES);
}

hipError_t launch_seed_sj_scatter(const SeedArgs& a, const SeedSjArgs& j, int sid, hipStream_t st)
{
	const SeedParams& c = a.params;
	uint64_t lo = 0, hi = 0;
	for (int l = 0; l < 32; ++l) {
		const uint64_t code = c.reduction[l] == L_MASK ? 15u : (uint64_t)c.reduction[l];
		(l < 16 ? lo : hi) |= code << ((l & 15) * 4);
	}
	uint64_t care64 = 0;
	for (int k = 0; k < c.shape_weight[sid]; ++k) care64 |= (uint64_t)15 << (4 * c.shape_pos[sid][k]);
	const int64_t base = a.t_begin & ~(int64_t)15;
	hipLaunchKernelGGL(seed_sj_scatter_kernel, dim3((unsigned)j.n_wg), dim3(256), 0, st, a, j, sid, lo, hi, base, care64);
	return hipGetLastError();
}

hipError_t launch_seed_sj_join(const SeedArgs& a, const SeedSjArgs& j, int sid, hipStream_t st)
{
	const int64_t base = a.t_begin & ~(int64_t)15;
	const int units = (int)((j.n_wg + SEED_SJ_GROUP - 1) / SEED_SJ_GROUP);
	hipLaunchKernelGGL(seed_sj_join_kernel, dim3((unsigned)(8 * 8 * units)), dim3(256), 0, st, a, j, sid, base, units);
	return hipGetLastError();
}

hipError_t launch_seed_sj_overflow(const SeedArgs& a, const SeedSjArgs& j, int sid, int64_t n, hipStream_t st)
{
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(seed_sj_overflow_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, j, sid, n);
	return hipGetLastError();
}

}  // namespace dmnd
