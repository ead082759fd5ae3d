// kd_sort.h -- the device-wide key sort of the GPU kd build (kd_build_gpu.hip): 64-bit keys (node slot << 32 | order-
// preserving bits of a coordinate), ascending on their low `key_bits` bits.  Round 2 called rocPRIM's radix_sort_keys here.
//
// A least-significant-digit radix sort, 8 bits per pass, three launches per pass, wave64 throughout:
//   k_kdsort_hist     every block counts the digits of its tile (4096 keys) in LDS and writes them digit-major:
//                     counts[digit * n_blocks + block]
//   k_kdsort_scan     block d: exclusive prefix sum over digit d's row in place + the row's total; the scatter blocks add the
//                     exclusive sum of the 256 totals themselves, so entry (d, b) is where block b's first key with digit d goes
//                     (all smaller digits of all blocks first, then the same digit of earlier blocks)
//   k_kdsort_scatter  the same tiles again.  Keys sit wave-striped (item i of lane l of wave w = w * 1024 + i * 64 + l), so the
//                     original order inside a wave is item-major; each item round ranks its 64 keys stably with a match-any
//                     built from 8 ballots (lanes holding the same digit find each other; rank = popcount of the peers below
//                     me) on top of a per-wave running count per digit in LDS; after the 16 rounds an exclusive sum over the 4
//                     waves per digit turns (wave, digit, rank) into the key's place.  Stable, so the passes compose.
// A pass in which every key has the same digit (common in the exponent bits of coordinates of one mesh) shows in the totals
// and is answered by a plain copy instead of the ranking.
// The sort only ever runs at commit time; it moves 24 bytes per key and pass and is sized for 10^5 ... 10^7 keys.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mipt {

constexpr int kKdSortBlock = 256;
constexpr int kKdSortWaves = kKdSortBlock / 64;
constexpr int kKdSortItems = 16;
constexpr int kKdSortTile = kKdSortBlock * kKdSortItems;  // 4096 keys per block

static __global__ __launch_bounds__(kKdSortBlock) void k_kdsort_hist(const unsigned long long *__restrict__ keys, int64_t n, int shift,
                                                                      unsigned int mask, unsigned int *__restrict__ counts, int n_blocks) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kKdSortTile;
#pragma unroll 4
    for (int i = 0; i < kKdSortItems; i++) {
        const int64_t idx = base + (int64_t)i * kKdSortBlock + threadIdx.x;
        if (idx < n) atomicAdd(&h[(unsigned int)(keys[idx] >> shift) & mask], 1u);
    }
    __syncthreads();
    counts[(size_t)threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];
}

// Block d: in-place exclusive scan of digit d's row counts[d * n_blocks .. +n_blocks) and the row's total in totals[d].
static __global__ __launch_bounds__(1024) void k_kdsort_scan(unsigned int *__restrict__ counts, int n_blocks, unsigned int *__restrict__ totals) {
    __shared__ unsigned int wave_sum[16];
    __shared__ unsigned int carry;
    unsigned int *row = counts + (size_t)blockIdx.x * n_blocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0u;
    __syncthreads();
    for (int base = 0; base < n_blocks; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const unsigned int v = i < n_blocks ? row[i] : 0u;
        unsigned int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int u = __shfl_up(incl, off, 64);
            if (lane >= off) incl += u;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        unsigned int before = carry;
        for (int w = 0; w < wave; w++) before += wave_sum[w];
        if (i < n_blocks) row[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

static __global__ __launch_bounds__(kKdSortBlock) void k_kdsort_scatter(const unsigned long long *__restrict__ in, unsigned long long *__restrict__ out,
                                                                         int64_t n, int shift, unsigned int mask,
                                                                         const unsigned int *__restrict__ offsets, int n_blocks,
                                                                         const unsigned int *__restrict__ totals) {
    __shared__ unsigned int cnt[kKdSortWaves][256];  // running count per (wave, digit); afterwards: where that wave's run starts
    __shared__ unsigned int digit_base[256];         // exclusive sum of the digits' totals
    __shared__ unsigned int scan_w[kKdSortWaves];
    __shared__ int uniform;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wave_base = (int64_t)blockIdx.x * kKdSortTile + (int64_t)wave * 64 * kKdSortItems;
    {
        if (threadIdx.x == 0) uniform = 0;
        __syncthreads();
        const unsigned int t = totals[threadIdx.x];
        if ((int64_t)t == n) uniform = 1;
        unsigned int incl = t;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int u = __shfl_up(incl, off, 64);
            if (lane >= off) incl += u;
        }
        if (lane == 63) scan_w[wave] = incl;
        __syncthreads();
        unsigned int before = 0u;
        for (int w = 0; w < wave; w++) before += scan_w[w];
        digit_base[threadIdx.x] = before + incl - t;
        __syncthreads();
    }
    if (uniform != 0) {  // every key has the same digit: the stable pass is the identity
#pragma unroll 4
        for (int i = 0; i < kKdSortItems; i++) {
            const int64_t idx = wave_base + (int64_t)i * 64 + lane;
            if (idx < n) out[idx] = in[idx];
        }
        return;
    }
    for (int d = threadIdx.x; d < kKdSortWaves * 256; d += kKdSortBlock) (&cnt[0][0])[d] = 0u;
    __syncthreads();
    unsigned long long key[kKdSortItems];
    unsigned int rank[kKdSortItems];
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < kKdSortItems; i++) {
        const int64_t idx = wave_base + (int64_t)i * 64 + lane;
        const bool valid = idx < n;
        key[i] = valid ? in[idx] : ~0ull;
        const unsigned int digit = (unsigned int)(key[i] >> shift) & mask;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long has = __ballot((digit >> b) & 1u);
            peers &= ((digit >> b) & 1u) ? has : ~has;
        }
        unsigned int prev = 0u;
        if (valid) prev = cnt[wave][digit];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & below) == 0ull) cnt[wave][digit] = prev + (unsigned int)__popcll(peers);  // the lowest peer
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        rank[i] = prev + (unsigned int)__popcll(peers & below);
    }
    __syncthreads();
    {   // thread d: exclusive sum of digit d over the waves, on top of the block's global offset for that digit
        const int d = threadIdx.x;
        unsigned int run = digit_base[d] + offsets[(size_t)d * n_blocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < kKdSortWaves; w++) {
            const unsigned int c = cnt[w][d];
            cnt[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kKdSortItems; i++) {
        const int64_t idx = wave_base + (int64_t)i * 64 + lane;
        if (idx < n) {
            const unsigned int digit = (unsigned int)(key[i] >> shift) & mask;
            out[(size_t)cnt[wave][digit] + rank[i]] = key[i];
        }
    }
}

// scratch: (256 * n_blocks + 256) uint32, n_blocks = ceil(n / 4096).  Sorts on bits [0, key_bits) of the keys; `a` holds the
// input and is overwritten, the sorted keys end up in whichever of a / b the return value points to.
inline size_t kd_sort_scratch_bytes(int64_t n) {
    const int64_t n_blocks = (n + kKdSortTile - 1) / kKdSortTile;
    return (size_t)(256 * n_blocks + 256) * sizeof(unsigned int);
}
inline unsigned long long *launch_kd_sort(hipStream_t st, unsigned long long *a, unsigned long long *b, int64_t n, int key_bits, void *scratch) {
    if (n <= 1) return a;
    const int n_blocks = (int)((n + kKdSortTile - 1) / kKdSortTile);
    unsigned int *counts = static_cast<unsigned int *>(scratch);
    unsigned int *totals = counts + (size_t)256 * n_blocks;
    unsigned long long *src = a, *dst = b;
    for (int shift = 0; shift < key_bits; shift += 8) {
        const int bits = key_bits - shift < 8 ? key_bits - shift : 8;
        const unsigned int mask = (1u << bits) - 1u;
        hipLaunchKernelGGL(k_kdsort_hist, dim3((unsigned)n_blocks), dim3(kKdSortBlock), 0, st, src, n, shift, mask, counts, n_blocks);
        hipLaunchKernelGGL(k_kdsort_scan, dim3(256), dim3(1024), 0, st, counts, n_blocks, totals);
        hipLaunchKernelGGL(k_kdsort_scatter, dim3((unsigned)n_blocks), dim3(kKdSortBlock), 0, st, src, dst, n, shift, mask, counts, n_blocks, totals);
        unsigned long long *t = src; src = dst; dst = t;
    }
    return src;
}

}  // namespace mipt
