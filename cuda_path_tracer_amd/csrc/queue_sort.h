// queue_sort.h -- binning of a ray queue by its direction keys (gfx950, wave64).  No counterpart in the reference (the
// megakernel kernel.cu:310-346 has no queues); it replaces the library radix sort rounds 1-2 used in front of the extend
// kernel.  The order of a queue changes no result -- only how coherent the 64 rays of a wave are -- so this is NOT a general
// sort: it is neither stable nor total, and it takes its length from device memory (no host-sized launch, no read-back).
//
// Two passes over (key, value) pairs, most significant digits first, 21 key bits in all:
//   k_sort_hist   LDS histogram of the top 10 key bits per block, flushed with one global atomic per non-empty bucket
//   k_sort_scan   one block: bucket starts (exclusive scan), the buckets' cursors, and the pass-2 work list (buckets cut into
//                 chunks of at most kSortChunk = 16384 entries: one huge bucket cannot serialise the pass, and a chunk's
//                 values fit in LDS)
//   k_sort_part1  tiles of 8192 entries: LDS histogram of the tile, ONE global atomic per (tile, bucket) reserves the run,
//                 the tile is ordered by bucket in LDS and written out run by run (coalesced) to the intermediate arrays
//   k_sort_part2  one block per chunk: LDS counting sort on the next 11 bits; only the VALUES are ordered (in LDS) and
//                 written, coalesced, over the chunk's own range
// A bucket larger than a chunk comes out as several sorted runs, which is as good for the consumer (a wave takes 64
// consecutive entries; 64 entries of a chunk span 2^-18 of the key space).  Lanes of a wave that all hold the same digit
// (runs of unbinned entries with the largest key, fully coherent bundles) are counted with one LDS atomic per wave
// instead of 64 conflicting ones.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mipt {

constexpr int kSortBits1 = 10, kSortBits2 = 11;
constexpr int kSortBuckets = 1 << kSortBits1, kSortBins = 1 << kSortBits2;
constexpr int kSortBlock = 512;
constexpr int kSortItems = 16;                       // per thread and tile
constexpr int kSortTile = kSortBlock * kSortItems;   // 8192
constexpr int kSortChunk = 1 << 14;                  // pass-2 work unit: ordered in LDS (64 KiB of values)
constexpr int kSortBatch = 8;                        // loads in flight per thread in the streaming loops
// scratch (ints): the bucket histogram / cursors, ONE PER 128-BYTE LINE (atomics on neighbouring words of a line serialise
// like same-address ones: with the 1024 cursors packed into 32 lines k_sort_part1 spent its time queueing on them); then the
// bucket starts [1025] and the chunk prefix [1025]
constexpr int kSortCursorStride = 32;
constexpr int kSortScratchStart = kSortBuckets * kSortCursorStride;
constexpr int kSortScratchChunks = kSortScratchStart + kSortBuckets + 64;
constexpr int kSortScratchInts = kSortScratchChunks + kSortBuckets + 64;

#ifdef MIPT_SORT_PROF
__device__ unsigned long long g_sort_prof[32];
#define SORT_T(var) const unsigned long long var = wall_clock64()
#define SORT_ACC(slot, a, b) do { if (threadIdx.x == 0) atomicAdd(&g_sort_prof[slot], (b) - (a)); } while (0)
#else
#define SORT_T(var)
#define SORT_ACC(slot, a, b)
#endif

__device__ __forceinline__ int sort_lane_rank(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
}

// old value of counter[d] for this lane's entry, counting every live lane of the wave once: one LDS atomic per wave when all
// live lanes agree on d, else one per lane.  Convergent within the wave.
__device__ __forceinline__ int sort_lds_take(int *counter, int d, bool live) {
    const unsigned long long act = __ballot(live);
    if (act == 0ull) return 0;
    const int leader = __ffsll((long long)act) - 1;
    const int d0 = __builtin_amdgcn_readlane(d, leader);
    if (__ballot(live && d == d0) == act) {
        int base = 0;
        if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(&counter[d0], __popcll(act));
        base = __builtin_amdgcn_readlane(base, leader);
        return base + sort_lane_rank(act);
    }
    return live ? atomicAdd(&counter[d], 1) : 0;
}

static __global__ __launch_bounds__(kSortBlock) void k_sort_hist(const uint32_t *__restrict__ keys, const int *__restrict__ count, int shift1,
                                                                 int *__restrict__ hist) {
    __shared__ int cnt[kSortBuckets];
    for (int b = threadIdx.x; b < kSortBuckets; b += kSortBlock) cnt[b] = 0;
    __syncthreads();
    const int n = *count;
    for (int i0 = blockIdx.x * (kSortBlock * kSortBatch); i0 < n; i0 += gridDim.x * (kSortBlock * kSortBatch)) {
        uint32_t k[kSortBatch];
#pragma unroll
        for (int j = 0; j < kSortBatch; j++) {
            const int i = i0 + j * kSortBlock + (int)threadIdx.x;
            k[j] = i < n ? keys[i] : 0u;
        }
#pragma unroll
        for (int j = 0; j < kSortBatch; j++) {
            const int i = i0 + j * kSortBlock + (int)threadIdx.x;
            (void)sort_lds_take(cnt, (int)((k[j] >> shift1) & (kSortBuckets - 1)), i < n);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kSortBuckets; b += kSortBlock) {
        const int c = cnt[b];
        if (c > 0) atomicAdd(&hist[b * kSortCursorStride], c);
    }
}

// exclusive scan over the 1024 threads of a block (v >= 0); `lds` holds 16 ints; returns the exclusive prefix, total in *total
__device__ __forceinline__ int sort_block_excl_scan(int v, int *lds, int *total) {
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    int before = 0, all = 0;
    for (int w = 0; w < 16; w++) {
        const int c = lds[w];
        if (w < wave) before += c;
        all += c;
    }
    __syncthreads();
    *total = all;
    return before + incl - v;
}

static __global__ __launch_bounds__(kSortBuckets) void k_sort_scan(int *__restrict__ scratch) {
    __shared__ int lds[16];
    const int t = (int)threadIdx.x;
    const int c = scratch[t * kSortCursorStride];
    int total;
    const int start = sort_block_excl_scan(c, lds, &total);
    scratch[t * kSortCursorStride] = start;  // the bucket's cursor for k_sort_part1
    scratch[kSortScratchStart + t] = start;  // its start, kept for k_sort_part2
    const int chunks = (c + kSortChunk - 1) / kSortChunk;
    int total_chunks;
    const int cstart = sort_block_excl_scan(chunks, lds, &total_chunks);
    scratch[kSortScratchChunks + t] = cstart;
    if (t == kSortBuckets - 1) {
        scratch[kSortScratchStart + kSortBuckets] = total;
        scratch[kSortScratchChunks + kSortBuckets] = total_chunks;
    }
}

// exclusive scan of `per * kSortBlock` LDS counters in place (thread t owns counters [t * per, (t + 1) * per)); wsum: 8 ints.
// Ends with a barrier.
template <int per>
__device__ __forceinline__ void sort_lds_scan(int *cnt, int *wsum) {
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    int v[per], s = 0;
#pragma unroll
    for (int q = 0; q < per; q++) { v[q] = cnt[(int)threadIdx.x * per + q]; s += v[q]; }
    int incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int q = 0; q < wave; q++) before += wsum[q];
    int run = before + incl - s;
#pragma unroll
    for (int q = 0; q < per; q++) { cnt[(int)threadIdx.x * per + q] = run; run += v[q]; }
    __syncthreads();
}

// Scattered 4-byte global stores cost ~3.5 clocks per LANE and CU on MI355X (64 lines per wave instruction): written straight
// to their runs, the pairs of a tile took 45 us per tile and the values of a chunk 115 us.  Both passes therefore order their
// tile / chunk in LDS first (scattered ds_write, ~20x cheaper) and write it out with consecutive lanes on consecutive
// addresses of a run.
static __global__ __launch_bounds__(kSortBlock, 2) void k_sort_part1(const uint32_t *__restrict__ keys, const int *__restrict__ vals,
                                                                     const int *__restrict__ count, int shift1, int *__restrict__ cursor,
                                                                     uint32_t *__restrict__ keys_mid, int *__restrict__ vals_mid) {
    __shared__ int cnt[kSortBuckets];    // counts, then running ranks
    __shared__ int first[kSortBuckets];  // start of the bucket's run inside the ordered tile
    __shared__ int base[kSortBuckets];   // start of the run reserved for it in the bucket (global)
    __shared__ int wsum[kSortBlock / 64];
    __shared__ uint32_t stage_k[kSortTile];
    __shared__ int stage_v[kSortTile];
    const int n = *count;
    const int n_tiles = (n + kSortTile - 1) / kSortTile;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        SORT_T(t0);
        for (int b = threadIdx.x; b < kSortBuckets; b += kSortBlock) cnt[b] = 0;
        __syncthreads();
        uint32_t k[kSortItems];
        int v[kSortItems];
        const int i0 = tile * kSortTile + (int)threadIdx.x;
        const int tile_n = n - tile * kSortTile < kSortTile ? n - tile * kSortTile : kSortTile;
#pragma unroll
        for (int j = 0; j < kSortItems; j++) {  // every load of the tile in flight at once
            const int i = i0 + j * kSortBlock;
            k[j] = i < n ? keys[i] : 0u;
            v[j] = i < n ? vals[i] : 0;
        }
#pragma unroll
        for (int j = 0; j < kSortItems; j++)
            (void)sort_lds_take(cnt, (int)((k[j] >> shift1) & (kSortBuckets - 1)), i0 + j * kSortBlock < n);
        __syncthreads();
        SORT_T(t1);
        for (int b = threadIdx.x; b < kSortBuckets; b += kSortBlock) {
            const int c = cnt[b];
            base[b] = c > 0 ? atomicAdd(&cursor[b * kSortCursorStride], c) : 0;
        }
        sort_lds_scan<kSortBuckets / kSortBlock>(cnt, wsum);
        for (int b = threadIdx.x; b < kSortBuckets; b += kSortBlock) first[b] = cnt[b];
        __syncthreads();
        SORT_T(t2);
#pragma unroll
        for (int j = 0; j < kSortItems; j++) {
            const bool live = i0 + j * kSortBlock < n;
            const int r = sort_lds_take(cnt, (int)((k[j] >> shift1) & (kSortBuckets - 1)), live);  // cnt[] runs on from first[]
            if (live) {
                stage_k[r] = k[j];
                stage_v[r] = v[j];
            }
        }
        __syncthreads();
        for (int p = (int)threadIdx.x; p < tile_n; p += kSortBlock) {
            const uint32_t key = stage_k[p];
            const int d = (int)((key >> shift1) & (kSortBuckets - 1));
            const int pos = base[d] + (p - first[d]);
            keys_mid[pos] = key;
            vals_mid[pos] = stage_v[p];
        }
        __syncthreads();
        SORT_T(t3);
        SORT_ACC(8, t0, t1); SORT_ACC(9, t1, t2); SORT_ACC(10, t2, t3); SORT_ACC(11, t3, t3 + 1);
    }
}

static __global__ __launch_bounds__(kSortBlock, 2) void k_sort_part2(const uint32_t *__restrict__ keys_mid, const int *__restrict__ vals_mid,
                                                                     const int *__restrict__ scratch, int shift2, int *__restrict__ out) {
    __shared__ int cnt[kSortBins];
    __shared__ int wsum[kSortBlock / 64];
    __shared__ int stage_v[kSortChunk];
    const int *start = scratch + kSortScratchStart, *cprefix = scratch + kSortScratchChunks;
    const int n_chunks = cprefix[kSortBuckets];
    for (int w = blockIdx.x; w < n_chunks; w += gridDim.x) {
        // the bucket of work item w: the last b with cprefix[b] <= w (empty buckets have no chunk and are never the answer)
        SORT_T(t0);
        int lo = 0, hi = kSortBuckets - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (cprefix[mid] <= w) lo = mid; else hi = mid - 1;
        }
        const int b_begin = start[lo], b_end = start[lo + 1];
        const int c_begin = b_begin + (w - cprefix[lo]) * kSortChunk;
        const int c_end = c_begin + kSortChunk < b_end ? c_begin + kSortChunk : b_end;
        for (int b = threadIdx.x; b < kSortBins; b += kSortBlock) cnt[b] = 0;
        __syncthreads();
        SORT_T(t1);
        for (int i0 = c_begin; i0 < c_end; i0 += kSortBlock * kSortBatch) {
            uint32_t k[kSortBatch];
#pragma unroll
            for (int j = 0; j < kSortBatch; j++) {
                const int i = i0 + j * kSortBlock + (int)threadIdx.x;
                k[j] = i < c_end ? keys_mid[i] : 0u;
            }
#pragma unroll
            for (int j = 0; j < kSortBatch; j++)
                (void)sort_lds_take(cnt, (int)((k[j] >> shift2) & (kSortBins - 1)), i0 + j * kSortBlock + (int)threadIdx.x < c_end);
        }
        __syncthreads();
        SORT_T(t2);
        sort_lds_scan<kSortBins / kSortBlock>(cnt, wsum);
        SORT_T(t3);
        for (int i0 = c_begin; i0 < c_end; i0 += kSortBlock * kSortBatch) {
            uint32_t k[kSortBatch];
            int v[kSortBatch];
#pragma unroll
            for (int j = 0; j < kSortBatch; j++) {
                const int i = i0 + j * kSortBlock + (int)threadIdx.x;
                k[j] = i < c_end ? keys_mid[i] : 0u;
                v[j] = i < c_end ? vals_mid[i] : 0;
            }
#pragma unroll
            for (int j = 0; j < kSortBatch; j++) {
                const bool live = i0 + j * kSortBlock + (int)threadIdx.x < c_end;
                const int r = sort_lds_take(cnt, (int)((k[j] >> shift2) & (kSortBins - 1)), live);
                if (live) stage_v[r] = v[j];
            }
        }
        __syncthreads();
        for (int p = (int)threadIdx.x; p < c_end - c_begin; p += kSortBlock) out[c_begin + p] = stage_v[p];
        __syncthreads();
        SORT_T(t4);
        SORT_ACC(0, t0, t1); SORT_ACC(1, t1, t2); SORT_ACC(2, t2, t3); SORT_ACC(3, t3, t4); SORT_ACC(4, t4, t4 + 1);
    }
}

// Bins vals[0, *count) by the top 21 of the `key_bits` (>= 21) significant bits of keys[] into out[0, *count), on `stream`.
// keys_mid / vals_mid: intermediate arrays of the queue's capacity; scratch: kSortScratchInts ints.  Nothing here depends on
// the queue's length on the host: the grids are fixed and every kernel reads *count.
inline void launch_queue_sort(hipStream_t stream, int num_cus, const uint32_t *keys, const int *vals, const int *count, unsigned key_bits,
                              uint32_t *keys_mid, int *vals_mid, int *scratch, int *out) {
    const int shift1 = (int)key_bits - kSortBits1, shift2 = (int)key_bits - kSortBits1 - kSortBits2;
    (void)hipMemsetAsync(scratch, 0, (size_t)kSortBuckets * kSortCursorStride * sizeof(int), stream);
    hipLaunchKernelGGL(k_sort_hist, dim3(num_cus * 4), dim3(kSortBlock), 0, stream, keys, count, shift1, scratch);
    hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(kSortBuckets), 0, stream, scratch);
    hipLaunchKernelGGL(k_sort_part1, dim3(num_cus * 2), dim3(kSortBlock), 0, stream, keys, vals, count, shift1, scratch, keys_mid, vals_mid);
    hipLaunchKernelGGL(k_sort_part2, dim3(num_cus * 2), dim3(kSortBlock), 0, stream, keys_mid, vals_mid, scratch, shift2 > 0 ? shift2 : 0, out);
}

}  // namespace mipt
