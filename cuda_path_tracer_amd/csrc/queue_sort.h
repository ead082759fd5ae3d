// queue_sort.h -- binning of a ray queue by its direction keys (gfx950, wave64).  No counterpart in the reference (the
// megakernel kernel.cu:310-346 has no queues); it replaces the library radix sort rounds 1-2 used in front of the extend
// kernel.  The order of a queue changes no result -- only how coherent the 64 rays of a wave are -- so this is NOT a general
// sort: it is not stable, entries with (nearly) equal keys come out in no particular order, and it takes its length from
// device memory (no host-sized launch, no read-back).
//
// The extend kernel is sensitive to the fine order (1080p frame: a queue ordered by 21 / 18 of the 24 key bits costs it
// 1 % / 4 %) and the keys of a real queue are far from uniform (shadow rays of a frame point into a small part of the
// direction map), so the two passes are a SAMPLE SORT with splitters taken from the queue's own histogram:
//   k_sort_hist   LDS histogram of the top 14 key bits ("fine bins"), a few blocks, flushed with global atomics
//   k_sort_scan   one block: 1024 buckets = runs of consecutive fine bins holding ~n/1024 entries each (a bucket never
//                 cuts a fine bin); the fine bin -> bucket table; per bucket its start, its cursor, its first key and the
//                 shift that maps its key range onto the 8192 counters of pass 2; the pass-2 work list (a bucket, or slices
//                 of 2^18 entries of one that a single crowded fine bin made larger)
//   k_sort_part1  tiles of 6144 entries: bucket of every entry through the table, LDS histogram of the tile, ONE global
//                 atomic per (tile, bucket) reserves the run, the tile is ordered by bucket in LDS and written out run by
//                 run (coalesced) to the intermediate arrays
//   k_sort_part2  one 1024-thread block per bucket: LDS counting sort on (key - first key) >> shift; only the VALUES are
//                 placed -- ordered in LDS, 24576 at a time, and written coalesced over the bucket's own range
// A crowded region of the key space gets buckets one fine bin wide, sorted down to the last key bit; a sparse one gets wide
// buckets and a coarser order (its rays are far apart anyway).  Lanes of a wave that all hold the same digit (runs of
// unbinned entries with the largest key, fully coherent bundles) are counted with one LDS atomic per wave instead of 64
// conflicting ones.
//
// Measured and rejected on the way (45 M pairs, MI355X; the library sort: 1.08 ms): one pass of returning global atomics
// per entry + scatter (count 1.8 ms + scatter 1.4 ms); fixed buckets of the top 10 bits with pass 2 on slices of 16384
// values (0.78 ms, but a bucket of the real frame came out as many interleaved runs: extend kernel 5 % slower) or with
// scattered 4-byte stores over a whole bucket (the L2 takes ~15-30 ps per partial write: 1.4 ms); fixed buckets with the
// round-based pass 2 below (0.77 ms on uniform keys, 1.5 ms per sort on the real frame, whose crowded buckets held
// hundreds of thousands of entries).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mipt {

constexpr int kSortFineBits = 14, kSortFine = 1 << kSortFineBits;  // splitter resolution
constexpr int kSortBuckets = 1024;
constexpr int kSortBits2 = 13, kSortBins = 1 << kSortBits2;        // counters of pass 2
constexpr int kSortBlock = 512;
constexpr int kSortItems = 12;                       // per thread and tile
constexpr int kSortTile = kSortBlock * kSortItems;   // 6144: key + value + bucket staged in 60 KiB, two blocks per CU
constexpr int kSortChunk = 1 << 18;                  // pass-2 work unit: a bucket, or a slice of one that is larger than this
constexpr int kSortBlock2 = 1024;                    // histogram, scan and pass 2: 1024-thread blocks
constexpr int kSortStage = 24576;                    // values ordered in LDS per round of pass 2 (96 KiB)
constexpr int kSortBatch = 8;                        // loads in flight per thread in the streaming loops
// scratch (ints).  The cursors sit ONE PER 128-BYTE LINE: atomics on neighbouring words of a line serialise like
// same-address ones (with the 1024 cursors packed into 32 lines k_sort_part1 spent its time queueing on them).
constexpr int kSortCursorStride = 32;
constexpr int kSortScratchFine = 0;                                                  // [16384] fine-bin histogram
constexpr int kSortScratchCursor = kSortScratchFine + kSortFine;                     // [1024 * 32] bucket cursors
constexpr int kSortScratchStart = kSortScratchCursor + kSortBuckets * kSortCursorStride;  // [1025] bucket starts
constexpr int kSortScratchChunks = kSortScratchStart + kSortBuckets + 64;            // [1025] work-list prefix
constexpr int kSortScratchKey0 = kSortScratchChunks + kSortBuckets + 64;             // [1024] first key of the bucket
constexpr int kSortScratchShift = kSortScratchKey0 + kSortBuckets;                   // [1024] key shift of pass 2
constexpr int kSortScratchTable = kSortScratchShift + kSortBuckets;                  // [16384 uint16] fine bin -> bucket
constexpr int kSortScratchInts = kSortScratchTable + kSortFine / 2 + 64;
constexpr int kSortScratchZeroInts = kSortScratchFine + kSortFine;                   // what a sort must find zeroed

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
// live lanes agree on d, else one per lane.  Convergent within the wave.  (Peeling large groups of equal digits off the wave
// one LDS atomic each -- a real queue arrives in bundles -- was measured: 8 % SLOWER on the 1080p frame's queues, 30 % on
// clumped synthetic keys; the conflicting per-lane atomics are cheaper than the ballots.)
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

static __global__ __launch_bounds__(kSortBlock2, 1) void k_sort_hist(const uint32_t *__restrict__ keys, const int *__restrict__ count,
                                                                     int shift_fine, int *__restrict__ fine) {
    __shared__ int cnt[kSortFine];
    for (int b = threadIdx.x; b < kSortFine; b += kSortBlock2) cnt[b] = 0;
    __syncthreads();
    const int n = *count;
    for (int i0 = blockIdx.x * (kSortBlock2 * kSortBatch); i0 < n; i0 += gridDim.x * (kSortBlock2 * kSortBatch)) {
        uint32_t k[kSortBatch];
#pragma unroll
        for (int j = 0; j < kSortBatch; j++) {
            const int i = i0 + j * kSortBlock2 + (int)threadIdx.x;
            k[j] = i < n ? keys[i] : 0u;
        }
#pragma unroll
        for (int j = 0; j < kSortBatch; j++) {
            const int i = i0 + j * kSortBlock2 + (int)threadIdx.x;
            (void)sort_lds_take(cnt, (int)((k[j] >> shift_fine) & (kSortFine - 1)), i < n);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kSortFine; b += kSortBlock2) {
        const int c = cnt[b];
        if (c > 0) atomicAdd(&fine[b], c);
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

// Splitters.  Thread t owns the 16 fine bins [16 t, 16 t + 16); bucket of a fine bin = (entries in front of it) / T with
// T = ceil(n / 1024): consecutive fine bins share a bucket until it holds T entries.  key_low_bits = key bits below the
// fine bins.
static __global__ __launch_bounds__(kSortBlock2) void k_sort_scan(int *__restrict__ scratch, int key_low_bits) {
    __shared__ int lds[16];
    __shared__ int bcount[kSortBuckets], bmin[kSortBuckets], bmax[kSortBuckets];
    const int t = (int)threadIdx.x;
    constexpr int per = kSortFine / kSortBlock2;
    int c[per], s = 0;
#pragma unroll
    for (int q = 0; q < per; q++) { c[q] = scratch[kSortScratchFine + t * per + q]; s += c[q]; }
    bcount[t] = 0; bmin[t] = kSortFine; bmax[t] = -1;
    int n;
    int run = sort_block_excl_scan(s, lds, &n);  // (its barriers also publish the three arrays)
    const int T = n > kSortBuckets ? (n + kSortBuckets - 1) / kSortBuckets : 1;
    uint16_t *table = reinterpret_cast<uint16_t *>(scratch + kSortScratchTable);
#pragma unroll
    for (int q = 0; q < per; q++) {
        const int f = t * per + q;
        int b = run / T;
        if (b > kSortBuckets - 1) b = kSortBuckets - 1;
        table[f] = (uint16_t)b;
        if (c[q] > 0) {
            atomicAdd(&bcount[b], c[q]);
            atomicMin(&bmin[b], f);
            atomicMax(&bmax[b], f);
        }
        run += c[q];
    }
    __syncthreads();
    const int bc = bcount[t];
    int total;
    const int start = sort_block_excl_scan(bc, lds, &total);
    scratch[kSortScratchCursor + t * kSortCursorStride] = start;  // the bucket's cursor for k_sort_part1
    scratch[kSortScratchStart + t] = start;                       // its start, kept for k_sort_part2
    const int chunks = (bc + kSortChunk - 1) / kSortChunk;
    int total_chunks;
    const int cstart = sort_block_excl_scan(chunks, lds, &total_chunks);
    scratch[kSortScratchChunks + t] = cstart;
    // pass 2 sorts the bucket by (key - first key) >> shift; the bucket spans fine bins [bmin, bmax]
    int shift = 0;
    uint32_t key0 = 0u;
    if (bc > 0) {
        const int range = bmax[t] - bmin[t] + 1;        // fine bins
        int lg = 0;
        while ((1 << lg) < range) lg++;                 // ceil(log2(range))
        shift = lg + key_low_bits - kSortBits2;
        if (shift < 0) shift = 0;
        key0 = (uint32_t)bmin[t] << key_low_bits;
    }
    scratch[kSortScratchKey0 + t] = (int)key0;
    scratch[kSortScratchShift + t] = shift;
    if (t == kSortBuckets - 1) {
        scratch[kSortScratchStart + kSortBuckets] = total;
        scratch[kSortScratchChunks + kSortBuckets] = total_chunks;
    }
}

// exclusive scan of `per * blockDim.x` LDS counters in place (thread t owns counters [t * per, (t + 1) * per)); wsum: one int
// per wave.  Ends with a barrier.
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
// to their runs, the pairs of a tile took 45 us per tile.  Pass 1 therefore orders its tile in LDS first (scattered ds_write,
// ~20x cheaper) and writes it out with consecutive lanes on consecutive addresses of a run.
static __global__ __launch_bounds__(kSortBlock, 2) void k_sort_part1(const uint32_t *__restrict__ keys, const int *__restrict__ vals,
                                                                     const int *__restrict__ count, int shift_fine, int *__restrict__ scratch,
                                                                     uint32_t *__restrict__ keys_mid, int *__restrict__ vals_mid) {
    __shared__ int cnt[kSortBuckets];    // counts, then running ranks
    __shared__ int first[kSortBuckets];  // start of the bucket's run inside the ordered tile
    __shared__ int base[kSortBuckets];   // start of the run reserved for it in the bucket (global)
    __shared__ int wsum[kSortBlock / 64];
    __shared__ uint32_t stage_k[kSortTile];
    __shared__ int stage_v[kSortTile];
    __shared__ uint16_t stage_b[kSortTile];
    const uint16_t *__restrict__ table = reinterpret_cast<const uint16_t *>(scratch + kSortScratchTable);
    int *cursor = scratch + kSortScratchCursor;
    const int n = *count;
    const int n_tiles = (n + kSortTile - 1) / kSortTile;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        SORT_T(t0);
        for (int b = threadIdx.x; b < kSortBuckets; b += kSortBlock) cnt[b] = 0;
        __syncthreads();
        uint32_t k[kSortItems];
        int v[kSortItems];
        int bk[kSortItems];
        const int i0 = tile * kSortTile + (int)threadIdx.x;
        const int tile_n = n - tile * kSortTile < kSortTile ? n - tile * kSortTile : kSortTile;
#pragma unroll
        for (int j = 0; j < kSortItems; j++) {  // every load of the tile in flight at once
            const int i = i0 + j * kSortBlock;
            k[j] = i < n ? keys[i] : 0u;
            v[j] = i < n ? vals[i] : 0;
        }
#pragma unroll
        for (int j = 0; j < kSortItems; j++) bk[j] = (int)table[(k[j] >> shift_fine) & (kSortFine - 1)];  // 32 KB table: L1 / L2
#pragma unroll
        for (int j = 0; j < kSortItems; j++) (void)sort_lds_take(cnt, bk[j], i0 + j * kSortBlock < n);
        __syncthreads();
        SORT_T(t1);
        // reserve the runs: the atomics' answers are only needed for the write-out, so they stay in flight (in registers)
        // while the tile is being ordered
        int reserved[kSortBuckets / kSortBlock];
#pragma unroll
        for (int q = 0; q < kSortBuckets / kSortBlock; q++) {
            const int b = (int)threadIdx.x + q * kSortBlock;
            const int c = cnt[b];
            reserved[q] = c > 0 ? atomicAdd(&cursor[b * kSortCursorStride], c) : 0;
        }
        sort_lds_scan<kSortBuckets / kSortBlock>(cnt, wsum);
        for (int b = threadIdx.x; b < kSortBuckets; b += kSortBlock) first[b] = cnt[b];
        __syncthreads();
        SORT_T(t2);
#pragma unroll
        for (int j = 0; j < kSortItems; j++) {
            const bool live = i0 + j * kSortBlock < n;
            const int r = sort_lds_take(cnt, bk[j], live);  // cnt[] runs on from first[]
            if (live) {
                stage_k[r] = k[j];
                stage_v[r] = v[j];
                stage_b[r] = (uint16_t)bk[j];
            }
        }
#pragma unroll
        for (int q = 0; q < kSortBuckets / kSortBlock; q++) base[(int)threadIdx.x + q * kSortBlock] = reserved[q];
        __syncthreads();
        for (int p = (int)threadIdx.x; p < tile_n; p += kSortBlock) {
            const int d = (int)stage_b[p];
            const int pos = base[d] + (p - first[d]);
            keys_mid[pos] = stage_k[p];
            vals_mid[pos] = stage_v[p];
        }
        __syncthreads();
        SORT_T(t3);
        SORT_ACC(8, t0, t1); SORT_ACC(9, t1, t2); SORT_ACC(10, t2, t3); SORT_ACC(11, t3, t3 + 1);
    }
}

// Pass 2: one 1024-thread block per bucket (or slice of a huge one).  Count (key - first key) >> shift in LDS, scan, then
// place the VALUES: scattered 4-byte stores straight to global memory measured 0.7 ms per 45 M entries, so the values are
// ordered in LDS and written coalesced -- in ROUNDS when the bucket is larger than the 24576 values the stage holds: a round
// takes the bins whose positions fit the stage, and every round sweeps the bucket's pairs again (they come from L2).  A
// single bin larger than the stage (equal keys: order immaterial) is written directly in arrival order.
static __global__ __launch_bounds__(kSortBlock2, 1) void k_sort_part2(const uint32_t *__restrict__ keys_mid, const int *__restrict__ vals_mid,
                                                                      const int *__restrict__ scratch, int *__restrict__ out) {
    __shared__ int cnt[kSortBins + 1];
    __shared__ int wsum[kSortBlock2 / 64];
    __shared__ int stage_v[kSortStage];
    __shared__ int round_hi;
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
        const uint32_t key0 = (uint32_t)scratch[kSortScratchKey0 + lo];
        const int shift = scratch[kSortScratchShift + lo];
        const int c_begin = b_begin + (w - cprefix[lo]) * kSortChunk;
        const int c_end = c_begin + kSortChunk < b_end ? c_begin + kSortChunk : b_end;
        const int m = c_end - c_begin;
        auto digit = [&](uint32_t key) {
            const uint32_t d = (key - key0) >> shift;
            return (int)(d < (uint32_t)kSortBins ? d : (uint32_t)(kSortBins - 1));
        };
        for (int b = threadIdx.x; b < kSortBins; b += kSortBlock2) cnt[b] = 0;
        __syncthreads();
        SORT_T(t1);
        for (int i0 = c_begin; i0 < c_end; i0 += kSortBlock2 * kSortBatch) {
            uint32_t k[kSortBatch];
#pragma unroll
            for (int j = 0; j < kSortBatch; j++) {
                const int i = i0 + j * kSortBlock2 + (int)threadIdx.x;
                k[j] = i < c_end ? keys_mid[i] : key0;
            }
#pragma unroll
            for (int j = 0; j < kSortBatch; j++) (void)sort_lds_take(cnt, digit(k[j]), i0 + j * kSortBlock2 + (int)threadIdx.x < c_end);
        }
        __syncthreads();
        SORT_T(t2);
        sort_lds_scan<kSortBins / kSortBlock2>(cnt, wsum);  // cnt[b] = first position of bin b
        if (threadIdx.x == 0) cnt[kSortBins] = m;
        __syncthreads();
        SORT_T(t3);
        int bin_lo = 0, pos_lo = 0;  // bins below bin_lo are placed (their counters have run on to the next bin's start)
        while (pos_lo < m) {         // (block-uniform loop)
            // this round: bins [bin_lo, bin_hi) with positions [pos_lo, pos_hi), pos_hi - pos_lo <= kSortStage -- or one oversized bin
            if (threadIdx.x == 0) {
                int a = bin_lo, z = kSortBins;  // largest bin index in [bin_lo, kSortBins] whose start still fits the stage
                while (a < z) {
                    const int mid = (a + z + 1) >> 1;
                    if (cnt[mid] <= pos_lo + kSortStage) a = mid; else z = mid - 1;
                }
                round_hi = a;
            }
            __syncthreads();
            int bin_hi = round_hi;
            const bool direct = bin_hi == bin_lo;  // bin_lo alone is larger than the stage
            if (direct) bin_hi = bin_lo + 1;
            const int pos_hi = cnt[bin_hi];
            __syncthreads();  // (everybody has read cnt[bin_hi] before the counters move)
            for (int i0 = c_begin; i0 < c_end; i0 += kSortBlock2 * kSortBatch) {
                uint32_t k[kSortBatch];
                int v[kSortBatch];
#pragma unroll
                for (int j = 0; j < kSortBatch; j++) {
                    const int i = i0 + j * kSortBlock2 + (int)threadIdx.x;
                    k[j] = i < c_end ? keys_mid[i] : key0;
                    v[j] = i < c_end ? vals_mid[i] : 0;
                }
#pragma unroll
                for (int j = 0; j < kSortBatch; j++) {
                    const int d = digit(k[j]);
                    const bool mine = i0 + j * kSortBlock2 + (int)threadIdx.x < c_end && d >= bin_lo && d < bin_hi;
                    const int r = sort_lds_take(cnt, d, mine);
                    if (mine) {
                        if (direct) out[c_begin + r] = v[j];
                        else stage_v[r - pos_lo] = v[j];
                    }
                }
            }
            __syncthreads();
            if (!direct)
                for (int p = (int)threadIdx.x; p < pos_hi - pos_lo; p += kSortBlock2) out[c_begin + pos_lo + p] = stage_v[p];
            __syncthreads();
            bin_lo = bin_hi;
            pos_lo = pos_hi;
        }
        SORT_T(t4);
        SORT_ACC(0, t0, t1); SORT_ACC(1, t1, t2); SORT_ACC(2, t2, t3); SORT_ACC(3, t3, t4); SORT_ACC(4, t4, t4 + 1);
    }
}

// Bins vals[0, *count) by the `key_bits` (14..32) significant bits of keys[] into out[0, *count), on `stream`.
// keys_mid / vals_mid: intermediate arrays of the queue's capacity; scratch: kSortScratchInts ints.  Nothing here depends on
// the queue's length on the host: the grids are fixed and every kernel reads *count.
inline void launch_queue_sort(hipStream_t stream, int num_cus, const uint32_t *keys, const int *vals, const int *count, unsigned key_bits,
                              uint32_t *keys_mid, int *vals_mid, int *scratch, int *out) {
    const int low_bits = (int)key_bits - kSortFineBits;
    (void)hipMemsetAsync(scratch, 0, (size_t)kSortScratchZeroInts * sizeof(int), stream);
    hipLaunchKernelGGL(k_sort_hist, dim3(num_cus / 2 > 0 ? num_cus / 2 : 1), dim3(kSortBlock2), 0, stream, keys, count, low_bits, scratch + kSortScratchFine);
    hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(kSortBlock2), 0, stream, scratch, low_bits);
    hipLaunchKernelGGL(k_sort_part1, dim3(num_cus * 2), dim3(kSortBlock), 0, stream, keys, vals, count, low_bits, scratch, keys_mid, vals_mid);
    hipLaunchKernelGGL(k_sort_part2, dim3(num_cus), dim3(kSortBlock2), 0, stream, keys_mid, vals_mid, scratch, out);
}

}  // namespace mipt
