// The renderer's queue-binning stage (csrc/queue_sort.h) next to rocPRIM's radix_sort_pairs on the same (key, value) pairs:
// random and locally coherent 24-bit keys.  -DMIPT_SORT_PROF adds per-phase clocks (100 MHz wall clock) of the two passes.
// (Round 3 first measured a ONE-pass counting sort with returning global atomics here: 45 M random keys -> count 1.8 ms +
//  scatter 1.4 ms against 1.1 ms for the whole library sort; rejected.)
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../cuda_path_tracer_amd/csrc/queue_sort.h"

static uint32_t rnd(uint64_t &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); }

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 45000000;
    std::vector<uint32_t> h_keys((size_t)n);
    std::vector<int> h_vals((size_t)n);
    uint32_t *keys, *keys_out;
    int *vals, *out, *d_count, *scratch, *vals_mid;
    (void)hipMalloc(&keys, (size_t)n * 4); (void)hipMalloc(&keys_out, (size_t)n * 4); (void)hipMalloc(&vals, (size_t)n * 4);
    (void)hipMalloc(&out, (size_t)n * 4); (void)hipMalloc(&d_count, 4); (void)hipMalloc(&scratch, mipt::kSortScratchInts * 4);
    (void)hipMalloc(&vals_mid, (size_t)n * 4);
    (void)hipMemcpy(d_count, &n, 4, hipMemcpyHostToDevice);
    hipEvent_t e[2];
    for (auto &x : e) (void)hipEventCreate(&x);
    for (int mode = 0; mode < 3; mode++) {
        uint64_t s = 12345;
        if (mode == 0) for (int i = 0; i < n; i++) h_keys[(size_t)i] = rnd(s) & 0xffffffu;  // random cells
        else if (mode == 1) {  // locally coherent: runs of 64 entries inside one 2^12-cell neighbourhood, clumped over a third of the map
            for (int i = 0; i < n; i += 64) {
                const uint32_t c = (rnd(s) % 5500000u) & 0xfff000u;
                for (int k = 0; k < 64 && i + k < n; k++) h_keys[(size_t)(i + k)] = c | (rnd(s) & 0xfffu);
            }
        } else {  // a bidirectional shadow queue: runs of 512 binned entries, then runs of 1536 unbinned ones (largest key)
            for (int i = 0; i < n; i++) h_keys[(size_t)i] = (i % 2048) < 512 ? (rnd(s) & 0xffffffu) : 0xffffffu;
        }
        for (int i = 0; i < n; i++) h_vals[(size_t)i] = i;
        (void)hipMemcpy(keys, h_keys.data(), (size_t)n * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(vals, h_vals.data(), (size_t)n * 4, hipMemcpyHostToDevice);
        printf("---- %s keys, n = %d\n", mode == 0 ? "random" : (mode == 1 ? "coherent" : "mostly unbinned"), n);
        float bestq = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            (void)hipEventRecord(e[0]);
            mipt::launch_queue_sort(nullptr, 256, keys, vals, d_count, 24, keys_out, vals_mid, scratch, out);
            (void)hipEventRecord(e[1]); (void)hipEventSynchronize(e[1]);
            float ms; (void)hipEventElapsedTime(&ms, e[0], e[1]); if (ms < bestq) bestq = ms;
        }
        printf("queue_sort.h sample-sort binning: %.3f ms\n", bestq);
#ifdef MIPT_SORT_PROF
        {
            unsigned long long p[32];
            (void)hipMemcpyFromSymbol(p, HIP_SYMBOL(mipt::g_sort_prof), sizeof(p));
            printf("  part1 per tile  (x10 ns): zero+count %.0f  reserve %.0f  scatter %.0f   (%llu tiles)\n", (double)p[8] / p[11], (double)p[9] / p[11], (double)p[10] / p[11], p[11]);
            printf("  part2 per chunk (x10 ns): search+zero %.0f  count %.0f  scan %.0f  scatter %.0f   (%llu chunks)\n", (double)p[0] / p[4], (double)p[1] / p[4], (double)p[2] / p[4], (double)p[3] / p[4], p[4]);
            unsigned long long z[32] = {0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(mipt::g_sort_prof), z, sizeof(z));
        }
#endif
        size_t tb = 0;
        (void)rocprim::radix_sort_pairs(nullptr, tb, keys, keys_out, vals, out, (size_t)n, 0, 24);
        void *tmp; (void)hipMalloc(&tmp, tb);
        float bestr = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e[0]);
            (void)rocprim::radix_sort_pairs(tmp, tb, keys, keys_out, vals, out, (size_t)n, 0, 24);
            (void)hipEventRecord(e[1]); (void)hipEventSynchronize(e[1]);
            float ms; (void)hipEventElapsedTime(&ms, e[0], e[1]); if (ms < bestr) bestr = ms;
        }
        printf("rocprim radix_sort_pairs 24 bits: %.3f ms\n", bestr);
        (void)hipFree(tmp);
    }
    return 0;
}
