// gather_calib.hip -- what do FETCH_SIZE and TCC_MISS report for the access shapes of the closest-hit kernel?
// MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced streams (it reads 1/2 of the bytes); k_extend reads
// 8..64-byte records at data-dependent addresses.  Every mode below touches a KNOWN number of distinct 128-byte lines of
// a 2 GiB buffer (8x the Infinity Cache, nothing is re-used), so bytes-per-miss can be read off the counters:
//   mode 0  stream      lane i reads 16 B at 16*i                         (N*16 bytes, N/8 lines, every line fully used)
//   mode 1  gather16    lane i reads 16 B at the start of line p(i)        (N lines, one 64-byte half of each touched)
//   mode 2  gather2x16  lane i reads 16 B at offsets 0 and 64 of line p(i)  (N lines, both halves touched by one lane)
//   mode 3  gather64    lane i reads 64 B (4 x 16) at the start of line p(i) (N lines, one half fully used)
//   mode 4  gather32@96 lane i reads 32 B at offset 48 of line p(i): a 32-byte record straddling the two halves
// p() is a bijection on the 2^24 lines (odd multiplier mod 2^24).  Run each mode under
//   rocprofv3 --pmc FETCH_SIZE TCC_MISS_sum   and   --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
// usage: gather_calib <mode> [log2 lanes, default 24]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr unsigned kLines = 1u << 24;  // 2 GiB / 128 B

__device__ __forceinline__ unsigned perm(unsigned i) { return (i * 2654435761u) & (kLines - 1u); }

__global__ __launch_bounds__(256) void k_calib(const float4 *__restrict__ buf, unsigned n, int mode, float *__restrict__ sink) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    if (mode == 0) {
        const float4 v = buf[i];
        acc = v.x + v.w;
    } else {
        const float4 *line = buf + (size_t)perm(i) * 8u;  // 8 float4 per 128-byte line
        if (mode == 1) { const float4 v = line[0]; acc = v.x + v.w; }
        if (mode == 2) { const float4 v = line[0], w = line[4]; acc = v.x + w.w; }
        if (mode == 3) { const float4 a = line[0], b = line[1], c = line[2], d = line[3]; acc = a.x + b.y + c.z + d.w; }
        if (mode == 4) { const float4 a = line[3], b = line[4]; acc = a.x + b.w; }
    }
    if (acc == 123456.789f) sink[0] = acc;  // keeps the loads alive, never true for a zeroed buffer
}

int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 1;
    const unsigned n = 1u << (argc > 2 ? atoi(argv[2]) : 24);
    if (n > kLines) { fprintf(stderr, "at most 2^24 lanes\n"); return 1; }
    float4 *buf;
    float *sink;
    CHECK(hipMalloc(&buf, (size_t)kLines * 128));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(buf, 0, (size_t)kLines * 128));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipMemset(buf, 0, (size_t)kLines * 128));  // 2 GiB of stores: nothing of the last pass is left in L2 / MALL
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k_calib, dim3((n + 255) / 256), dim3(256), 0, 0, buf, n, mode, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double lines = mode == 0 ? n / 8.0 : (double)n;
    printf("mode %d lanes %u distinct_lines %.0f best_ms %.4f lines_per_us %.1f GB/s_at_128B_per_line %.1f GB/s_at_64B_per_line %.1f\n", mode, n, lines,
           best, lines / best / 1e3, lines * 128 / best / 1e6, lines * 64 / best / 1e6);
    return 0;
}
