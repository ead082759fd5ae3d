// throughput of 32x32 -> 64-bit multiplies on gfx950: v_mul_hi_u32 + v_mul_lo_u32 against one v_mad_u64_u32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ void mul_split(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
    hi = __umulhi(a, b); lo = a * b;
}
__device__ __forceinline__ void mul_mad(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
    uint64_t r; uint64_t carry;
    asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r), "=s"(carry) : "v"(a), "v"(b));
    hi = (uint32_t)(r >> 32); lo = (uint32_t)r;
}
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, int rounds) {
    uint32_t c0 = threadIdx.x, c1 = blockIdx.x, c2 = 17, c3 = 99, k0 = 1, k1 = 2;
    for (int r = 0; r < rounds; r++) {
        uint32_t h0, l0, h1, l1;
        if (MODE == 0) { mul_split(0xD2511F53u, c0, h0, l0); mul_split(0xCD9E8D57u, c2, h1, l1); }
        else { mul_mad(0xD2511F53u, c0, h0, l0); mul_mad(0xCD9E8D57u, c2, h1, l1); }
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3;
}
int main() {
    uint32_t *d; hipMalloc(&d, 256 * 4096 * 4);
    uint32_t h[2][4];
    for (int mode = 0; mode < 2; mode++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(4096), dim3(256), 0, 0, d, 4000);
            else hipLaunchKernelGGL(k<1>, dim3(4096), dim3(256), 0, 0, d, 4000);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("mode %d: %.3f ms  (%.1f G rounds/s)\n", mode, ms, 4096.0 * 256 * 4000 / ms / 1e6);
        }
        hipMemcpy(h[mode], d, 16, hipMemcpyDeviceToHost);
    }
    printf("results %s\n", (h[0][0] == h[1][0] && h[0][3] == h[1][3]) ? "equal" : "DIFFER");
    return 0;
}
