// Same-address device-scope atomicAdd throughput: one atomic per wave (what the queue appends do).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_one(int *ctr, int *out, int per_wave) {
    const int lane = threadIdx.x & 63;
    int v = 0;
    for (int i = 0; i < per_wave; i++) {
        int b = 0;
        if (lane == 0) b = atomicAdd(ctr, 64);
        v += __shfl(b, 0, 64);
    }
    if (v == -1) out[0] = v;
}
__global__ void k_multi(int *ctr, int *out, int per_wave, int n_addr) {
    const int lane = threadIdx.x & 63;
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int v = 0;
    for (int i = 0; i < per_wave; i++) {
        int b = 0;
        if (lane == 0) b = atomicAdd(ctr + 64 * ((w + i) % n_addr), 64);
        v += __shfl(b, 0, 64);
    }
    if (v == -1) out[0] = v;
}
int main() {
    int *ctr, *out;
    hipMalloc(&ctr, 1 << 20); hipMalloc(&out, 4);
    hipMemset(ctr, 0, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int n_addr : {1, 4, 16, 256}) {
        for (int blocks : {1024, 65536}) {
            const int per_wave = 4;
            k_multi<<<blocks, 256>>>(ctr, out, per_wave, n_addr);
            hipEventRecord(e0);
            k_multi<<<blocks, 256>>>(ctr, out, per_wave, n_addr);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)blocks * 4 * per_wave;
            printf("addrs %4d blocks %6d: %.3f ms, %.1f ns per atomic, %.1f M atomics/s\n", n_addr, blocks, ms, ms * 1e6 / n, n / ms / 1e3);
        }
    }
    return 0;
}
