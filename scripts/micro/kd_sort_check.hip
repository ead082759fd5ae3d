// The kd build's own key sort (csrc/kd_sort.h) against std::sort (exactness) and rocPRIM's radix_sort_keys (time) on keys
// shaped like the build's: (node slot << 32 | 32 coordinate bits), sorted on 32 + owner_bits bits.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/micro/kd_sort_check scripts/micro/kd_sort_check.hip
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../cuda_path_tracer_amd/csrc/kd_sort.h"

static uint64_t rnd(uint64_t &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return s >> 11; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); return 1; } } while (0)

int main() {
    struct Case { int64_t n; int owner_bits; int mode; };
    const Case cases[] = {{1, 1, 0}, {2, 1, 0}, {63, 1, 0}, {4095, 3, 0}, {4096, 3, 0}, {4097, 3, 0}, {100003, 9, 0}, {1000003, 13, 1},
                          {3932160, 1, 1}, {6000000, 17, 1}, {6000000, 17, 2}, {12582912, 12, 1}};
    int bad = 0;
    for (const Case &c : cases) {
        uint64_t seed = 12345 + c.n;
        std::vector<unsigned long long> h((size_t)c.n);
        for (int64_t i = 0; i < c.n; i++) {
            const unsigned long long slot = rnd(seed) % (1ull << c.owner_bits);
            unsigned long long low = rnd(seed) & 0xffffffffull;
            if (c.mode == 1) low = 0xc0000000ull | (low & 0x00ffffffull);      // coordinates of one mesh: same sign and exponent
            if (c.mode == 2) low = (rnd(seed) % 3 == 0) ? 0xff800000ull : low;  // a third of the candidates are +inf
            h[(size_t)i] = (slot << 32) | low;
        }
        unsigned long long *a, *b, *r_in, *r_out;
        void *scratch, *r_temp = nullptr;
        CK(hipMalloc(&a, c.n * 8)); CK(hipMalloc(&b, c.n * 8)); CK(hipMalloc(&r_in, c.n * 8)); CK(hipMalloc(&r_out, c.n * 8));
        CK(hipMalloc(&scratch, mipt::kd_sort_scratch_bytes(c.n)));
        size_t temp_bytes = 0;
        CK(rocprim::radix_sort_keys(nullptr, temp_bytes, r_in, r_out, (size_t)c.n, 0, 32 + c.owner_bits, nullptr));
        CK(hipMalloc(&r_temp, temp_bytes + 16));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms_own = 1e9f, ms_lib = 1e9f;
        const unsigned long long *res = nullptr;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipMemcpy(a, h.data(), c.n * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(r_in, h.data(), c.n * 8, hipMemcpyHostToDevice));
            CK(hipEventRecord(e0, nullptr));
            res = mipt::launch_kd_sort(nullptr, a, b, c.n, 32 + c.owner_bits, scratch);
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_own = std::min(ms_own, ms);
            CK(hipEventRecord(e0, nullptr));
            CK(rocprim::radix_sort_keys(r_temp, temp_bytes, r_in, r_out, (size_t)c.n, 0, 32 + c.owner_bits, nullptr));
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); ms_lib = std::min(ms_lib, ms);
        }
        std::vector<unsigned long long> got((size_t)c.n);
        CK(hipMemcpy(got.data(), res, c.n * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const bool ok = got == h;
        bad += ok ? 0 : 1;
        printf("n = %9lld  key bits %2d  mode %d : %s   kd_sort.h %.3f ms   rocprim radix_sort_keys %.3f ms\n", (long long)c.n, 32 + c.owner_bits, c.mode,
               ok ? "equal to std::sort" : "MISMATCH", ms_own, ms_lib);
        (void)hipFree(a); (void)hipFree(b); (void)hipFree(r_in); (void)hipFree(r_out); (void)hipFree(scratch); (void)hipFree(r_temp);
    }
    printf(bad ? "FAILED\n" : "OK\n");
    return bad ? 1 : 0;
}
