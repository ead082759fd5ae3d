// pt_math.h -- fp32 vector helpers shared by the host packer and the HIP kernels of libmipt.
// Semantics follow math/math.cuh:18-47 of the reference (notably normalize() = v * (1/(|v|+eps)),
// math.cuh:28).  Everything is evaluated left to right in fp32; the library is compiled with
// -ffp-contract=off so the compiler never fuses a*b+c.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD inline
#endif

namespace mipt {

constexpr float kEps = 1e-5f;                     // constants.cuh:7
constexpr float kPi = 3.14159265358979323846f;    // constants.cuh:10

struct f3 {
    float x, y, z;
};
PT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
PT_HD f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
PT_HD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_HD f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
PT_HD float len3(f3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
PT_HD f3 norm3(f3 a) { return a * (1 / (len3(a) + kEps)); }
PT_HD float avg3(f3 a) { return (a.x + a.y + a.z) / 3.0f; }

}  // namespace mipt
