// Test-only program: the device implementation of the deterministic transcendental functions
// (raytracing-in-one-weekend_amd/csrc/rtow_detmath.hip.h) against the oracle's (oracle/detmath.h), bit for bit, over dense sweeps.
// Built by tests/test_gpu_detmath.py with hipcc and run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "../../raytracing-in-one-weekend_amd/csrc/rtow_detmath.hip.h"
#include "../../oracle/detmath.h"

__global__ void k_sincos(const float* x, float* s, float* c, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) rtow::det_sincos(x[i], s[i], c[i]); }
__global__ void k_log(const float* x, float* y, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = rtow::det_log(x[i]); }
__global__ void k_pow(const float* x, const float* e, float* y, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = rtow::det_pow(x[i], e[i]); }
__global__ void k_divsqrt(const float* a, const float* b, float* q, float* r, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { q[i] = a[i] / b[i]; r[i] = __builtin_sqrtf(a[i]); } }

static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static float frombits(unsigned u) { float f; memcpy(&f, &u, 4); return f; }

int main()
{
    const int n = 1 << 22;
    std::vector<float> x(n), e(n), o1(n), o2(n);
    float *dx, *de, *d1, *d2;
    hipMalloc(&dx, n * 4); hipMalloc(&de, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4);
    long long bad = 0;
    // every NextFloat() value is k * 2^-23, k in [0, 2^23): sweep ALL of them for log (ProbabilisticHit) and for sincos(2*pi*u)
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < n; i++) x[i] = frombits(0x3f800000u | (unsigned)(i + pass * n)) - 1.0f;
        hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
        k_log<<<n / 256, 256>>>(dx, d1, n);
        hipMemcpy(o1.data(), d1, n * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; i++) if (bits(o1[i]) != bits(dm_logf(x[i]))) { if (bad < 5) printf("log(%a): gpu %a cpu %a\n", x[i], o1[i], dm_logf(x[i])); bad++; }
        for (int i = 0; i < n; i++) e[i] = x[i] * 2 * 3.14159265f;
        hipMemcpy(dx, e.data(), n * 4, hipMemcpyHostToDevice);
        k_sincos<<<n / 256, 256>>>(dx, d1, d2, n);
        hipMemcpy(o1.data(), d1, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(o2.data(), d2, n * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; i++) { float s, c; dm_sincosf(e[i], &s, &c); if (bits(o1[i]) != bits(s) || bits(o2[i]) != bits(c)) { if (bad < 5) printf("sincos(%a)\n", e[i]); bad++; } }
    }
    // pow: gamma exponent over [0, 4], integer exponents, and log over a wide range
    for (int i = 0; i < n; i++) { x[i] = 4.0f * (float)i / n; e[i] = (i & 7) == 0 ? 2.0f : (i & 7) == 1 ? 5.0f : (i & 7) == 2 ? (float)(i % 40) : 0.416666667f; }
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(de, e.data(), n * 4, hipMemcpyHostToDevice);
    k_pow<<<n / 256, 256>>>(dx, de, d1, n);
    hipMemcpy(o1.data(), d1, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) if (bits(o1[i]) != bits(dm_powf(x[i], e[i]))) { if (bad < 5) printf("pow(%a,%a): gpu %a cpu %a\n", x[i], e[i], o1[i], dm_powf(x[i], e[i])); bad++; }
    for (int i = 0; i < n; i++) x[i] = frombits(0x00800000u + (unsigned)i * 509u);   // positive normal floats, strided over the whole range
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k_log<<<n / 256, 256>>>(dx, d1, n);
    hipMemcpy(o1.data(), d1, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) if (bits(o1[i]) != bits(dm_logf(x[i]))) { if (bad < 5) printf("log(%a)\n", x[i]); bad++; }
    // IEEE division and square root must be correctly rounded on the device (hipcc's default), like on the host
    for (int i = 0; i < n; i++) { x[i] = frombits(0x30000000u + (unsigned)i * 97u); e[i] = frombits(0x3a000000u + (unsigned)((i * 2654435761u) >> 6)); }
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(de, e.data(), n * 4, hipMemcpyHostToDevice);
    k_divsqrt<<<n / 256, 256>>>(dx, de, d1, d2, n);
    hipMemcpy(o1.data(), d1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), d2, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) if (bits(o1[i]) != bits(x[i] / e[i]) || bits(o2[i]) != bits(sqrtf(x[i]))) { if (bad < 5) printf("div/sqrt %a %a\n", x[i], e[i]); bad++; }
    printf("detmath parity: %lld mismatches\n", bad);
    return bad ? 1 : 0;
}
