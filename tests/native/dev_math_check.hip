// Device-vs-host check of the shared exact-math header (development aid and GPU test helper):
// every routine of s2r_math.h evaluated on the GPU over a dense sample of its domain and compared
// bit for bit with the SAME source compiled for the host (which libm_xcheck pins to the host libm).
// Built and run by tests/test_device_math.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "s2r_math.h"

__constant__ uint64_t c_T[S2R_EXP2F_N] = S2R_EXP2F_TABLE_INIT;
static const uint64_t h_T[S2R_EXP2F_N] = S2R_EXP2F_TABLE_INIT;

__global__ void eval(int fn, const float *in, float *out, size_t n) {
    __shared__ uint64_t sT[S2R_EXP2F_N];
    if (threadIdx.x < S2R_EXP2F_N) sT[threadIdx.x] = c_T[threadIdx.x];
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    float y = 0.0f;
    switch (fn) {
    case 0: y = s2r_sinf(x); break;
    case 1: y = s2r_cosf(x); break;
    case 2: y = s2r_tanf(x); break;
    case 3: y = s2r_expf(x, sT); break;
    case 4: y = s2r_pow2_libm(x, sT); break;
    case 5: y = s2r_pow2_sleef(x); break;
    }
    out[i] = y;
}

static float host_eval(int fn, float x) {
    switch (fn) {
    case 0: return s2r_sinf(x);
    case 1: return s2r_cosf(x);
    case 2: return s2r_tanf(x);
    case 3: return s2r_expf(x, h_T);
    case 4: return s2r_pow2_libm(x, h_T);
    default: return s2r_pow2_sleef(x);
    }
}

int main(int argc, char **argv) {
    const size_t stride = argc > 1 ? strtoul(argv[1], nullptr, 10) : 97;   // every stride-th bit pattern
    const char *names[] = {"sinf", "cosf", "tanf", "expf", "pow2_libm", "pow2_sleef"};
    std::vector<float> in;
    for (uint64_t u = 0; u < (1ull << 32); u += stride) { const float x = s2r_u2f((uint32_t)u); if (x == x) in.push_back(x); }
    const size_t n = in.size();
    float *d_in, *d_out;
    (void)hipMalloc(&d_in, n * 4); (void)hipMalloc(&d_out, n * 4);
    (void)hipMemcpy(d_in, in.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<float> out(n);
    int rc = 0;
    for (int fn = 0; fn < 6; fn++) {
        hipLaunchKernelGGL(eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, fn, d_in, d_out, n);
        (void)hipMemcpy(out.data(), d_out, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0; int shown = 0;
        for (size_t i = 0; i < n; i++) {
            const float h = host_eval(fn, in[i]);
            if (h != h && out[i] != out[i]) continue;
            if (s2r_f2u(h) != s2r_f2u(out[i])) { bad++; if (shown++ < 5) printf("  %s(%a): device %a host %a\n", names[fn], in[i], out[i], h); }
        }
        printf("%-10s %zu inputs, %zu device/host mismatches\n", names[fn], n, bad);
        if (bad) rc = 1;
    }
    return rc;
}
