// Micro-benchmark (development aid, not part of the product): per-instruction issue cost of
// dependent / independent VALU chains on gfx950 at 1, 2 and 4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/issue_rates.hip -o /tmp/issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int N = 1 << 19;

template <int KIND, int ILP>
__global__ void __launch_bounds__(1024) k(float *out, float a, float b, int n) {
    float x[ILP];
    double d[ILP];
    unsigned u[ILP];
    for (int j = 0; j < ILP; j++) { x[j] = threadIdx.x * 1e-3f + j; d[j] = x[j]; u[j] = threadIdx.x + j; }
#pragma unroll 32
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) {
            if (KIND == 0) x[j] = __builtin_fmaf(x[j], a, b);
            if (KIND == 1) x[j] = x[j] + a;
            if (KIND == 2) d[j] = __builtin_fma(d[j], (double)a, (double)b);
            if (KIND == 3) u[j] = u[j] * 0x9e3779b9u + 1u;
            if (KIND == 4) x[j] = (float)(double)x[j] + a;           // cvt f64 round trip
            if (KIND == 5) x[j] = __builtin_truncf(x[j] * a);
            if (KIND == 6) { typedef float f2 __attribute__((ext_vector_type(2))); }
            if (KIND == 7) x[j] = (x[j] < b) ? x[j] + a : x[j] * a;   // cmp + cndmask
            if (KIND == 8) x[j] = x[j] / a;                            // exact division
        }
    }
    float s = 0;
    for (int j = 0; j < ILP; j++) s += x[j] + (float)d[j] + (float)u[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float f2 __attribute__((ext_vector_type(2)));
template <int ILP>
__global__ void __launch_bounds__(1024) kpk(float *out, float a, float b, int n) {
    f2 x[ILP];
    for (int j = 0; j < ILP; j++) { x[j].x = threadIdx.x * 1e-3f + j; x[j].y = j; }
    f2 va = {a, a}, vb = {b, b};
#pragma unroll 32
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) x[j] = __builtin_elementwise_fma(x[j], va, vb);
    }
    float s = 0;
    for (int j = 0; j < ILP; j++) s += x[j].x + x[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double time_ms(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; r++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}

int main() {
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    const double clk = 2.4e9;
    const char *names[] = {"fma_f32", "add_f32", "fma_f64", "mul_lo_u32+add", "cvt_f64 roundtrip+add", "mul+trunc", "", "cmp+cndmask+2", "div_f32"};
    for (int waves : {1, 2, 4}) {
        int threads = 256 * waves;
        printf("--- %d wave(s) per SIMD (256 blocks x %d threads) ---\n", waves, threads);
#define RUN(KIND, ILP) { double ms = time_ms([&] { hipLaunchKernelGGL((k<KIND, ILP>), dim3(256), dim3(threads), 0, 0, out, 1.0001f, 0.5f, N); }); \
        printf("%-24s ilp %d: %7.2f cycles per op per wave (%.3f ms)\n", names[KIND], ILP, ms * 1e-3 * clk / (double(N) * ILP), ms); }
        RUN(0, 1) RUN(0, 2) RUN(0, 4) RUN(1, 1) RUN(2, 1) RUN(2, 2) RUN(3, 1) RUN(3, 2) RUN(4, 1) RUN(5, 1) RUN(7, 1) RUN(8, 1) RUN(8, 2)
        { double ms = time_ms([&] { hipLaunchKernelGGL((kpk<1>), dim3(256), dim3(threads), 0, 0, out, 1.0001f, 0.5f, N); });
          printf("%-24s ilp 1: %7.2f cycles per pk op per wave\n", "pk_fma_f32", ms * 1e-3 * clk / double(N)); }
        { double ms = time_ms([&] { hipLaunchKernelGGL((kpk<2>), dim3(256), dim3(threads), 0, 0, out, 1.0001f, 0.5f, N); });
          printf("%-24s ilp 2: %7.2f cycles per pk op per wave\n", "pk_fma_f32", ms * 1e-3 * clk / double(N * 2)); }
    }
    return 0;
}
