// Micro-benchmark (development aid): does a single wave per SIMD slow down when the loop body
// is a long straight-line block (instruction fetch) rather than a short one?
// Same ILP-8 independent fma/add chains, loop bodies of 64 .. 2048 instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int TOTAL = 1 << 20;   // instructions per chain set

template <int UNROLL, int MIX>
__global__ void __launch_bounds__(256) k(float *out, float a, float b, int iters) {
    float x[8];
    for (int j = 0; j < 8; j++) x[j] = threadIdx.x * 1e-3f + j;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (MIX == 0) x[j] = __builtin_fmaf(x[j], a, b);
                else x[j] = (u & 1) ? x[j] + a : x[j] * b;
            }
        }
    }
    float s = 0;
    for (int j = 0; j < 8; j++) s += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double time_ms(F launch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for (int r = 0; r < 5; r++) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}

int main() {
    float *out; (void)hipMalloc(&out, 256 * 1024 * 4);
#define RUN(U, MIX) { int iters = TOTAL / (U * 8); double ms = time_ms([&] { hipLaunchKernelGGL((k<U, MIX>), dim3(256), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters); }); \
        printf("%s body %4d instrs: %6.3f ns per instr per wave = %5.2f cyc @2.4GHz\n", MIX ? "add/mul" : "fma    ", U * 8, ms * 1e6 / TOTAL, ms * 1e6 / TOTAL * 2.4); }
    RUN(8, 0) RUN(32, 0) RUN(64, 0) RUN(128, 0) RUN(256, 0)
    RUN(8, 1) RUN(32, 1) RUN(64, 1) RUN(128, 1) RUN(256, 1)
    return 0;
}
