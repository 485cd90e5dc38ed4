// Micro-benchmark (development aid, not part of the product): throughput of individual VALU
// instruction classes on gfx950 with ILP inside ONE wave per SIMD (the render kernels' regime)
// and with 2/4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/issue_rates2.hip -o /tmp/issue_rates2
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int N = 1 << 17;

template <int KIND, int ILP>
__global__ void __launch_bounds__(1024) k(float *out, float a, float b, unsigned m, int n) {
    float x[ILP];
    unsigned u[ILP];
    for (int j = 0; j < ILP; j++) { x[j] = threadIdx.x * 1e-3f + j; u[j] = threadIdx.x + j * 77u; }
#pragma unroll 8
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) {
            if (KIND == 0) x[j] = __builtin_fmaf(x[j], a, b);
            if (KIND == 1) x[j] = x[j] + a;
            if (KIND == 2) x[j] = x[j] * a;
            if (KIND == 3) x[j] = x[j] - __builtin_truncf(x[j]);                 // trunc + sub
            if (KIND == 4) u[j] = u[j] ^ m;                                        // xor
            if (KIND == 5) u[j] = u[j] + m;                                        // add_u32
            if (KIND == 6) x[j] = (float)(u[j] = (unsigned)x[j] + m);            // cvt_u32_f32 + add + cvt_f32_u32
            if (KIND == 7) x[j] = (x[j] != b) ? x[j] : a;                         // cmp + cndmask
            if (KIND == 8) u[j] = u[j] * 0x9e3779b9u;                              // mul_lo_u32
            if (KIND == 9) u[j] = (unsigned short)((unsigned short)u[j] * (unsigned short)0x79b9u) + m;   // mul_lo_u16 (+add)
        }
    }
    float s = 0;
    for (int j = 0; j < ILP; j++) s += x[j] + (float)u[j];
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
    const char *names[] = {"fma_f32", "add_f32", "mul_f32", "trunc+sub (2 ops)", "xor_b32", "add_u32", "cvt,add,cvt (3 ops)", "cmp+cndmask (2 ops)", "mul_lo_u32", "mul_lo_u16+add (2+ ops)"};
    const int nops[] = {1, 1, 1, 2, 1, 1, 3, 2, 1, 2};
    for (int waves : {1, 2, 4}) {
        int threads = 256 * waves;
        printf("--- %d wave(s) per SIMD (256 blocks x %d threads); ns per INSTRUCTION per wave, and cycles at 2.4 GHz ---\n", waves, threads);
#define RUN(KIND, ILP) { double ms = time_ms([&] { hipLaunchKernelGGL((k<KIND, ILP>), dim3(256), dim3(threads), 0, 0, out, 1.0001f, 0.5f, 0x1234567u, N); }); \
        double ns = ms * 1e6 / (double(N) * ILP * nops[KIND]); \
        printf("%-26s ilp %d: %6.3f ns  %5.2f cyc  (SIMD: %5.2f cyc per wave-instr)\n", names[KIND], ILP, ns, ns * 2.4, ns * 2.4 / waves); }
        RUN(0, 1) RUN(0, 4) RUN(0, 8) RUN(1, 1) RUN(1, 4) RUN(1, 8) RUN(2, 4) RUN(2, 8) RUN(3, 4) RUN(3, 8) RUN(4, 8) RUN(5, 8) RUN(6, 4) RUN(6, 8) RUN(7, 4) RUN(7, 8) RUN(8, 4) RUN(8, 8) RUN(9, 8)
    }
    return 0;
}
