// Micro-benchmark (development aid): the branch-free 16-frame chunk of the render kernel on its
// own — one wave per SIMD, nothing else in the loop — against the same chunk inside the kernel.
#include "../../synth2_amd/csrc/s2r_kernels.hip"
#include <cstdio>

namespace {
template <int WITH_REDUCE>
__global__ void __launch_bounds__(256) chunk_only(S2rRenderParams p, float *out, int n_chunks) {
    __shared__ uint64_t sT[S2R_EXP2F_N];
    __shared__ __attribute__((aligned(8))) float sSin[2];
    __shared__ float tile[4][kChunk * 65];
    __shared__ float sW[4][4][256];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x < S2R_EXP2F_N) sT[threadIdx.x] = c_exp2f_table[threadIdx.x];
    __syncthreads();
    VoiceRegs r{};
    r.pitch = 110.0f + threadIdx.x; r.phase = 0.1f; r.last = 0.0f; r.seed_rot = threadIdx.x * 2654435761u;
    EnvRun ea; ea.slope = 0.0f; ea.base = 0.0f; ea.y0 = 0.5f; ea.thr = __builtin_inff();
    EnvRun em = ea; em.y0 = 0.0f;
    FlatCache fc; fc.xc = 0.97f; fc.k = make_osck<S2R_OSC_SAW>(p.sr / r.pitch);
    const OscK k = fc.k;
    uint32_t o = 100000u;
    for (int c = 0; c < n_chunks; ++c) {
        chunk_fast<S2R_OSC_SAW, 0>(p, r, ea, em, fc, k, o, nullptr, sT, sSin, true, &tile[wave][lane], 65, nullptr);
        if (WITH_REDUCE) {
            const uint32_t f = lane & 15u, grp = lane >> 4;
            const float *src = &tile[wave][f * 65 + grp * 16u];
            float acc = src[0];
#pragma unroll
            for (int q = 1; q < 16; ++q) acc += src[q];
            sW[wave][grp][((c & 15) << 4) + f] = acc;
        }
        o += 16u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r.last + r.phase + tile[wave][lane] + sW[wave][lane & 3][lane];
}
}  // namespace

int main() {
    float *out; (void)hipMalloc(&out, 256 * 256 * 4);
    S2rRenderParams p{};
    p.osc_gain = 1.0f; p.noise_level = 0.0f; p.sr = 48000.0f; p.rcp_sr = 1.0f / 48000.0f; p.fast_div_sr = 1;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int n_chunks = 64 * 16;      // 16 buffers of 1024 frames
    for (int with = 0; with < 2; ++with) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (with) hipLaunchKernelGGL(chunk_only<1>, dim3(256), dim3(256), 0, 0, p, out, n_chunks);
            else hipLaunchKernelGGL(chunk_only<0>, dim3(256), dim3(256), 0, 0, p, out, n_chunks);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("chunk only%s: %.3f ms for %d chunks = %.1f ns per chunk = %.1f us per 1024 frames (%.0f cycles @2.4GHz per chunk)\n",
                            with ? " + transpose-add" : "", ms, n_chunks, ms * 1e6 / n_chunks, ms * 1e3 / n_chunks * 64, ms * 1e6 / n_chunks * 2.4);
        }
    }
    return 0;
}
