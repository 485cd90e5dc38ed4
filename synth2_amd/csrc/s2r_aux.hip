// s2r_aux.hip — the small kernels around the render kernels (coefficient tables, note events, mix, decimator) and
// the launch dispatch.  gfx950 only; -ffp-contract=off.
#include "s2r_kern_common.h"

namespace {

// ---------------------------------------------------------------------------------------
// Coefficient tables of one patch (S2rTabRef): one thread per entry.  The mod envelope's value at the
// entry — the very expression the render kernels evaluate, env_value() on the stage's line — goes through the same
// routines a voice would run per frame: sleef pow (process.rs:231-250), then filters.rs:20-24 or dsp_filters.rs.
//   entries [0, n_ad)                     attack + decay, index = frame offset t
//           [n_ad, n_ad + n_rel)          release that starts at the clamp attack + decay, index = t - rc_t0
//           [n_ad + n_rel, n_ad + 2 n_rel) later release, index = t - release_frame_offset
//           then 16 x sustain, 16 x end, 16 x "no voice" (x = 1, 1 - x = 0)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) s2r_table_kernel(const S2rTabBuild b) {
    __shared__ uint64_t sT[S2R_EXP2F_N];
    if (threadIdx.x < S2R_EXP2F_N) sT[threadIdx.x] = c_exp2f_table[threadIdx.x];
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n_entries) return;
    const S2rEnv &e = b.mod;
    float mod;
    bool dead = false;
    if (i < b.n_ad) {
        const float t = (float)i;
        const EnvRun s = env_stage_at(e, __builtin_inff(), __builtin_inff(), t);      // attack, decay (then sustain: padding)
        mod = env_value(s, t);
    } else if (i < b.n_ad + b.n_rel) {
        const float t = (float)(b.rc_t0 + (i - b.n_ad));
        mod = e.slope_rel * (t - e.sus_off) + e.S;               // simdtest.rs:312-318 with release_offset = attack + decay
    } else if (i < b.n_ad + 2u * b.n_rel) {
        const float d = (float)(i - b.n_ad - b.n_rel);           // t - release_offset, exact below 2^24
        mod = e.slope_rel * d + e.S;
    } else if (i < b.n_ad + 2u * b.n_rel + 16u) {
        mod = 0.0f * 1.0f + e.S;                                 // sustain: 0 * (t - 0) + S
    } else {
        mod = 0.0f;                                              // end: 0 * (t - 0) + 0
        dead = i >= b.n_ad + 2u * b.n_rel + 32u;
    }
    const float f_lpf = s2r_pow2_sleef_core(mod * b.amt_lpf) * b.lpf_freq;            // process.rs:148-152
    float c0, c1, c2 = 0.0f;
    if (b.lpf_kind == S2R_FILT_ONEPOLE) {
        const float num = (-2.0f * 3.14159274101257324f) * f_lpf;                     // filters.rs:20-21
        const float arg = b.fast_div_sr ? s2r_div_const_nocheck(num, b.sr, b.rcp_sr) : (num / b.sr);
        c0 = s2r_expf(arg, sT);
        c1 = 1.0f - c0;                                          // a0, filters.rs:23
        if (dead) { c0 = 1.0f; c1 = 0.0f; }
    } else {
        const FiltCoef fc = dsp_filter_coef(b.lpf_kind, b.lpf_damping, b.sr, f_lpf);
        c0 = fc.alpha; c1 = fc.beta; c2 = fc.gamma;
        if (dead) c0 = 1.0f;                                     // (what a lane without a voice reads as the one-pole's x: chunk_bank)
    }
    b.base[i] = c0;
    b.base[(size_t)b.plane + i] = c1;
    if (b.lpf_kind != S2R_FILT_ONEPOLE) b.base[2u * (size_t)b.plane + i] = c2;
    if (b.fm_plane) b.base[(size_t)b.fm_plane * b.plane + i] = dead ? 1.0f : s2r_pow2_sleef_core(mod * b.amt_osc);   // process.rs:146-147
}

// The noise table: entry x = hashnoise.rs:33-51 for the hashed word x (hash_word_x16's input `start.rotl(5) ^ word`, of
// which only the low 16 bits reach the u16 cast): v = (x * 0x9e3779b9) & 0xffff, ((v / 65535) * 2) - 1 with the
// two-operation quotient of s2r_div_u16_by_65535 in its f = v * 2^-16 form — the operations the render kernels ran per
// frame before the table existed.
__global__ void __launch_bounds__(256) s2r_noise_table_kernel(float *t) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const float f = (float)((x * 0x79b9u) & 0xffffu) * 0x1p-16f;
    const float q = __builtin_fmaf(f, 0x1.0001p-16f, f);
    t[x] = __builtin_fmaf(q, 2.0f, -1.0f);
}

// ---------------------------------------------------------------------------------------
// mix kernel: adds the workgroup partial rows in the fixed order of DESIGN.md 4.3:
//   runs of 16 consecutive workgroups sequentially -> the run sums of a mix group sequentially
//   -> the mix groups sequentially -> root (+0.0) + total.
// One workgroup handles 16 frames: thread (slot, f) adds whole runs (16 independent loads in
// flight each), the run sums meet in LDS, 16 threads finish.  Runs never straddle a mix group.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void mix_body(const S2rMixParams &m, uint32_t block, float *s_run) {
    const bool ov = m.ov_render_counter != nullptr;              // two streams: the rows' render kernel runs beside this launch
    if (ov) {
        if (threadIdx.x == 0 && !ov_wait(m.ov_render_counter, m.ov_render_target)) ov_raise(m.ov_fail, 2u);
        __syncthreads();
    }
    mix_block(m, block, s_run, ov, m.done.flag != nullptr);
    signal_done(m.done, (m.frames + 15u) / 16u);
}

__global__ void __launch_bounds__(256) s2r_mix_kernel(const S2rMixParams m) {
    extern __shared__ float s_run[];                             // [total runs][16 frames]
    tl_mark(m.timeline, m.tl_slot, 0);
    mix_body(m, blockIdx.x, s_run);
    tl_mark(m.timeline, m.tl_slot, 1);
}

// out[i] = ((+0.0 + rows[0][i]) + rows[1][i]) + ...   (rank-order combine of shard partials)
__global__ void s2r_sum_rows_kernel(const float *rows, uint32_t n_rows, uint32_t frames, uint32_t stride, int stereo, float *out, const S2rDone done) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < frames) {
        float total = 0.0f;                                      // accum = splat(0.0), synth.rs:176
        for (uint32_t r = 0; r < n_rows; ++r) total += rows[(size_t)r * stride + f];
        if (stereo) { out_store(done, out + 2u * f, total); out_store(done, out + 2u * f + 1u, total); }   // audio_player.rs:224-228
        else out_store(done, out + f, total);
    }
    signal_done(done, gridDim.x);
}

// build-defined 4x decimator: out[n] = sum over k of h[k] * x[4n + k], taps in index order,
// product and sum rounded separately
constexpr int kDecimTaps = 63;
__global__ void s2r_decimate4_kernel(const float *x, const float *h, uint32_t n_out, float *out) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_out) return;
    float acc = 0.0f;
    for (int k = 0; k < kDecimTaps; ++k) acc = acc + h[k] * x[4u * n + (uint32_t)k];
    out[n] = acc;
}
__global__ void s2r_decimate4_history_kernel(float *x, uint32_t n_out) {
    const uint32_t i = threadIdx.x;                              // one workgroup of 64: read, then write (ranges may overlap)
    float v = 0.0f;
    if (i < kDecimTaps - 1) v = x[4u * n_out + i];
    __syncthreads();
    if (i < kDecimTaps - 1) x[i] = v;
}

// publishes the first timed event of every touched voice
// ... and moves the records from mapped host memory into HBM in one coalesced sweep: the coefficient pass and
// the render kernel follow per-voice chains through them, and a PCIe round trip per hop is what they cannot afford
// (It touches no voice state — a chain's first record that sits at frame 0, the fill's folded untimed events, is applied
// by the render kernel's prologue — so the host may run it beside the previous fill's render kernel.)
// (`ov`: two streams — the copy and the heads are handed to a render kernel that may already be running: write-through stores)
__device__ __forceinline__ void heads_body(int32_t *heads, const S2rTimedEvent *tev, S2rTimedEvent *tev_copy, uint32_t n, uint32_t i, bool ov) {
    if (i >= n) return;
    const S2rTimedEvent e = tev[i];
    if (ov) {
        static_assert(sizeof(S2rTimedEvent) == 32, "two 16-byte stores");
        float *dst = reinterpret_cast<float *>(tev_copy + i);
        ov_store4(dst, (f4){s2r_u2f(e.voice), s2r_u2f(e.frame), s2r_u2f(e.flags), e.pitch});
        ov_store4(dst + 4, (f4){s2r_u2f(e.seed), s2r_u2f((uint32_t)e.next), s2r_u2f(e.program), 0.0f});
        if (e.flags & S2R_TEV_FIRST) ov_store(heads + e.voice, (int32_t)i);
    } else {
        tev_copy[i] = e;
        if (e.flags & S2R_TEV_FIRST) heads[e.voice] = (int32_t)i;
    }
}

__global__ void s2r_tev_heads_kernel(int32_t *heads, const S2rTimedEvent *tev, S2rTimedEvent *tev_copy, uint32_t n, uint32_t *ov_heads_counter) {
    heads_body(heads, tev, tev_copy, n, blockIdx.x * blockDim.x + threadIdx.x, ov_heads_counter != nullptr);
    if (ov_heads_counter != nullptr) ov_signal(ov_heads_counter);
}

// Between two render kernels of a caller with several fills in flight: the PREVIOUS fill's mix (its first `mix_blocks`
// workgroups) and THIS fill's chain heads (the rest) in one launch — the two have nothing to do with each other, and a
// launch boundary costs the stream more than either of them.
__global__ void __launch_bounds__(256) s2r_mix_and_heads_kernel(const S2rMixParams m, uint32_t mix_blocks, int32_t *heads,
                                                                const S2rTimedEvent *tev, S2rTimedEvent *tev_copy, uint32_t n) {
    extern __shared__ float s_run[];
    tl_mark(m.timeline, m.tl_slot, 0);
    if (blockIdx.x < mix_blocks) mix_body(m, blockIdx.x, s_run);
    else {
        heads_body(heads, tev, tev_copy, n, (blockIdx.x - mix_blocks) * blockDim.x + threadIdx.x, m.ov_heads_counter != nullptr);
        if (m.ov_heads_counter != nullptr) ov_signal(m.ov_heads_counter);
    }
    tl_mark(m.timeline, m.tl_slot, 1);
}

// note events folded per voice by the host (synth.rs:61-80)
__global__ void s2r_events_kernel(const S2rVoiceArrays v, const S2rVoiceEvent *ev, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const S2rVoiceEvent e = ev[i];
    const uint32_t vi = e.voice;
    if (e.flags & S2R_EV_RESTART) {                              // *voice = Voice { .. }, synth.rs:63-69
        v.pitch[vi] = e.pitch;
        v.offset[vi] = 0u;
        v.release[vi] = 0u;                                      // a release right after the on is at offset 0
        v.flags[vi] = S2R_VF_STARTED | ((e.flags & S2R_EV_RELEASE) ? S2R_VF_RELEASED : 0u);
        v.phase[vi] = 0.0f;
        v.lpf_last[vi] = 0.0f;
        v.fx1[vi] = 0.0f; v.fx2[vi] = 0.0f; v.fy1[vi] = 0.0f; v.fy2[vi] = 0.0f;
        v.seed[vi] = e.seed;
        v.program[vi] = e.flags >> S2R_EV_PROGRAM_SHIFT;
        v.osc_z[vi] = s2r_u2f(S2R_OSC_Z_NONE);
    } else if (e.flags & S2R_EV_RELEASE) {                       // synth.rs:74-75
        const uint32_t fl = v.flags[vi];
        if ((fl & S2R_VF_STARTED) && !(fl & S2R_VF_RELEASED)) {
            v.release[vi] = v.offset[vi];
            v.flags[vi] = fl | S2R_VF_RELEASED;
        }
    }
}
}  // namespace

hipError_t s2r_launch_onepole_resident_osc0(const S2rRenderArgs &a, const S2rResident &rs, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_onepole_resident_osc1(const S2rRenderArgs &a, const S2rResident &rs, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_onepole_resident_osc2(const S2rRenderArgs &a, const S2rResident &rs, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_onepole_resident_osc3(const S2rRenderArgs &a, const S2rResident &rs, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_onepole_pool_osc0(const S2rRenderArgs &a, const S2rPool &pl, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_onepole_pool_osc1(const S2rRenderArgs &a, const S2rPool &pl, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_onepole_pool_osc2(const S2rRenderArgs &a, const S2rPool &pl, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_onepole_pool_osc3(const S2rRenderArgs &a, const S2rPool &pl, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_onepole_osc0(const S2rRenderArgs &a, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_onepole_osc1(const S2rRenderArgs &a, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_onepole_osc2(const S2rRenderArgs &a, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_onepole_osc3(const S2rRenderArgs &a, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_general_pool_osc0(const S2rRenderArgs &a, const S2rPool &pl, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_general_pool_osc1(const S2rRenderArgs &a, const S2rPool &pl, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_general_pool_osc2(const S2rRenderArgs &a, const S2rPool &pl, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_general_pool_osc3(const S2rRenderArgs &a, const S2rPool &pl, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_general_pool_osc15(const S2rRenderArgs &a, const S2rPool &pl, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_general_osc0(const S2rRenderArgs &a, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_general_osc1(const S2rRenderArgs &a, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_general_osc2(const S2rRenderArgs &a, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_general_osc3(const S2rRenderArgs &a, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_general_osc15(const S2rRenderArgs &a, uint32_t block_voices, hipStream_t stream);   // the patch bank

hipError_t s2r_launch_tables(const S2rTabBuild &b, hipStream_t stream) {
    if (b.n_entries == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(s2r_table_kernel, dim3((b.n_entries + 255u) / 256u), dim3(256), 0, stream, b);
    return hipGetLastError();
}

hipError_t s2r_launch_noise_table(float *table_65536, hipStream_t stream) {
    hipLaunchKernelGGL(s2r_noise_table_kernel, dim3(65536 / 256), dim3(256), 0, stream, table_65536);
    return hipGetLastError();
}

hipError_t s2r_launch_render(const S2rRenderArgs &a, uint32_t block_voices, hipStream_t stream) {
    const S2rRenderParams &p = a.p;
    if (p.n_voices == 0 || p.frames == 0) return hipSuccess;
    if (block_voices < 64 || block_voices > 1024 || (block_voices & 63u)) return hipErrorInvalidValue;
    // patch banks, and the DPW oscillator shapes (which exist in that kernel only): oscillator and filter kind per lane
    if (p.bank_size > 1 || p.osc_kind > S2R_OSC_SINE) return s2r_launch_general_osc15(a, block_voices, stream);
    const bool general = p.lpf_kind != S2R_FILT_ONEPOLE;
    switch (p.osc_kind) {
    case S2R_OSC_SQUARE: return general ? s2r_launch_general_osc0(a, block_voices, stream) : s2r_launch_onepole_osc0(a, block_voices, stream);
    case S2R_OSC_SAW: return general ? s2r_launch_general_osc1(a, block_voices, stream) : s2r_launch_onepole_osc1(a, block_voices, stream);
    case S2R_OSC_TRIANGLE: return general ? s2r_launch_general_osc2(a, block_voices, stream) : s2r_launch_onepole_osc2(a, block_voices, stream);
    case S2R_OSC_SINE: return general ? s2r_launch_general_osc3(a, block_voices, stream) : s2r_launch_onepole_osc3(a, block_voices, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t s2r_launch_resident(const S2rRenderArgs &a, const S2rResident &rs, uint32_t block_voices, hipStream_t stream) {
    const S2rRenderParams &p = a.p;
    if (p.n_voices == 0 || p.frames == 0 || p.bank_size > 1 || p.lpf_kind != S2R_FILT_ONEPOLE) return hipErrorInvalidValue;
    switch (p.osc_kind) {
    case S2R_OSC_SQUARE: return s2r_launch_onepole_resident_osc0(a, rs, block_voices, stream);
    case S2R_OSC_SAW: return s2r_launch_onepole_resident_osc1(a, rs, block_voices, stream);
    case S2R_OSC_TRIANGLE: return s2r_launch_onepole_resident_osc2(a, rs, block_voices, stream);
    case S2R_OSC_SINE: return s2r_launch_onepole_resident_osc3(a, rs, block_voices, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t s2r_launch_pool(const S2rRenderArgs &a, const S2rPool &pl, uint32_t block_voices, hipStream_t stream) {
    const S2rRenderParams &p = a.p;
    if (p.n_voices == 0 || p.frames == 0) return hipErrorInvalidValue;
    // patch banks, and the DPW oscillator shapes: the kernel with oscillator and filter kind per lane
    if (p.bank_size > 1 || p.osc_kind > S2R_OSC_SINE) return s2r_launch_general_pool_osc15(a, pl, block_voices, stream);
    const bool general = p.lpf_kind != S2R_FILT_ONEPOLE;
    switch (p.osc_kind) {
    case S2R_OSC_SQUARE: return general ? s2r_launch_general_pool_osc0(a, pl, block_voices, stream) : s2r_launch_onepole_pool_osc0(a, pl, block_voices, stream);
    case S2R_OSC_SAW: return general ? s2r_launch_general_pool_osc1(a, pl, block_voices, stream) : s2r_launch_onepole_pool_osc1(a, pl, block_voices, stream);
    case S2R_OSC_TRIANGLE: return general ? s2r_launch_general_pool_osc2(a, pl, block_voices, stream) : s2r_launch_onepole_pool_osc2(a, pl, block_voices, stream);
    case S2R_OSC_SINE: return general ? s2r_launch_general_pool_osc3(a, pl, block_voices, stream) : s2r_launch_onepole_pool_osc3(a, pl, block_voices, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t s2r_launch_mix(const S2rMixParams &m, hipStream_t stream) {
    if (m.frames == 0) return hipSuccess;
    const uint32_t runs_per_group = (m.blocks_per_group + kMixRun - 1) / kMixRun;
    const size_t lds = (size_t)runs_per_group * m.n_groups * 16u * sizeof(float);
    if (lds > 64u * 1024u) return hipErrorInvalidValue;          // > 16 k workgroups in one shard
    hipLaunchKernelGGL(s2r_mix_kernel, dim3((m.frames + 15) / 16), dim3(256), lds, stream, m);
    return hipGetLastError();
}

hipError_t s2r_launch_mix_and_heads(const S2rMixParams &m, int32_t *heads, const S2rTimedEvent *tev, S2rTimedEvent *tev_copy, uint32_t n,
                                    hipStream_t stream) {
    if (m.frames == 0) return s2r_launch_tev_heads(heads, tev, tev_copy, n, stream, m.ov_heads_counter);
    if (n == 0) return s2r_launch_mix(m, stream);
    const uint32_t runs_per_group = (m.blocks_per_group + kMixRun - 1) / kMixRun;
    const size_t lds = (size_t)runs_per_group * m.n_groups * 16u * sizeof(float);
    if (lds > 64u * 1024u) return hipErrorInvalidValue;
    const uint32_t mix_blocks = (m.frames + 15) / 16;
    hipLaunchKernelGGL(s2r_mix_and_heads_kernel, dim3(mix_blocks + (n + 255) / 256), dim3(256), lds, stream, m, mix_blocks, heads, tev, tev_copy, n);
    return hipGetLastError();
}

hipError_t s2r_launch_events(const S2rVoiceArrays &v, const S2rVoiceEvent *dev_events, uint32_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(s2r_events_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, v, dev_events, n);
    return hipGetLastError();
}

hipError_t s2r_launch_tev_heads(int32_t *heads, const S2rTimedEvent *tev, S2rTimedEvent *tev_copy, uint32_t n, hipStream_t stream,
                                uint32_t *ov_heads_counter) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(s2r_tev_heads_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, heads, tev, tev_copy, n, ov_heads_counter);
    return hipGetLastError();
}

hipError_t s2r_launch_decimate4(float *x_with_history, const float *taps, uint32_t n_out, float *out, hipStream_t stream) {
    if (n_out == 0) return hipSuccess;
    hipLaunchKernelGGL(s2r_decimate4_kernel, dim3((n_out + 255) / 256), dim3(256), 0, stream, x_with_history, taps, n_out, out);
    hipLaunchKernelGGL(s2r_decimate4_history_kernel, dim3(1), dim3(64), 0, stream, x_with_history, n_out);
    return hipGetLastError();
}

hipError_t s2r_launch_sum_rows(const float *rows, uint32_t n_rows, uint32_t frames, uint32_t stride, int stereo, float *out, hipStream_t stream,
                               const S2rDone *done) {
    if (frames == 0) return hipSuccess;
    hipLaunchKernelGGL(s2r_sum_rows_kernel, dim3((frames + 255) / 256), dim3(256), 0, stream, rows, n_rows, frames, stride, stereo, out,
                       done ? *done : S2rDone{nullptr, 0u, nullptr});
    return hipGetLastError();
}
