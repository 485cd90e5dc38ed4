// s2r_kernels.hip — gfx950 kernels of the voice-render path.
//
// One voice per lane.  Per-voice recurrence state (phase, LPF history, frame offset) is
// loaded coalesced from the SoA arrays in HBM into registers, the frames of the fill are
// walked serially (phase accumulation and the one-pole LPF are recurrences over time), and
// the cross-voice mixdown is a wave64 DPP reduction -> LDS across the waves of a workgroup
// -> one partial row per workgroup in HBM -> a second tiny kernel that adds the rows in a
// fixed order.  No MFMA: this is a scalar-per-voice recurrence.
//
// Everything arithmetic follows the reference op for op (citations relative to
// /root/reference/components/s2_lib/src/); this file must be compiled with
// -ffp-contract=off and without fast-math.
#include <hip/hip_runtime.h>
#include "s2r_device.h"
#include "s2r_math.h"

namespace {

__constant__ uint64_t c_exp2f_table[S2R_EXP2F_N] = S2R_EXP2F_TABLE_INIT;

constexpr int kMaxWaves = 16;      // 1024-thread workgroup
constexpr int kChunk = 16;         // the reference's x16 chunk (synth.rs:158, process.rs:25)

// ---------------------------------------------------------------------------------------
// wave64 sum by DPP.  After the six steps lane 63 holds
//   (((v0+v1)+(v2+v3)) + ...)   — a balanced pairwise tree over the lanes in index order,
// which is the tree oracle/s2_oracle.c:wave_tree64 spells out.  Must run with all 64 lanes
// enabled.
// ---------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v = v + dpp_mov<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]  : pairs
    v = v + dpp_mov<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]  : quads
    v = v + dpp_mov<0x141, 0xf>(v);   // row_half_mirror      : 8
    v = v + dpp_mov<0x140, 0xf>(v);   // row_mirror           : 16 (row totals in every lane)
    v = v + dpp_mov<0x142, 0xa>(v);   // row_bcast15 -> rows 1,3 : (r0+r1), (r2+r3)
    v = v + dpp_mov<0x143, 0xc>(v);   // row_bcast31 -> rows 2,3 : total in lane 63
    return v;
}

// ---------------------------------------------------------------------------------------
// per-voice registers
// ---------------------------------------------------------------------------------------
struct VoiceRegs {
    float pitch;
    uint32_t offset;          // current_frame_offset at the start of the fill
    uint32_t release_u;
    bool released;
    float phase;              // OscillatorState.phase_accum (None == 0.0)
    float last;               // LowPassFilterState.last
    uint32_t seed_rot;        // rotl(seed, 5), hashnoise.rs:61-63
    // x16 ADSR per-voice constants (simdtest.rs:283-286)
    float ro_a, end_a, ro_m, end_m;
    // hoisted oscillator constants when mod_env_to_osc_freq == 0 (period never changes)
    float period, inv_period, slope1, slope2, half_period;
};

// old/simdtest.rs:270-331 AdsrX16::sample for one frame.  `ro`/`end` are the per-voice
// release_offset.max(sustain_offset) and + release.  Lines are mul then add, separately
// rounded (simdtest.rs:247-261).
__device__ __forceinline__ float adsr_x16(const S2rEnv &e, float ro, float end, float t) {
    const float att = e.slope_att * t + 0.0f;
    const float dec = e.slope_dec * (t - e.A) + 1.0f;
    const float rel = e.slope_rel * (t - ro) + e.S;
    float v = 0.0f;                  // in_end
    v = (t < end) ? rel : v;         // in_release
    v = (t < ro) ? e.S : v;          // in_sustain
    v = (t < e.sus_off) ? dec : v;   // in_decay
    v = (t < e.A) ? att : v;         // in_attack
    return v;
}

// math.rs:11-19 with feature fma: slope = rise / run; slope.mul_add(x, y0)
__device__ __forceinline__ float line_fma(float rise, float run, float x, float y0) {
    return __builtin_fmaf(rise / run, x, y0);
}

// envelopes.rs:21-150 Adsr::sample (scalar tail path)
__device__ float adsr_scalar(const S2rEnv &e, float t, float release_offset) {
    const float decay_offset = e.A, sustain_offset = e.sus_off;
    const float end_offset = release_offset + e.R;
    const bool in_release = t >= release_offset && t < end_offset;
    const bool in_end = t >= end_offset;
    const bool in_attack = !in_release && !in_end && t < decay_offset;
    const bool in_decay = !in_release && !in_end && !in_attack && t < sustain_offset;
    const bool in_sustain = !in_release && !in_end && !in_attack && !in_decay && t < release_offset;
    float rss;                                                         // release_start_sample, :57-93
    if (release_offset < decay_offset) rss = line_fma(1.0f, e.A, release_offset, 0.0f);
    else if (release_offset < sustain_offset) rss = line_fma(e.S - 1.0f, e.D, release_offset - decay_offset, 1.0f);
    else rss = e.S;
    if (in_attack) return line_fma(1.0f, e.A, t, 0.0f);
    if (in_decay) return line_fma(e.S - 1.0f, e.D, t - decay_offset, 1.0f);
    if (in_sustain) return e.S;
    if (in_release) return line_fma(-rss, e.R, t - release_offset, rss);
    return 0.0f;
}

// hashnoise.rs:33-51 (x16) == :14-27 (scalar): stateless noise at one frame offset
__device__ __forceinline__ float hash_noise(uint32_t seed_rot, float t) {
    const uint32_t off = s2r_f32_as_u32(t);                     // offset.cast::<u32>()
    const uint32_t h = (seed_rot ^ off) * 0x9e3779b9u;          // hash_word_x16, :57-68
    const float value = (float)(h & 0xffffu);                   // cast::<u16>() then ::<f32>()
    const float q = s2r_div_const(value, 65535.0f, 0x1.0001p-16f);   // value / u16_max
    return __builtin_fmaf(q, 2.0f, -1.0f);                      // (q * 2) is exact, then - 1
}

// oscillators.rs basic::{Square,Saw,Triangle,Table}Oscillator[X16]::sample given the
// phased offset.  `FM` selects per-frame slopes vs. the hoisted per-voice ones.
template <int OSC, bool FM>
__device__ __forceinline__ float osc_value(const VoiceRegs &r, float period, float off, const float *sSin) {
    const float x = s2r_fmod_period(off, period);               // offset % period
    if (OSC == S2R_OSC_SAW) {
        const float slope = FM ? (-2.0f / period) : r.slope1;   // :107-112
        return __builtin_fmaf(slope, x, 1.0f);
    } else if (OSC == S2R_OSC_SQUARE) {
        const float half = FM ? (period / 2.0f) : r.half_period;    // :68-73
        return x < half ? 1.0f : -1.0f;
    } else if (OSC == S2R_OSC_TRIANGLE) {
        const float half = FM ? (period / 2.0f) : r.half_period;    // :156-176
        const float s1 = FM ? (-2.0f / half) : r.slope1;
        const float s2 = FM ? (2.0f / half) : r.slope2;
        const float first = __builtin_fmaf(s1, x, 1.0f);
        const float second = __builtin_fmaf(s2, x - half, -1.0f);
        return x < half ? first : second;
    } else {
        // lookup.rs:46-85 table_lookup_exclusive_x16 on SIN_TABLE (len 1024)
        const float tv = x * 1024.0f / period;                  // :63
        const uint32_t i1 = s2r_f32_as_u32(tv);                 // :64
        const uint32_t i2 = (i1 + 1u) & 1023u;                  // :67  (% 1024, wrapping add)
        const float s1 = i1 < 1024u ? sSin[i1] : 0.0f;          // :72 gather_or_default
        const float s2 = sSin[i2];
        return __builtin_fmaf((s2 - s1) / 1.0f, tv - (float)i1, s1);   // :75-84
    }
}

// filters.rs:16-34 LowPassFilter::process
__device__ __forceinline__ float lpf_step(const S2rRenderParams &p, float f_lpf, float in, float &last, const uint64_t *sT) {
    const float num = (-2.0f * 3.14159274101257324f) * f_lpf;   // -2.0 * pi * freq
    const float arg = p.fast_div_sr ? s2r_div_const(num, p.sr, p.rcp_sr) : (num / p.sr);
    const float x = s2r_expf(arg, sT);
    const float a0 = 1.0f - x;
    const float out = __builtin_fmaf(a0, in, x * last);         // a0.mul_add(input, -b1 * last), b1 = -x
    last = out;
    return out;
}

// One frame of process_layer_x16 (process.rs:88-99,137-174,306-379) for one voice.
template <int OSC, bool FM>
__device__ __forceinline__ float frame_x16(const S2rRenderParams &p, VoiceRegs &r, uint32_t oi,
                                           const uint64_t *sT, const float *sSin) {
    const float t = (float)oi;                                   // offsets as f32 (simdtest.rs:277-279, process.rs:348)
    const float amp = adsr_x16(p.amp, r.ro_a, r.end_a, t);       // process.rs:144
    const float mod = adsr_x16(p.mod, r.ro_m, r.end_m, t);       // process.rs:145
    float period, inv_period;
    if (FM) {
        const float f_osc = s2r_pow2_sleef(mod * p.amt_osc) * r.pitch;   // process.rs:146-147,231-250
        period = p.sr / f_osc;                                   // units.rs:32-42
        inv_period = 1.0f / period;                              // oscillators.rs:378
    } else {
        period = r.period; inv_period = r.inv_period;
    }
    const float f_lpf = s2r_pow2_sleef(mod * p.amt_lpf) * p.lpf_freq;   // process.rs:148-152

    const float ph = r.phase;                                    // oscillators.rs:391-400
    r.phase = s2r_fmod1(ph + inv_period);
    const float off = __builtin_fmaf(period, ph, 0.0f);          // phased_offset_x16, :235
    const float osc = osc_value<OSC, FM>(r, period, off, sSin);

    const float osc_s = osc + p.osc_gain;                        // process.rs:342-345 (ADD)
    const float noise_s = hash_noise(r.seed_rot, t) + p.noise_level;   // process.rs:347-356 (ADD)
    const float s = osc_s + noise_s;                             // process.rs:358
    const float y = lpf_step(p, f_lpf, s, r.last, sT);           // process.rs:363-371
    return y * amp;                                              // process.rs:373-376
}

// One frame of process_layer (scalar "sisd" path: process.rs:101-135,252-304).
template <int OSC>
__device__ float frame_sisd(const S2rRenderParams &p, VoiceRegs &r, uint32_t oi,
                            const uint64_t *sT, const float *sSin) {
    const float t = (float)oi;
    const float rel = r.released ? (float)r.release_u : 4294967296.0f;   // envelopes.rs:35
    const float amp = adsr_scalar(p.amp, t, rel);
    const float mod = adsr_scalar(p.mod, t, rel);
    const float f_osc = s2r_pow2_libm(mod * p.amt_osc, sT) * r.pitch;    // process.rs:221-229
    const float f_lpf = s2r_pow2_libm(mod * p.amt_lpf, sT) * p.lpf_freq;
    const float period = p.sr / f_osc;
    const float ph = r.phase;
    const float off = __builtin_fmaf(period, ph, 0.0f);                  // oscillators.rs:212
    const float osc = osc_value<OSC, true>(r, period, off, sSin);
    r.phase = s2r_fmod1(ph + 1.0f / period);                             // oscillators.rs:377-381
    const float osc_s = osc * p.osc_gain;                                // process.rs:287 (MULTIPLY)
    const float noise_s = hash_noise(r.seed_rot, t) * p.noise_level;     // process.rs:292 (MULTIPLY)
    const float s = osc_s + noise_s;
    const float y = lpf_step(p, f_lpf, s, r.last, sT);
    return y * amp;
}

// ---------------------------------------------------------------------------------------
// render kernel: grid = ceil(n_voices / blockDim.x), blockDim.x = block_voices (64..1024).
// ---------------------------------------------------------------------------------------
template <int OSC, bool FM>
__global__ void __launch_bounds__(1024) s2r_render_kernel(const S2rRenderParams p) {
    __shared__ uint64_t sT[S2R_EXP2F_N];
    __shared__ float sW[2][kMaxWaves][kChunk];
    __shared__ float sSin[OSC == S2R_OSC_SINE ? 1024 : 1];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const uint32_t vi = blockIdx.x * blockDim.x + tid;

    if (tid < S2R_EXP2F_N) sT[tid] = c_exp2f_table[tid];
    if (OSC == S2R_OSC_SINE)
        for (uint32_t i = tid; i < 1024u; i += blockDim.x) sSin[i] = p.sin_table[i];

    // ---- load per-voice state (coalesced SoA reads) ----
    const bool in_range = vi < p.n_voices;
    const uint32_t flags = in_range ? p.v.flags[vi] : 0u;
    const bool live = (flags & S2R_VF_STARTED) != 0u;          // synth.rs:178
    VoiceRegs r;
    r.pitch = live ? p.v.pitch[vi] : 440.0f;
    r.offset = live ? p.v.offset[vi] : 0u;
    r.release_u = live ? p.v.release[vi] : 0u;
    r.released = live && (flags & S2R_VF_RELEASED) != 0u;
    r.phase = live ? p.v.phase[vi] : 0.0f;
    r.last = live ? p.v.lpf_last[vi] : 0.0f;
    const uint32_t seed = live ? p.v.seed[vi] : 0u;
    r.seed_rot = (seed << 5) | (seed >> 27);

    const float rel_f = r.released ? (float)r.release_u : 4294967296.0f;     // u32::MAX as f32
    r.ro_a = __builtin_fmaxf(rel_f, p.amp.sus_off); r.end_a = r.ro_a + p.amp.R;
    r.ro_m = __builtin_fmaxf(rel_f, p.mod.sus_off); r.end_m = r.ro_m + p.mod.R;

    // mod_env_to_osc_freq == 0: pow(2, mod*0) == 1 exactly, so freq == pitch and the period
    // (and everything derived by one correctly rounded division) is constant per voice.
    r.period = p.sr / (1.0f * r.pitch);
    r.inv_period = 1.0f / r.period;
    r.half_period = r.period / 2.0f;
    if (OSC == S2R_OSC_TRIANGLE) { r.slope1 = -2.0f / r.half_period; r.slope2 = 2.0f / r.half_period; }
    else { r.slope1 = -2.0f / r.period; r.slope2 = 0.0f; }

    __syncthreads();

    const bool wave_live = __ballot(live) != 0ull;
    const uint32_t n_chunks = p.frames / kChunk, tail = p.frames % kChunk;
    const size_t pv_base = (size_t)vi * p.frames;
    float *bp = p.block_partials + (size_t)blockIdx.x * p.frames_stride;
    uint32_t buf = 0;

    for (uint32_t c = 0; c <= n_chunks; ++c) {
        const uint32_t n_here = c < n_chunks ? kChunk : tail;
        if (n_here == 0) break;
        const uint32_t f0 = c * kChunk;
        if (wave_live) {
            if (c < n_chunks) {
#pragma unroll 2
                for (uint32_t i = 0; i < kChunk; ++i) {
                    float out = frame_x16<OSC, FM>(p, r, r.offset + f0 + i, sT, sSin);
                    out = live ? out : 0.0f;
                    if (p.per_voice && in_range) p.per_voice[pv_base + f0 + i] = out;
                    const float tot = wave_sum_lane63(out);
                    if (lane == 63u) sW[buf][wave][i] = tot;
                }
            } else {
                for (uint32_t i = 0; i < tail; ++i) {
                    float out = frame_sisd<OSC>(p, r, r.offset + f0 + i, sT, sSin);
                    out = live ? out : 0.0f;
                    if (p.per_voice && in_range) p.per_voice[pv_base + f0 + i] = out;
                    const float tot = wave_sum_lane63(out);
                    if (lane == 63u) sW[buf][wave][i] = tot;
                }
            }
        } else {
            if (lane < n_here) sW[buf][wave][lane] = 0.0f;
            if (p.per_voice && in_range)
                for (uint32_t i = 0; i < n_here; ++i) p.per_voice[pv_base + f0 + i] = 0.0f;
        }
        __syncthreads();
        if (tid < n_here) {                       // waves of the block, in wave order
            float acc = sW[buf][0][tid];
            for (uint32_t w = 1; w < n_waves; ++w) acc += sW[buf][w][tid];
            bp[f0 + tid] = acc;
        }
        buf ^= 1u;
    }

    // ---- write back the recurrence state ----
    if (live) {
        const uint32_t o = r.offset;
        p.v.offset[vi] = (o > 0xffffffffu - p.frames) ? 0xffffffffu : o + p.frames;   // synth.rs:197
        p.v.phase[vi] = r.phase;
        p.v.lpf_last[vi] = r.last;
    }
}

// ---------------------------------------------------------------------------------------
// mix kernel: one thread per frame adds the workgroup partial rows in a fixed order:
// blocks sequentially inside each group, groups sequentially, root (+0.0) + total.
// ---------------------------------------------------------------------------------------
__global__ void s2r_mix_kernel(const S2rMixParams m) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= m.frames) return;
    float total = 0.0f;                                          // accum = splat(0.0), synth.rs:176
    for (uint32_t g = 0; g < m.n_groups; ++g) {
        const uint32_t b0 = g * m.blocks_per_group;
        uint32_t b1 = b0 + m.blocks_per_group;
        if (b1 > m.n_blocks) b1 = m.n_blocks;
        if (b0 >= b1) continue;
        float acc = m.block_partials[(size_t)b0 * m.frames_stride + f];
        for (uint32_t b = b0 + 1; b < b1; ++b) acc += m.block_partials[(size_t)b * m.frames_stride + f];
        total = (m.root_add || g > 0) ? total + acc : acc;
    }
    if (m.stereo) { m.out[2 * f] = total; m.out[2 * f + 1] = total; }
    else m.out[f] = total;
}

// out[i] = ((+0.0 + rows[0][i]) + rows[1][i]) + ...   (rank-order combine of shard partials)
__global__ void s2r_sum_rows_kernel(const float *rows, uint32_t n_rows, uint32_t frames, float *out) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames) return;
    float total = 0.0f;
    for (uint32_t r = 0; r < n_rows; ++r) total += rows[(size_t)r * frames + f];
    out[f] = total;
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
        v.seed[vi] = e.seed;
    } else if (e.flags & S2R_EV_RELEASE) {                       // synth.rs:74-75
        const uint32_t fl = v.flags[vi];
        if ((fl & S2R_VF_STARTED) && !(fl & S2R_VF_RELEASED)) {
            v.release[vi] = v.offset[vi];
            v.flags[vi] = fl | S2R_VF_RELEASED;
        }
    }
}

template <int OSC>
hipError_t launch_osc(const S2rRenderParams &p, uint32_t block_voices, hipStream_t stream) {
    const uint32_t grid = (p.n_voices + block_voices - 1) / block_voices;
    // pow(2, mod * amount) == 1 exactly iff amount is +-0 (mod is always finite and >= 0)
    if (p.amt_osc == 0.0f) hipLaunchKernelGGL((s2r_render_kernel<OSC, false>), dim3(grid), dim3(block_voices), 0, stream, p);
    else hipLaunchKernelGGL((s2r_render_kernel<OSC, true>), dim3(grid), dim3(block_voices), 0, stream, p);
    return hipGetLastError();
}

}  // namespace

hipError_t s2r_launch_render(const S2rRenderParams &p, uint32_t block_voices, hipStream_t stream) {
    if (p.n_voices == 0 || p.frames == 0) return hipSuccess;
    if (block_voices < 64 || block_voices > 1024 || (block_voices & 63u)) return hipErrorInvalidValue;
    switch (p.osc_kind) {
    case S2R_OSC_SQUARE: return launch_osc<S2R_OSC_SQUARE>(p, block_voices, stream);
    case S2R_OSC_SAW: return launch_osc<S2R_OSC_SAW>(p, block_voices, stream);
    case S2R_OSC_TRIANGLE: return launch_osc<S2R_OSC_TRIANGLE>(p, block_voices, stream);
    case S2R_OSC_SINE: return launch_osc<S2R_OSC_SINE>(p, block_voices, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t s2r_launch_mix(const S2rMixParams &m, hipStream_t stream) {
    if (m.frames == 0) return hipSuccess;
    hipLaunchKernelGGL(s2r_mix_kernel, dim3((m.frames + 255) / 256), dim3(256), 0, stream, m);
    return hipGetLastError();
}

hipError_t s2r_launch_events(const S2rVoiceArrays &v, const S2rVoiceEvent *dev_events, uint32_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(s2r_events_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, v, dev_events, n);
    return hipGetLastError();
}

hipError_t s2r_launch_sum_rows(const float *rows, uint32_t n_rows, uint32_t frames, float *out, hipStream_t stream) {
    if (frames == 0) return hipSuccess;
    hipLaunchKernelGGL(s2r_sum_rows_kernel, dim3((frames + 255) / 256), dim3(256), 0, stream, rows, n_rows, frames, out);
    return hipGetLastError();
}
