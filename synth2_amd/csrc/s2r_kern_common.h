// s2r_kern_common.h — device code shared by the gfx950 render kernels (one translation unit per oscillator kind, so
// that they compile in parallel: synth2_amd/build.py).  Included inside each .hip file; everything here is
// file-local.
//
// One voice per lane.  Per-voice recurrence state (phase, LPF history, frame offset) is loaded coalesced from the SoA
// arrays in HBM into registers, the frames of the fill are walked serially (phase accumulation and the one-pole LPF
// are recurrences over time), and the cross-voice mixdown is an LDS transpose-and-add per wave (16 voices in index
// order, the reference's own order) -> LDS across the waves of a workgroup -> one partial row per workgroup in HBM ->
// the rows added in a fixed order.  No MFMA: this is a scalar-per-voice recurrence.
//
// Everything arithmetic follows the reference op for op (citations relative to
// /root/reference/components/s2_lib/src/); compile with -ffp-contract=off and without fast-math.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "s2r_device.h"
#include "s2r_math.h"

namespace {

__constant__ uint64_t c_exp2f_table[S2R_EXP2F_N] = S2R_EXP2F_TABLE_INIT;

constexpr int kChunk = 16;         // the reference's x16 chunk (synth.rs:158, process.rs:25)
constexpr uint32_t kSuperMax = 256; // frames between two cross-wave combines: 256 (small workgroups) or 64
constexpr uint32_t kSuperWhole = 1024; // ... or the whole fill, where a workgroup has its compute unit to itself (below)
constexpr int kP = 4;              // frames whose closed-form work one lane carries at once (ILP)
// Row length (floats) of a wave's [16 frames][64 voices] transpose tile.  68 = 64 + 4: a row starts on a 16-byte boundary,
// so the lane that adds up frame f of voice group g reads its 16 values (tile[f * 68 + 16 g ...]) with four 16-byte
// LDS reads instead of eight 8-byte pairs, and the rows' bank offset (68 mod 64 = 4 banks per row) keeps those reads
// conflict-free: within one ds_read_b128 lane group the 16 frames land on banks 0, 4, ..., 60.
constexpr uint32_t kTileRow = 68;

// The closed-form part of kP = 4 consecutive frames is evaluated together on 4-wide vectors.
// Measured on MI355X (tools/ubench/issue_rates.hip): a SIMD retires one DEPENDENT VALU op per
// ~4.4 cycles however many waves it holds, but ~2.2-2.8 cycles per op once each wave offers two
// to four independent instructions — so the parallelism has to come from inside the wave.
// Element-wise vector code is exactly that (and the add/mul/fma halves become v_pk_*_f32).
// Each lane of every vector op is the same IEEE operation as the scalar code in s2r_math.h.
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned long long ul4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 vfma(f4 a, f4 b, f4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ d4 vfma(d4 a, d4 b, d4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f4 splat(float x) { return (f4)(x); }

// s2r_pow2_sleef_core (s2r_math.h) on four lanes
__device__ __forceinline__ f4 pow2_sleef_core4(f4 y) {
    const f4 Lh = splat(0.69314718246459960938f), Ll = splat(-1.904654323148236017e-09f);
    const f4 dx = Lh * y;
    const f4 dy = vfma(Ll, y, vfma(Lh, y, -dx));
    const f4 R_LN2f = splat(1.442695040888963407359924681001892137426645954152985934135449406931f);
    const f4 L2Uf = splat(0.693145751953125f), L2Lf = splat(1.428606765330187045e-06f);
    f4 u = (dx + dy) * R_LN2f;
    const f4 qf = __builtin_elementwise_rint(u);
    const i4 q = __builtin_convertvector(qf, i4);
    f4 a = qf * -L2Uf;
    f4 sx = dx + a, v = sx - dx;
    f4 sy = ((dx - (sx - v)) + (a - v)) + dy;
    a = qf * -L2Lf;
    f4 tx = sx + a; v = tx - sx;
    f4 ty = ((sx - (tx - v)) + (a - v)) + sy;
    sx = tx + ty; sy = (tx - sx) + ty;
    u = splat(0.00136324646882712841033936f);
    u = vfma(u, sx, splat(0.00836596917361021041870117f));
    u = vfma(u, sx, splat(0.0416710823774337768554688f));
    u = vfma(u, sx, splat(0.166665524244308471679688f));
    u = vfma(u, sx, splat(0.499999850988388061523438f));
    const f4 wx = sx * sx;
    const f4 wy = vfma(sx + sx, sy, vfma(sx, sx, -wx));
    const f4 mx = wx * u;
    const f4 my = vfma(wy, u, vfma(wx, u, -mx));
    tx = sx + mx; v = tx - sx;
    ty = ((sx - (tx - v)) + (mx - v)) + (sy + my);
    const f4 ox = splat(1.0f) + tx;
    const f4 oy = ((splat(1.0f) - ox) + tx) + ty;
    u = ox + oy;
    const i4 q1 = q >> 1, q2 = q - q1;
    u = u * (f4)((u4)(q1 + 127) << 23) * (f4)((u4)(q2 + 127) << 23);
    u = (dx < splat(-104.0f)) ? splat(0.0f) : u;
    return (y == splat(0.0f)) ? splat(1.0f) : u;
}

// s2r_div_const_nocheck on four lanes
__device__ __forceinline__ f4 div_const_nocheck4(f4 x, float c, float rc) {
    const f4 q0 = x * splat(rc);
    const f4 e = vfma(-q0, splat(c), x);
    return vfma(e, splat(rc), q0);
}

// s2r_expf on four lanes: the four LDS table reads are independent and issued together
__device__ __forceinline__ f4 expf4(f4 x, const uint64_t *T) {
    const d4 xd = __builtin_convertvector(x, d4);
    const d4 InvLn2N = (d4)(0x1.71547652b82fep+0 * S2R_EXP2F_N);
    const d4 Shift = (d4)(0x1.8p+52);
    const d4 C0 = (d4)(0x1.c6af84b912394p-5 / S2R_EXP2F_N / S2R_EXP2F_N / S2R_EXP2F_N);
    const d4 C1 = (d4)(0x1.ebfce50fac4f3p-3 / S2R_EXP2F_N / S2R_EXP2F_N);
    const d4 C2 = (d4)(0x1.62e42ff0c52d6p-1 / S2R_EXP2F_N);
    const d4 z0 = InvLn2N * xd;
    d4 kd = z0 + Shift;
    const ul4 ki = (ul4)kd;
    kd = kd - Shift;
    const d4 r = vfma(InvLn2N, xd, -kd);
    ul4 t;
    t.x = T[ki.x % S2R_EXP2F_N]; t.y = T[ki.y % S2R_EXP2F_N];
    t.z = T[ki.z % S2R_EXP2F_N]; t.w = T[ki.w % S2R_EXP2F_N];
    t += ki << (52 - 5);
    const d4 s = (d4)t;
    const d4 z = vfma(C0, r, C1);
    const d4 r2 = r * r;
    d4 y = vfma(C2, r, (d4)(1.0));
    y = vfma(z, r2, y);
    y = y * s;
    f4 res = __builtin_convertvector(y, f4);
    // |x| >= 88 or NaN in any lane: redo those lanes with the scalar routine (rare)
    const u4 abstop = (((u4)x) >> 20) & 0x7ffu;
    const i4 special = abstop >= 0x42bu;
    if (__builtin_expect((special.x | special.y | special.z | special.w) != 0, 0)) {
        if (special.x) res.x = s2r_expf(x.x, T);
        if (special.y) res.y = s2r_expf(x.y, T);
        if (special.z) res.z = s2r_expf(x.z, T);
        if (special.w) res.w = s2r_expf(x.w, T);
    }
    return res;
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
};

// The x16 ADSR (old/simdtest.rs:270-331) is a cascade of four `t < threshold` tests selecting one
// of five expressions.  A stage, once entered, lasts until t reaches its end threshold: keep the
// ACTIVE stage's line  slope * (t - base) + y0  and that threshold in registers and re-run the
// cascade only when t reaches it.  (t only grows within a fill; thresholds never precede the
// stage they end, so "t < thr of the stage found at an earlier t" implies the same stage now.)
// The value produced is the reference's selected expression, operation for operation.
// The record is ONE vector value — elements: .s0 slope, .s1 base, .s2 y0, .s3 thr (the stage's end threshold), .s4 stage
// (0 attack, 1 decay, 2 sustain, 3 release, 4 end; a float like its neighbours) — not a struct: as a struct of five floats
// the kernels' two records lived in scratch memory (the optimiser merges neighbouring field accesses of the still
// addressable struct into overlapping vector loads and stores and can then no longer take the struct apart: DESIGN.md 2),
// and every stage change paid trips to it.
typedef float EnvRun __attribute__((ext_vector_type(5)));

// the cascade of simdtest.rs:288-292 for one frame offset t, from scratch: the first stage whose
// `t < threshold` test holds (0 attack, 1 decay, 2 sustain, 3 release, 4 end), returned as that
// stage's line and end threshold.  Straight-line selects only, so everything stays in registers.
__device__ __forceinline__ EnvRun env_stage_at(const S2rEnv &e, float ro, float end, float t) {
    const bool s0 = t < e.A;
    const bool s1 = !s0 && t < e.sus_off;
    const bool s2 = !s0 && !s1 && t < ro;
    const bool s3 = !s0 && !s1 && !s2 && t < end;
    EnvRun s;
    //  0: (1/A) * t + 0        1: ((S-1)/D) * (t-A) + 1     2: S  (0*t + S == S)
    //  3: (-S/R) * (t-ro) + S  4: 0
    s.s0 = s0 ? e.slope_att : s1 ? e.slope_dec : s3 ? e.slope_rel : 0.0f;
    s.s1  = s1 ? e.A : s3 ? ro : 0.0f;
    s.s2    = s1 ? 1.0f : (s2 || s3) ? e.S : 0.0f;
    s.s3   = s0 ? e.A : s1 ? e.sus_off : s2 ? ro : s3 ? end : __builtin_inff();
    s.s4 = s0 ? 0.0f : s1 ? 1.0f : s2 ? 2.0f : s3 ? 3.0f : 4.0f;
    return s;
}

__device__ __forceinline__ float env_value(const EnvRun &s, float t) {
    return s.s0 * (t - s.s1) + s.s2;       // mul then add, separately rounded (simdtest.rs:247-261)
}

// math.rs:11-19 with feature fma: slope = rise / run; slope.mul_add(x, y0)
__device__ __forceinline__ float line_fma(float rise, float run, float x, float y0) {
    return __builtin_fmaf(rise / run, x, y0);
}

// envelopes.rs:21-150 Adsr::sample (scalar tail path)
__device__ __forceinline__ float adsr_scalar(const S2rEnv &e, float t, float release_offset) {
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
    // hash_word_x16, :57-68, then cast::<u16>(): the low 16 bits of (seed_rot ^ off) * 0x9e3779b9
    // depend only on the low 16 bits of both factors
    const uint16_t h = (uint16_t)((uint16_t)(seed_rot ^ off) * (uint16_t)0x79b9u);
    const float value = (float)h;                               // cast::<f32>()
    const float q = s2r_div_u16_by_65535(value);                // value / u16_max, correctly rounded
    return __builtin_fmaf(q, 2.0f, -1.0f);                      // (q * 2) is exact, then - 1
}

// Per-frame oscillator constants.  With mod_env_to_osc_freq == 0 they never change and live
// in VoiceRegs; with FM they are part of the frame's closed-form work.
struct OscK {
    float period, inv_period;
    float a, b, c;      // SAW: a = -2/period | SQUARE: a = period/2 | TRIANGLE: a = period/2, b = -2/a, c = 2/a
};

template <int OSC>
__device__ __forceinline__ OscK make_osck(float period) {
    OscK k;
    k.period = period;
    // One division: RN(c / period) for c = -2, -4, 4 is c/1 times RN(1 / period) exactly (a power of two commutes
    // with rounding), and period / 2 is exact, so -2 / (period / 2) == RN(-4 / period).
    k.inv_period = 1.0f / period;                                // oscillators.rs:378
    k.a = k.b = k.c = 0.0f;
    if (OSC == S2R_OSC_SAW) k.a = -2.0f * k.inv_period;          // -2.0 / period                  oscillators.rs:107-112
    if (OSC == S2R_OSC_SQUARE) k.a = period * 0.5f;              // period / 2.0                   :68-69
    if (OSC == S2R_OSC_TRIANGLE) { k.a = period * 0.5f; k.b = -4.0f * k.inv_period; k.c = 4.0f * k.inv_period; }   // -2.0 / half, 2.0 / half   :156-172
    return k;
}

// SIN_TABLE in LDS as pairs: entry i2 holds (SIN_TABLE[(i2 - 1) mod 1024], SIN_TABLE[i2]), so the two
// neighbours a lookup interpolates between (lookup.rs:64-72: i1 and i2 = (i1 + 1) mod 1024) come with one read
__device__ __forceinline__ float2 sin_pair(const float *sSin, uint32_t i2) {
    return *reinterpret_cast<const float2 *>(sSin + 2u * i2);
}

// oscillators.rs basic::{Square,Saw,Triangle,Table}Oscillator[X16]::sample given the phased offset
template <int OSC>
__device__ __forceinline__ float osc_value(const OscK &k, float off, const float *sSin) {
    // offset % period: `off` itself while 0 <= off < period (one unsigned compare on the bit
    // patterns, see s2r_fmod_period); the exact library fmodf only if some lane of the wave needs it
    float x = off;
    const bool slow = !(s2r_f2u(off) < s2r_f2u(k.period) && k.period > 0.0f);
    if (__builtin_expect(__ballot(slow) != 0ull, 0)) { if (slow) x = ::fmodf(off, k.period); }
    if (OSC == S2R_OSC_SAW) {
        return __builtin_fmaf(k.a, x, 1.0f);
    } else if (OSC == S2R_OSC_SQUARE) {
        return x < k.a ? 1.0f : -1.0f;
    } else if (OSC == S2R_OSC_TRIANGLE) {
        const float first = __builtin_fmaf(k.b, x, 1.0f);
        const float second = __builtin_fmaf(k.c, x - k.a, -1.0f);
        return x < k.a ? first : second;
    } else {
        // lookup.rs:46-85 table_lookup_exclusive_x16 on SIN_TABLE (len 1024)
        const float tv = x * 1024.0f / k.period;                // :63
        const uint32_t i1 = s2r_f32_as_u32(tv);                 // :64
        const uint32_t i2 = (i1 + 1u) & 1023u;                  // :67  (% 1024, wrapping add)
        const float2 pr = sin_pair(sSin, i2);                   // one 8-byte LDS read: SIN_TABLE[i2 - 1], SIN_TABLE[i2]
        const float s1 = i1 < 1024u ? pr.x : 0.0f;              // :72 gather_or_default
        const float s2 = pr.y;
        return __builtin_fmaf((s2 - s1) / 1.0f, tv - (float)i1, s1);   // :75-84
    }
}

// OSC == S2R_OSC_ANY: the oscillator kind is a per-lane run-time value (patch banks)
constexpr int S2R_OSC_ANY = 15;      // (above every s2r_osc_kind)

template <int OSC>
__device__ __forceinline__ OscK make_osck_any(int kind, float period) {
    if (OSC != S2R_OSC_ANY) return make_osck<OSC>(period);
    OscK k;
    k.period = period;
    k.inv_period = 1.0f / period;
    k.a = k.b = k.c = 0.0f;
    if (kind == S2R_OSC_SAW) k.a = -2.0f * k.inv_period;
    if (kind == S2R_OSC_SQUARE || kind == S2R_OSC_TRIANGLE) k.a = period * 0.5f;
    if (kind == S2R_OSC_TRIANGLE) { k.b = -4.0f * k.inv_period; k.c = 4.0f * k.inv_period; }
    return k;
}

template <int OSC>
__device__ __forceinline__ float osc_value_any(int kind, const OscK &k, float off, const float *sSin) {
    if (OSC != S2R_OSC_ANY) return osc_value<(OSC == S2R_OSC_ANY ? 0 : OSC)>(k, off, sSin);
    if (kind == S2R_OSC_SAW) return osc_value<S2R_OSC_SAW>(k, off, sSin);
    if (kind == S2R_OSC_SQUARE) return osc_value<S2R_OSC_SQUARE>(k, off, sSin);
    if (kind == S2R_OSC_TRIANGLE) return osc_value<S2R_OSC_TRIANGLE>(k, off, sSin);
    return osc_value<S2R_OSC_SINE>(k, off, sSin);
}

// Build-defined alias-suppressed oscillators (s2r_osc_kind DPW_*; the oracle's dpw_sample): s is the
// naive saw's sample at this phase, F the integral over s of the shape (saw s^2 / 2, square |s|, triangle s (|s| - 1)),
// the output F's first difference over s's step per frame: (-0.5 * period) * (F - z).  Separately rounded f32 ops.
__device__ __forceinline__ float dpw_value(int kind, const OscK &k, float off, float &z) {
    float x = off;
    const bool slow = !(s2r_f2u(off) < s2r_f2u(k.period) && k.period > 0.0f);
    if (__builtin_expect(__ballot(slow) != 0ull, 0)) { if (slow) x = ::fmodf(off, k.period); }
    const float s = __builtin_fmaf(-2.0f * k.inv_period, x, 1.0f);       // (-2 / period) * x + 1, oscillators.rs:107-112
    const float a = __builtin_fabsf(s);
    const float F = kind == S2R_OSC_DPW_SAW ? 0.5f * (s * s) : kind == S2R_OSC_DPW_SQUARE ? a : s * (a - 1.0f);
    // (a voice's first DPW frame — its memory still the NaN that a restart leaves — differences against its own F)
    const float y = (-0.5f * k.period) * (F - (z != z ? F : z));
    z = F;
    return y;
}

// filters.rs:20-21: x = exp(-2 pi f / sr)
template <bool FASTDIV, class P = S2rRenderParams>
__device__ __forceinline__ float lpf_arg(const P &p, float f_lpf) {
    const float num = (-2.0f * 3.14159274101257324f) * f_lpf;   // -2.0 * pi * freq
    return FASTDIV ? s2r_div_const_nocheck(num, p.sr, p.rcp_sr) : (num / p.sr);
}
template <bool FASTDIV, class P = S2rRenderParams>
__device__ __forceinline__ float lpf_coeff(const P &p, float f_lpf, const uint64_t *sT) {
    return s2r_expf(lpf_arg<FASTDIV, P>(p, f_lpf), sT);
}

// filters.rs:23-33: out = a0.mul_add(input, -b1 * last) with a0 = 1 - x, b1 = -x
__device__ __forceinline__ float lpf_apply(float x, float in, float &last) {
    const float a0 = 1.0f - x;
    const float out = __builtin_fmaf(a0, in, x * last);
    last = out;
    return out;
}

// The part of a frame that is closed-form in the frame offset (no recurrence): envelopes,
// filter coefficient, noise (+ the oscillator constants under FM).  process.rs:137-174 and
// the noise/LPF-coefficient halves of process.rs:306-379.  One scalar frame:
struct FrameCF {
    float amp;       // amp envelope                         process.rs:144
    float xc;        // exp(-2 pi f_lpf / sr)                filters.rs:21
    float nz;        // noise(offset) + noise level          process.rs:347-356 (ADD)
};
// ... and kP = 4 consecutive frames at once:
struct FrameCF4 { f4 amp, xc, nz; };
struct OscK4 { f4 period, inv_period, a, b, c; };

template <int OSC>
__device__ __forceinline__ OscK4 make_osck4(f4 period) {
    OscK4 k;
    k.period = period;
    k.inv_period = splat(1.0f) / period;                         // oscillators.rs:378 (the one division, see make_osck)
    k.a = k.b = k.c = splat(0.0f);
    if (OSC == S2R_OSC_SAW) k.a = splat(-2.0f) * k.inv_period;   // oscillators.rs:107-112
    if (OSC == S2R_OSC_SQUARE) k.a = period * splat(0.5f);       // :68-69
    if (OSC == S2R_OSC_TRIANGLE) { k.a = period * splat(0.5f); k.b = splat(-4.0f) * k.inv_period; k.c = splat(4.0f) * k.inv_period; }
    return k;
}

// s2r_div_u16_by_65535 on four lanes
__device__ __forceinline__ f4 div_u16_by_65535_4(f4 value) {
    return vfma(value, splat(0x1.0001p-32f), value * splat(0x1p-16f));
}

// hashnoise.rs:33-51 on four offsets
__device__ __forceinline__ f4 hash_noise4(uint32_t seed_rot, f4 t) {
    u4 off;
    off.x = s2r_f32_as_u32(t.x); off.y = s2r_f32_as_u32(t.y); off.z = s2r_f32_as_u32(t.z); off.w = s2r_f32_as_u32(t.w);
    // only the low 16 bits of the product are used (cast::<u16>()), and they depend only on the low
    // 16 bits of the factors: a full-rate 16-bit multiply instead of the quarter-rate 32-bit one
    const us4 h = __builtin_convertvector(off ^ seed_rot, us4) * (unsigned short)0x79b9u;
    const f4 value = __builtin_convertvector(h, f4);
    return vfma(div_u16_by_65535_4(value), splat(2.0f), splat(-1.0f));
}

// The same for four offsets below 2^24, given as the low 16 bits of two pairs of them: there (offset as f32) as u32
// is the offset itself (hashnoise.rs:37 casts a value that is exact), and the hash's low 16 bits need only the low
// 16 bits of offset and seed.  Two frames per 32-bit register: one xor and one packed 16-bit multiply per pair, no
// conversions of the offset.  `seed_pair` holds the low half of rotl(seed, 5) in both halves.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b));
}
__device__ __forceinline__ f4 hash_noise4_low16(uint32_t seed_pair, uint32_t off01, uint32_t off23) {
    const us2 h01 = __builtin_bit_cast(us2, off01 ^ seed_pair) * (unsigned short)0x79b9u;
    const us2 h23 = __builtin_bit_cast(us2, off23 ^ seed_pair) * (unsigned short)0x79b9u;
    const f4 value = {(float)h01.x, (float)h01.y, (float)h23.x, (float)h23.y};
    return vfma(div_u16_by_65535_4(value), splat(2.0f), splat(-1.0f));
}

// While the mod envelope sits in a stage whose slope is zero (sustain, end, or a degenerate
// decay/release) its value is the constant y0, so everything derived from it alone — the LPF
// coefficient exp(-2 pi f / sr) and, under FM, the oscillator period constants — is the same
// number frame after frame.  It is computed once when the stage is entered (scalar routines,
// bit-identical to the vector ones) and reused while EVERY voice of the wave is in such a stage.
struct FlatCache {
    float xc;
    OscK k;
};

// The two values a flat mod envelope can have are the patch's sustain level S (sustain; a decay with S == 1; a release
// with S == 0) and +0 (after the end), so what refresh_flat() hands out is one of two per-voice constants, computed when
// the kernel starts (and again when a timed event restarts the voice with another pitch) instead of ~100 instructions
// per stage change of any lane of the wave.
struct FlatConsts {
    float xc_s, xc_0;       // filter coefficient at mod == S and at mod == +0
    OscK k_s, k_0;          // under FM: the oscillator constants at those two values
};

template <int OSC, bool FM>
__device__ __forceinline__ FlatConsts make_flat_consts(const S2rRenderParams &p, float pitch, const uint64_t *sT) {
    FlatConsts c;
    const float f_s = s2r_pow2_sleef_core(p.mod.S * p.amt_lpf) * p.lpf_freq;         // process.rs:148-152 at mod == S
    const float f_0 = s2r_pow2_sleef_core(0.0f * p.amt_lpf) * p.lpf_freq;
    c.xc_s = s2r_expf(p.fast_div_sr ? lpf_arg<true>(p, f_s) : lpf_arg<false>(p, f_s), sT);
    c.xc_0 = s2r_expf(p.fast_div_sr ? lpf_arg<true>(p, f_0) : lpf_arg<false>(p, f_0), sT);
    c.k_s = make_osck<OSC>(p.sr / (s2r_pow2_sleef_core(p.mod.S * p.amt_osc) * pitch));  // process.rs:146-147, units.rs:32-42
    c.k_0 = make_osck<OSC>(p.sr / (s2r_pow2_sleef_core(0.0f * p.amt_osc) * pitch));
    return c;
}

// ... read from the patch's coefficient tables where there are some: their sustain and end regions hold exactly these
// values (s2r_table_kernel runs the same routines on mod == 0 * 1 + S and mod == 0), at wave-uniform addresses
template <int OSC, bool FM>
__device__ __forceinline__ FlatConsts flat_consts_from_tables(const S2rRenderParams &p, float pitch) {
    FlatConsts c;
    c.xc_s = p.tab.base[p.tab.sus];
    c.xc_0 = p.tab.base[p.tab.end];
    if (FM) {
        const float *fm = p.tab.base + (size_t)p.tab.fm_plane * p.tab.plane;
        c.k_s = make_osck<OSC>(p.sr / (fm[p.tab.sus] * pitch));  // process.rs:146-147, units.rs:32-42
        c.k_0 = make_osck<OSC>(p.sr / (fm[p.tab.end] * pitch));
    } else {
        c.k_s = c.k_0 = make_osck<OSC>(p.sr / (1.0f * pitch));    // (not looked at without FM)
    }
    return c;
}

template <int OSC, bool FM>
__device__ __forceinline__ FlatCache refresh_flat(const FlatConsts &c, const EnvRun em, FlatCache fc) {
    if (em.s0 == 0.0f) {
        // mod = 0 * (t - base) + y0 == y0, and y0 is S or +0 (or 1.0 in a decay whose slope (S - 1) / D is zero: S)
        const bool zero = em.s2 == 0.0f;
        fc.xc = zero ? c.xc_0 : c.xc_s;
        if (FM) fc.k = zero ? c.k_0 : c.k_s;
    }
    return fc;
}

// frames oi .. oi+3 of one voice
template <int OSC, bool FM>
__device__ __forceinline__ void closed_form_x4(const S2rRenderParams &p, const VoiceRegs &r, EnvRun &ea, EnvRun &em,
                                               float &thr_min, FlatCache &fc, const FlatConsts &fcc, uint32_t oi,
                                               const uint64_t *sT, FrameCF4 &cf, OscK4 &k) {
    const u4 ou = (u4)(oi) + (u4){0u, 1u, 2u, 3u};               // offsets_x16: wrapping u32 add (process.rs:213-219)
    const f4 t = __builtin_convertvector(ou, f4);                // offsets as f32 (simdtest.rs:277-279, process.rs:348)
    // fast path first: the active stages' lines for all four frames
    f4 amp = splat(ea.s0) * (t - splat(ea.s1)) + splat(ea.s2);           // process.rs:144
    f4 mod = splat(em.s0) * (t - splat(em.s1)) + splat(em.s2);           // process.rs:145
    bool moving = p.no_flat_shortcut != 0;
    const bool cold = !(t.w < thr_min);
    if (__builtin_expect(__ballot(cold) != 0ull, 0)) {      // wave-uniform branch: no exec juggling when nobody is cold
        if (cold) {
            // an envelope stage ends inside these four frames: walk them one by one
            moving = true;
#define S2R_ENV_STEP(C)                                                                   \
            {                                                                             \
                const float tj = t.C;                                                     \
                if (!(tj < thr_min)) {                                                    \
                    ea = env_stage_at(p.amp, r.ro_a, r.end_a, tj);                        \
                    em = env_stage_at(p.mod, r.ro_m, r.end_m, tj);                        \
                    thr_min = __builtin_fminf(ea.s3, em.s3);                            \
                }                                                                         \
                amp.C = env_value(ea, tj);                                                \
                mod.C = env_value(em, tj);                                                \
            }
            S2R_ENV_STEP(x) S2R_ENV_STEP(y) S2R_ENV_STEP(z) S2R_ENV_STEP(w)
#undef S2R_ENV_STEP
            fc = refresh_flat<OSC, FM>(fcc, em, fc);
        }
    }
    cf.amp = amp;
    cf.nz = hash_noise4(r.seed_rot, t) + splat(p.noise_level);   // process.rs:347-356 (ADD)
    moving = moving || em.s0 != 0.0f;
    if (__ballot(moving) == 0ull) {
        // every voice of this wave has a flat mod envelope over these four frames
        cf.xc = splat(fc.xc);
        if (FM) { k.period = splat(fc.k.period); k.inv_period = splat(fc.k.inv_period);
                  k.a = splat(fc.k.a); k.b = splat(fc.k.b); k.c = splat(fc.k.c); }
        return;
    }
    const f4 f_lpf = pow2_sleef_core4(mod * splat(p.amt_lpf)) * splat(p.lpf_freq);   // process.rs:148-152
    const f4 num = splat(-2.0f * 3.14159274101257324f) * f_lpf;  // -2.0 * pi * freq   (filters.rs:21)
    const f4 arg = p.fast_div_sr ? div_const_nocheck4(num, p.sr, p.rcp_sr) : (num / splat(p.sr));   // wave-uniform choice
    cf.xc = expf4(arg, sT);
    if (FM) {
        const f4 f_osc = pow2_sleef_core4(mod * splat(p.amt_osc)) * splat(r.pitch);  // process.rs:146-147,231-250
        k = make_osck4<OSC>(splat(p.sr) / f_osc);                // units.rs:32-42
    }
}

// The recurrence step of a frame: phase accumulation, oscillator, LPF, gain.
template <int OSC>
__device__ __forceinline__ float recur_x16(const S2rRenderParams &p, VoiceRegs &r, const FrameCF &cf, const OscK &k,
                                           const float *sSin) {
    const float ph = r.phase;                                    // oscillators.rs:391-400
    r.phase = s2r_fmod1(ph + k.inv_period);
    const float off = __builtin_fmaf(k.period, ph, 0.0f);        // phased_offset_x16, :235
    const float osc = osc_value<OSC>(k, off, sSin);
    const float s = (osc + p.osc_gain) + cf.nz;                  // process.rs:342-345 (ADD), :358
    const float y = lpf_apply(cf.xc, s, r.last);                 // process.rs:363-371
    return y * cf.amp;                                           // process.rs:373-376
}

// dsp_filters.rs:12-17,82-89: the delayed inputs / outputs of the first- and second-order filters
struct Filt2 { float x1, x2, y1, y2; };

// dsp_filters.rs:25-45 (LP1), :60-80 (HP1), :99-130 (LP2), :149-180 (HP2), :199-230 (BP2): one step at cutoff f.
// That file has no `fma` switch: every operation is rounded separately, in Rust's evaluation
// order; sin/cos are the libm routines (s2r_sinf/s2r_cosf, bit-exact for every finite theta).
struct FiltCoef { float alpha, beta, gamma, k; };

__device__ __forceinline__ FiltCoef dsp_filter_coef(int kind, float damping, float sr, float cutoff) {
    const float theta = 2.0f * 3.14159274101257324f * cutoff / sr;           // 2.0 * PI * cutoff_freq / sample_rate
    FiltCoef c;
    c.k = 0.0f;
    if (kind >= S2R_FILT_SVF_LP) {
        // build-defined trapezoidal SVF (oracle/s2_oracle.c s2o_dsp_filter_process):
        // alpha, beta, gamma hold a1, a2, a3; damping == q
        const float fcl = __builtin_fminf(cutoff, 0.49f * sr);  // below Nyquist: g > 0, unconditionally stable
        const float g = s2r_tanf(3.14159274101257324f * fcl / sr);
        c.k = 1.0f / damping;
        c.alpha = 1.0f / (1.0f + g * (g + c.k));
        c.beta = g * c.alpha;
        c.gamma = g * c.beta;
        return c;
    }
    const float cs = s2r_cosf(theta);
    if (kind == S2R_FILT_BP2) {                                  // dsp_filters.rs:204-209; damping == quality_factor
        const float tq = s2r_tanf(theta / (2.0f * damping));
        c.beta = 0.5f * ((1.0f - tq) / (1.0f + tq));
        c.gamma = (0.5f + c.beta) * cs;
        c.alpha = (0.5f - c.beta) / 2.0f;
        return c;
    }
    const float sn = s2r_sinf(theta);
    if (kind == S2R_FILT_LP1 || kind == S2R_FILT_HP1) {
        c.beta = 0.0f;
        c.gamma = cs / (1.0f + sn);
        c.alpha = (kind == S2R_FILT_LP1) ? (1.0f - c.gamma) / 2.0f : (1.0f + c.gamma) / 2.0f;
        return c;
    }
    const float hd = damping / 2.0f;
    c.beta = 0.5f * ((1.0f - hd * sn) / (1.0f + hd * sn));
    c.gamma = (0.5f + c.beta) * cs;
    c.alpha = (kind == S2R_FILT_LP2) ? (0.5f + c.beta - c.gamma) / 4.0f : (0.5f + c.beta + c.gamma) / 4.0f;
    return c;
}

__device__ __forceinline__ float dsp_filter_apply(int kind, const FiltCoef &c, float x, Filt2 &f) {
    float y;
    if (kind >= S2R_FILT_SVF_LP) {                               // x1, x2: the two integrator states
        const float v3 = x - f.x2;
        const float v1 = c.alpha * f.x1 + c.beta * v3;
        const float v2 = f.x2 + c.beta * f.x1 + c.gamma * v3;
        f.x1 = 2.0f * v1 - f.x1;
        f.x2 = 2.0f * v2 - f.x2;
        return kind == S2R_FILT_SVF_LP ? v2 : kind == S2R_FILT_SVF_BP ? v1 : x - c.k * v1 - v2;
    }
    if (kind == S2R_FILT_LP1 || kind == S2R_FILT_HP1) {
        const float xs = (kind == S2R_FILT_LP1) ? x + f.x1 : x - f.x1;
        y = c.alpha * xs + c.gamma * f.y1;
        f.x1 = x; f.y1 = y;
        return y;
    }
    const float px1 = f.x1, px2 = f.x2, py1 = f.y1, py2 = f.y2;
    const float xs = (kind == S2R_FILT_LP2) ? (x + 2.0f * px1 + px2)
                   : (kind == S2R_FILT_HP2) ? (x - 2.0f * px1 + px2) : (x - px2);         // BP2: dsp_filters.rs:217-221
    y = 2.0f * (c.alpha * xs + c.gamma * py1 - c.beta * py2);
    f.x2 = px1; f.x1 = x; f.y2 = py1; f.y1 = y;
    return y;
}

__device__ __forceinline__ float dsp_filter_step(int kind, float damping, float sr, float cutoff, float x, Filt2 &f) {
    const FiltCoef c = dsp_filter_coef(kind, damping, sr, cutoff);
    return dsp_filter_apply(kind, c, x, f);
}

// One frame of process_layer (scalar "sisd" path: process.rs:101-135,252-304).  DSPF: the layer's
// filter is one of dsp_filters.rs (state in *f2) instead of the one-pole of filters.rs.
template <int OSC, bool DSPF = false, class P = S2rRenderParams>
__device__ __forceinline__ float frame_sisd(const P &p, VoiceRegs &r, uint32_t oi,
                            const uint64_t *sT, const float *sSin, Filt2 *f2 = nullptr, float *osc_z = nullptr) {
    const float t = (float)oi;
    const float rel = r.released ? (float)r.release_u : 4294967296.0f;   // envelopes.rs:35
    const float amp = adsr_scalar(p.amp, t, rel);
    const float mod = adsr_scalar(p.mod, t, rel);
    const float f_osc = s2r_pow2_libm(mod * p.amt_osc, sT) * r.pitch;    // process.rs:221-229
    const float f_lpf = s2r_pow2_libm(mod * p.amt_lpf, sT) * p.lpf_freq;
    const OscK k = make_osck_any<OSC>(p.osc_kind, p.sr / f_osc);
    const float ph = r.phase;
    const float off = __builtin_fmaf(k.period, ph, 0.0f);                // oscillators.rs:212
    float osc;
    if (OSC == S2R_OSC_ANY && osc_z != nullptr && p.osc_kind >= S2R_OSC_DPW_SAW) osc = dpw_value(p.osc_kind, k, off, *osc_z);
    else osc = osc_value_any<OSC>(p.osc_kind, k, off, sSin);
    r.phase = s2r_fmod1(ph + k.inv_period);                              // oscillators.rs:377-381
    const float osc_s = osc * p.osc_gain;                                // process.rs:287 (MULTIPLY)
    const float noise_s = (hash_noise(r.seed_rot, t)) * p.noise_level;   // process.rs:292 (MULTIPLY)
    const float s = osc_s + noise_s;
    if (DSPF) return dsp_filter_step(p.lpf_kind, p.lpf_damping, p.sr, f_lpf, s, *f2) * amp;
    const float x = lpf_coeff<false, P>(p, f_lpf, sT);
    const float y = lpf_apply(x, s, r.last);
    return y * amp;
}


// The end of a fill's last kernel (S2rDone): every workgroup, once its part of the output has left for host memory,
// counts itself in; the last one of `n_workgroups` resets the counter and stores the fill's sequence number where the
// host is polling.  Called by every thread of the workgroup.  The output lives in mapped, coherent host memory
// (hipHostMallocCoherent, asked for explicitly: s2r_host.cpp kHostPolled) and is stored at system scope (out_store): a wave's
// `s_waitcnt vmcnt(0)` then says those stores are performed in host memory, and the flag store that follows the last such
// wait cannot overtake them — no agent- or system-scope cache write-back (2-6 us with a fill's partial rows freshly
// dirtied in the L2) is needed.
// diagnostic builds (-DS2R_STAMPS): a launch's first entry and last exit on the GPU's own 100 MHz clock (tools/gpu_timeline.py)
__device__ __forceinline__ void tl_mark(unsigned long long *timeline, uint32_t slot, int which) {
#if defined(S2R_STAMPS)
    if (timeline && threadIdx.x == 0) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        if (which == 0) atomicMin(&timeline[2u * slot], t); else atomicMax(&timeline[2u * slot + 1u], t);
    }
#else
    (void)timeline; (void)slot; (void)which;
#endif
}

// ... so the output itself is stored at SYSTEM scope when a completion word follows it (global_store ... sc0 sc1: written
// through to host memory, and acknowledged to the wave — vmcnt — only once performed there; a plain store to mapped
// host memory may wait in the L2 until the kernel ends, which is later than the flag)
__device__ __forceinline__ void out_store(const S2rDone &d, float *p, float v) {
    if (d.flag != nullptr) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else *p = v;
}

__device__ __forceinline__ void signal_done(const S2rDone &d, uint32_t n_workgroups) {
    if (d.flag == nullptr) return;                               // (uniform over the launch)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's output stores are acknowledged
    __syncthreads();
    if (n_workgroups == 1u) {                                    // nobody to count: the flag alone (an atomic's round trip less)
        if (threadIdx.x == 0) __hip_atomic_store(d.flag, d.value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (threadIdx.x == 0) {
        const uint32_t arrived = __hip_atomic_fetch_add(d.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived + 1u == n_workgroups) {
            __hip_atomic_store(d.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(d.flag, d.value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Two streams (S2rOverlapWords, s2r_device.h): data handed from the workgroups of one kernel to those of another that runs
// beside it, possibly on another XCD, whose L2 is not coherent with this one's.  The form used is the write-through one of
// MI355X_MICROARCH.md ("Valid forms", first row of its table; measured on gfx950, cheaper by microseconds per workgroup
// than an agent-scope release fence, which writes back the XCD's whole L2):
//   producer  EVERY handed-off byte is stored `sc1` (ov_store / ov_store4: write-through, the line leaves this L2); every
//             storing wave waits for its stores (`s_waitcnt vmcnt(0)`); a workgroup barrier; ONE lane adds to the counter
//             (agent-scope atomic) — ov_signal, called by every thread of the workgroup;
//   consumer  ONE lane polls the counter with relaxed `sc1` loads (ov_wait; a few polls per microsecond per workgroup: the
//             pollers must not eat the fabric) until it has reached the value the host computed — counters only grow, the
//             comparison survives their wrap; a workgroup barrier (the caller's); then EVERY load of the handed-off bytes
//             is an `sc1` load (ov_load), never a plain one.
// Every wait is bounded: 50 ms of the device's 100 MHz clock, then the waiter raises the fill's `fail` word and goes on.
__device__ __forceinline__ bool ov_wait(const uint32_t *counter, uint32_t target) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int32_t)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) return false;
        __builtin_amdgcn_s_sleep(100);
    }
    return true;
}
__device__ __forceinline__ void ov_signal(uint32_t *counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ov_raise(uint32_t *fail, uint32_t who) {      // who: 1 a render kernel (its chain heads), 2 a mix (its rows)
    if (fail) __hip_atomic_store(fail, who, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <typename T> __device__ __forceinline__ T ov_load(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void ov_store(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// sixteen bytes with one write-through store (a dword store of this kind is one fabric write each: six times the time per byte)
// (the two wait states behind it are the hardware's: a store of more than 64 bits reads its data registers late, and a vector
// instruction that writes one of them may not follow within two states on gfx940-class parts — a hazard the compiler pads for
// its own stores and cannot see inside this statement; found as every fourth record of fused_heads carrying the NEXT
// instruction's result in its second word)
__device__ __forceinline__ void ov_store4(float *p, f4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

// a record of the fill's event copy: plainly, or (two streams: the copy comes from a kernel on the other one) `sc1`
__device__ __forceinline__ S2rTimedEvent tev_load(const S2rTimedEvent *tev, bool ovh, int32_t i) {
    if (!ovh) return tev[i];
    const uint32_t *w = reinterpret_cast<const uint32_t *>(tev + i);
    S2rTimedEvent e;
    e.voice = 0u; e._pad = 0u;
    e.frame = ov_load(w + 1); e.flags = ov_load(w + 2); e.pitch = s2r_u2f(ov_load(w + 3));
    e.seed = ov_load(w + 4); e.next = (int32_t)ov_load(w + 5); e.program = ov_load(w + 6);
    return e;
}

// minimum of a value over the wavefront, in a scalar register: four DPP row shifts, two row broadcasts (the classic
// GFX9 reduction; ~35 cycles of issue for a lone wave where a ballot round trip costs ~50 and answers less)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#define S2R_MIN_DPP(CTRL, ROWMASK)                                                                                   \
    { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, CTRL, ROWMASK, 0xf, false);  \
      v = o_ < v ? o_ : v; }
    S2R_MIN_DPP(0x111, 0xf)      // row_shr:1
    S2R_MIN_DPP(0x112, 0xf)      // row_shr:2
    S2R_MIN_DPP(0x114, 0xf)      // row_shr:4
    S2R_MIN_DPP(0x118, 0xf)      // row_shr:8   -> lane 15 of each row holds the row's minimum
    S2R_MIN_DPP(0x142, 0xa)      // row_bcast:15 into rows 1 and 3
    S2R_MIN_DPP(0x143, 0xc)      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's minimum
#undef S2R_MIN_DPP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// 16-byte load from a dword-aligned address (global_load_dwordx4 needs no more on gfx950)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ f4 load_f4u(const float *p) { return *reinterpret_cast<const f4u *>(p); }

// Where a lane's coefficients for frame offset `o` sit in the tables: entry idx + (o & mask) (S2rTabRef).  `stage` is
// the mod envelope's active stage; a release that starts at the clamp attack + decay (simdtest.rs:283) is indexed by
// the offset, a later one by the frames since the release.
struct TabCur { int32_t idx; uint32_t mask; };
__device__ __forceinline__ TabCur tab_cursor(const S2rTabRef &t, float stage, bool late_release, uint32_t release_u, bool live) {
    TabCur c;
    const bool moving = stage <= 1.0f || stage == 3.0f;
    c.mask = (live && moving) ? 0xffffffffu : 0u;
    const int32_t rel = late_release ? t.ru - (int32_t)release_u : t.rc - (int32_t)t.rc_t0;
    const int32_t idx = stage <= 1.0f ? t.ad : stage == 2.0f ? t.sus : stage == 3.0f ? rel : t.end;
    c.idx = live ? idx : t.dead;
    return c;
}

// Super-chunk length of a render launch (host side).  Every super-chunk boundary costs each wave a barrier and a fresh
// run decision — about three 16-frame chunks' worth (tools/stamps.py) — so where the grid has no more workgroups than
// the device has compute units (one workgroup per CU whatever its LDS size: 65 536 voices in 256-voice workgroups on
// an MI355X) the whole fill is ONE super-chunk: group sums for up to 1024 frames in LDS (64 KiB, one buffer), one
// barrier, one combine.  Bigger grids keep 256-frame super-chunks in two buffers, small enough for two workgroups per CU.
inline uint32_t s2r_pick_super_frames(uint32_t n_groups, uint32_t grid, uint32_t frames) {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n_cu = v;
        else n_cu = 1;
    }
    if (n_groups <= 16 && grid <= (uint32_t)n_cu && frames <= kSuperWhole) return kSuperWhole;
    return n_groups <= 16 ? kSuperMax : (n_groups <= 32 ? 64u : 32u);
}
// group-sum buffers the launch needs: two (a super-chunk's combine overlaps the next one's chunks) unless the fill is one
__host__ __device__ inline uint32_t s2r_sw_buffers(uint32_t frames, uint32_t super_frames) { return frames > super_frames ? 2u : 1u; }

// What changes from one fill of a handle to the next, as the render kernels' fill loop sees it: the kernel arguments' own
// values when every fill is a launch; the posted command's when a resident kernel renders fill after fill (s2r_resident_kernel).
struct FillCtl {
    uint32_t frames;         // this fill's length (<= the launch's p.frames, which sizes the staging)
    uint32_t n_events;       // untimed note events applied before the state is loaded (apply_arg_events)
    const uint32_t *ev;      // [3 * n_events]: voice, flags, pitch bits
    S2rDone done;            // the fill's completion word
    // Short fills of the resident kernel: the output as GRANULES in mapped host memory — one naturally aligned 8-byte word
    // per frame, the sample's bits below and the fill's tag above, each written by ONE 8-byte system-scope store and read by
    // the host with one 8-byte load: a frame is there when its tag is, so the fill needs no completion word and nobody
    // waits for a store to be acknowledged across the link (MI355X_MICROARCH.md's R2 granule).  nullptr: not used.
    unsigned long long *granules;
    uint32_t granule_tag;
    // the fill's buffers (by parity where two fills are in flight)
    float *partials;         // [n_blocks][frames_stride]: the workgroups' partial rows
    int32_t *heads;          // [padded voices]: the voices' chain heads
    const S2rTimedEvent *tev;   // the fill's records in HBM
    // one launch per fill (S2rMixTail, s2r_device.h): this workgroup's slice of the fill's records and the in-kernel mix
    bool fused;
    const S2rTimedEvent *tev_src;
    S2rTimedEvent *tev_copy;
    uint32_t slice_lo, slice_hi;
    uint32_t *arrive;
    uint32_t arrive_target;
    uint32_t *fail;
    S2rMixTail mt;
    // two streams (S2rOverlapWords): the chain heads come from a kernel on the other stream, the rows go to a mix there
    const uint32_t *ov_heads_counter;
    uint32_t ov_heads_target;
    uint32_t *ov_render_counter;
};

// a launch is a fill: everything from the kernel arguments
__device__ __forceinline__ FillCtl fill_ctl_from_args(const S2rRenderArgs &a) {
    const S2rRenderParams &p = a.p;
    FillCtl c;
    c.frames = p.frames; c.n_events = a.n_events; c.ev = a.ev; c.done = p.done; c.granules = nullptr; c.granule_tag = 0u;
    c.partials = p.block_partials; c.heads = p.voice_ev_head; c.tev = p.tev;
    c.fused = p.arrive != nullptr;
    c.tev_src = p.tev_src; c.tev_copy = p.tev_copy;
    c.slice_lo = 0u; c.slice_hi = 0u;
    if (c.fused && p.tev_src != nullptr) {
        const uint32_t *sl = p.slices ? p.slices : a.ev;
        c.slice_lo = sl[blockIdx.x]; c.slice_hi = sl[blockIdx.x + 1u];
    }
    c.arrive = p.arrive; c.arrive_target = p.arrive_target; c.fail = p.ov_fail;
    c.mt = p.mt;
    c.ov_heads_counter = p.ov_heads_counter; c.ov_heads_target = p.ov_heads_target; c.ov_render_counter = p.ov_render_counter;
    return c;
}

// ---------------------------------------------------------------------------------------
// The fill's note events when they ride in the kernel arguments (S2rRenderArgs): every wave looks at 64 records per
// step, one per lane; the few that hit one of its 64 voices are handed to their lane, which rewrites its voice's
// words in HBM — restart: *voice = Voice { .. } (synth.rs:63-69), release: release_frame_offset = current_frame_offset
// (synth.rs:74-75) — before the kernel loads its state.  The host folds a fill's events to at most one record per
// voice (s2r_host.cpp push_event), so no two lanes of the grid write the same voice.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void apply_arg_events(const S2rRenderParams &p, uint32_t n_events, const uint32_t *ev, uint32_t vi, uint32_t lane) {
    if (n_events == 0u) return;
    const uint32_t group = __builtin_amdgcn_readfirstlane(vi >> 6);
    uint32_t my_flags = 0u, my_pitch = 0u;
    for (uint32_t k0 = 0; k0 < n_events; k0 += 64u) {          // wave-uniform
        const uint32_t i = k0 + lane;
        const bool have = i < n_events;
        const uint32_t ev_voice = have ? ev[3u * i] : 0xffffffffu;
        const uint32_t ev_flags = have ? ev[3u * i + 1u] : 0u;
        const uint32_t ev_pitch = have ? ev[3u * i + 2u] : 0u;
        uint64_t hits = __ballot(ev_voice != 0xffffffffu && (ev_voice >> 6) == group);
        while (hits) {                                           // wave-uniform
            const int src = __builtin_ctzll(hits);
            hits &= hits - 1ull;
            const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)ev_voice, src);
            const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)ev_flags, src);
            const uint32_t pb = (uint32_t)__builtin_amdgcn_readlane((int)ev_pitch, src);
            if ((v & 63u) == lane) { my_flags = f; my_pitch = pb; }
        }
    }
    if (my_flags == 0u || vi >= p.n_voices) return;
    if (my_flags & S2R_EV_RESTART) {
        p.v.pitch[vi] = s2r_u2f(my_pitch);
        p.v.offset[vi] = 0u;
        p.v.release[vi] = 0u;                                    // a release right after the on is at offset 0
        p.v.flags[vi] = S2R_VF_STARTED | ((my_flags & S2R_EV_RELEASE) ? S2R_VF_RELEASED : 0u);
        p.v.phase[vi] = 0.0f;
        p.v.lpf_last[vi] = 0.0f;
        p.v.fx1[vi] = 0.0f; p.v.fx2[vi] = 0.0f; p.v.fy1[vi] = 0.0f; p.v.fy2[vi] = 0.0f;
        p.v.seed[vi] = 0u;                                       // this path carries no seed overrides
        p.v.program[vi] = my_flags >> S2R_EV_PROGRAM_SHIFT;
        p.v.osc_z[vi] = s2r_u2f(S2R_OSC_Z_NONE);
    } else if (my_flags & S2R_EV_RELEASE) {
        const uint32_t fl = p.v.flags[vi];
        if ((fl & S2R_VF_STARTED) && !(fl & S2R_VF_RELEASED)) {
            p.v.release[vi] = p.v.offset[vi];
            p.v.flags[vi] = fl | S2R_VF_RELEASED;
        }
    }
}

// ---------------------------------------------------------------------------------------

// Four frames of one quad into the wave's [16 frames][64 voices + 1] transpose tile, voice per lane: frame f of lane l
// goes to tile[f * kTileRow + l].  ds_write_addtid_b32 (LDS address = M0 + offset + 4 * lane, no address register) costs a
// lone wave 8 cycles of issue where ds_write_b32 costs 16 (tools/ubench/issue_rates3.hip) — with one wave per SIMD
// the store of every voice-frame is a sixth of the chunk otherwise.  tile_set_base() puts the tile's LDS byte address
// (wave-uniform) into M0 once per chunk; nothing the compiler generates for these kernels touches M0 in between
// (tests/test_abi_c.py::test_m0_is_ours_between_the_tile_stores checks the disassembly).
// (No "m0" in the clobber list, deliberately: declaring it makes the compiler save and restore M0 around every one of these
// statements; that nothing else in the kernels reads or writes M0 between the set and the stores is what
// tests/test_isa_contracts.py asserts on the shipped ISA instead.)
__device__ __forceinline__ void tile_set_base(uint32_t tile_m0) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(tile_m0) : "memory");
}
__device__ __forceinline__ void tile_store4(int q, f4 v) {
#define S2R_ST4(Q)                                                                                              \
    asm volatile("ds_write_addtid_b32 %0 offset:%c4\n\tds_write_addtid_b32 %1 offset:%c5\n\t"                     \
                 "ds_write_addtid_b32 %2 offset:%c6\n\tds_write_addtid_b32 %3 offset:%c7"                          \
                 : : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "n"((4 * Q) * (kTileRow * 4)), "n"((4 * Q + 1) * (kTileRow * 4)),         \
                     "n"((4 * Q + 2) * (kTileRow * 4)), "n"((4 * Q + 3) * (kTileRow * 4)) : "memory")
    switch (q) { case 0: S2R_ST4(0); break; case 1: S2R_ST4(1); break; case 2: S2R_ST4(2); break; default: S2R_ST4(3); break; }
#undef S2R_ST4
}

// Four sine-table lookups (lookup.rs:46-85 table_lookup_exclusive_x16 on SIN_TABLE): the four LDS reads are issued
// together and waited for once (left to itself the compiler waits behind each read, and an `s_waitcnt lgkmcnt(0)` also
// waits for the tile stores it does not count).  x * 1024 / period: through the run's double reciprocal, exactly
// (s2r_math.h), or a true division when the period changes every frame (FMV).
struct SineQuad { float tv[4]; uint32_t i1[4]; float2 pr[4]; };
template <bool FMV>
__device__ __forceinline__ SineQuad sine_quad_issue(f4 x, f4 period, double rcp_period, const float *sSin) {
    SineQuad s;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s.tv[j] = FMV ? x[j] * 1024.0f / period[j] : s2r_div_by_rcp64(x[j] * 1024.0f, rcp_period);    // :63
        s.i1[j] = s2r_f32_as_u32(s.tv[j]);                       // :64
    }
    // :67 i2 = (i1 + 1) % 1024 as the pair's byte offset ((i1 + 1) & 1023) * 8 == ((i1 << 3) + 8) & 0x1ff8: two instructions
#pragma unroll
    for (int j = 0; j < 4; ++j)
        s.pr[j] = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(sSin) + (((s.i1[j] << 3) + 8u) & 0x1ff8u));
    return s;
}
__device__ __forceinline__ f4 sine_quad_finish(SineQuad &s) {
    asm("" : "+v"(s.pr[0].x), "+v"(s.pr[0].y), "+v"(s.pr[1].x), "+v"(s.pr[1].y), "+v"(s.pr[2].x), "+v"(s.pr[2].y), "+v"(s.pr[3].x), "+v"(s.pr[3].y));
    f4 out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // s1 = i1 < 1024 ? SIN_TABLE[i1] : 0 (:68-70) without the compare: in the branch-free chunks 0 <= x < period
        // (chunk_fast), so tv = RN(x * 1024 / period) <= 1024 and i1 <= 1024; for i1 == 1024 the pair read is entry
        // i2 = 1, whose first half is SIN_TABLE[0] == +0.0 — the very value the compare would pick
        const float s1 = s.pr[j].x, s2 = s.pr[j].y;
        out[j] = __builtin_fmaf((s2 - s1) / 1.0f, s.tv[j] - (float)s.i1[j], s1);
    }
    return out;
}
template <bool FMV>
__device__ __forceinline__ f4 sine_quad(f4 x, f4 period, double rcp_period, const float *sSin) {
    SineQuad s = sine_quad_issue<FMV>(x, period, rcp_period, sSin);
    return sine_quad_finish(s);
}

// v_pk_add_f32 on a pair (the compiler splits some of the chunk's packed adds into two scalar ones)
typedef float fp2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f4 pk_add4(f4 a, f4 b) {
    fp2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(a.xy), "v"(b.xy));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(a.zw), "v"(b.zw));
    return (f4){lo.x, lo.y, hi.x, hi.y};
}

// The workgroup's cross-wave combine: per frame the block's 16-voice group sums in group (= voice index) order
// (synth.rs:177-195's order), from sWbuf[n_groups][super_frames] into the block's partial row (and, for a one-workgroup
// launch, through (+0.0) + sum — accum = splat(0.0), synth.rs:176 — into the output: DESIGN.md 4.3).  A thread takes four
// consecutive frames: one 16-byte LDS read per group, four groups' reads in flight at a time, packed adds (the same IEEE
// additions in the same order as one frame at a time; a loop of dependent 4-byte reads cost every wave ~6 000 cycles per
// 1 024 frames, tools/stamps.py).  n_groups is a multiple of 4 (four groups per wave), super_frames of 4; the rows'
// entries past n_sc are read and not used.  bp: the block's row at the super-chunk's first frame, 16-byte aligned.
__device__ __forceinline__ void combine_groups(const S2rRenderParams &p, const S2rDone &done, const float *sWbuf, uint32_t n_groups, uint32_t super_frames,
                                               uint32_t n_sc, uint32_t sc0, float *bp_sc, uint32_t tid, uint32_t n_threads,
                                               unsigned long long *granules = nullptr, uint32_t granule_tag = 0u, bool wt = false) {
    for (uint32_t f = 4u * tid; f < n_sc; f += 4u * n_threads) {
        const float *row = sWbuf + f;
        f4 acc;
        {
            const f4 v0 = *reinterpret_cast<const f4 *>(row), v1 = *reinterpret_cast<const f4 *>(row + super_frames),
                     v2 = *reinterpret_cast<const f4 *>(row + 2u * super_frames), v3 = *reinterpret_cast<const f4 *>(row + 3u * super_frames);
            acc = pk_add4(pk_add4(pk_add4(v0, v1), v2), v3);
        }
        for (uint32_t g = 4u; g < n_groups; g += 4u) {
            const float *rg = row + (size_t)g * super_frames;
            const f4 v0 = *reinterpret_cast<const f4 *>(rg), v1 = *reinterpret_cast<const f4 *>(rg + super_frames),
                     v2 = *reinterpret_cast<const f4 *>(rg + 2u * super_frames), v3 = *reinterpret_cast<const f4 *>(rg + 3u * super_frames);
            acc = pk_add4(pk_add4(pk_add4(pk_add4(acc, v0), v1), v2), v3);
        }
        const uint32_t n = n_sc - f < 4u ? n_sc - f : 4u;
        if (wt) {                                                // the row is handed to workgroups of another compute unit (a mix beside or behind this kernel)
            if (n == 4u) ov_store4(bp_sc + f, acc);
            else for (uint32_t j = 0; j < n; ++j) ov_store(bp_sc + f + j, acc[j]);
        } else if (n == 4u) *reinterpret_cast<f4 *>(bp_sc + f) = acc;
        else for (uint32_t j = 0; j < n; ++j) bp_sc[f + j] = acc[j];
        if (p.direct_out) {                                      // one workgroup: this IS the mix
            for (uint32_t j = 0; j < n; ++j) {
                const float total = 0.0f + acc[j];
                const uint32_t fo = sc0 + f + j;
                if (granules != nullptr) {                       // (mono: the host doubles the frame for a stereo caller)
                    __hip_atomic_store(granules + fo, ((unsigned long long)granule_tag << 32) | (unsigned long long)s2r_f2u(total),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    continue;
                }
                if (p.direct_stereo) { out_store(done, p.direct_out + 2u * fo, total); out_store(done, p.direct_out + 2u * fo + 1u, total); }
                else out_store(done, p.direct_out + fo, total);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// The mix of the workgroups' partial rows in the fixed order of DESIGN.md 4.3:
//   runs of 16 consecutive workgroups sequentially -> the run sums of a mix group sequentially
//   -> the mix groups sequentially -> root (+0.0) + total.
// One workgroup (256 threads) handles one block of 16 frames: thread (slot, f) adds whole runs (16 independent loads in
// flight each), the run sums meet in LDS (s_run: [total runs][16 frames]), 16 threads finish.  Runs never straddle a mix
// group.  `ov`: the rows come from workgroups of other compute units that may still be running (sc1 loads, ov_load).
// `sys`: the output goes to memory the host or another device reads behind a flag (system-scope stores).
// ---------------------------------------------------------------------------------------
constexpr uint32_t kMixRun = 16;

__device__ __forceinline__ void mix_block(const S2rMixParams &m, uint32_t block, float *s_run, bool ov, bool sys) {
    const uint32_t f_local = threadIdx.x & 15u, slot = threadIdx.x >> 4;
    const uint32_t f = block * 16u + f_local;
    const uint32_t runs_per_group = (m.blocks_per_group + kMixRun - 1) / kMixRun;
    const uint32_t total_runs = runs_per_group * m.n_groups;
    if (f < m.frames) {
        for (uint32_t run = slot; run < total_runs; run += 16u) {
            const uint32_t g = run / runs_per_group, rg = run % runs_per_group;
            const uint32_t gb0 = g * m.blocks_per_group;
            uint32_t gb1 = gb0 + m.blocks_per_group; if (gb1 > m.n_blocks) gb1 = m.n_blocks;
            const uint32_t b0 = gb0 + rg * kMixRun;
            float v[kMixRun];
#pragma unroll
            for (uint32_t j = 0; j < kMixRun; ++j) {
                const float *src = m.block_partials + (size_t)(b0 + j) * m.frames_stride + f;
                v[j] = (b0 + j < gb1) ? (ov ? ov_load(src) : *src) : 0.0f;
            }
            float acc = v[0];
#pragma unroll
            for (uint32_t j = 1; j < kMixRun; ++j) if (b0 + j < gb1) acc += v[j];
            s_run[run * 16u + f_local] = (b0 < gb1) ? acc : 0.0f;
        }
    }
    __syncthreads();
    if (threadIdx.x < 16u && f < m.frames) {
        float total = 0.0f;                                      // accum = splat(0.0), synth.rs:176
        for (uint32_t g = 0; g < m.n_groups; ++g) {
            const uint32_t gb0 = g * m.blocks_per_group;
            uint32_t gb1 = gb0 + m.blocks_per_group; if (gb1 > m.n_blocks) gb1 = m.n_blocks;
            if (gb0 >= gb1) continue;
            const uint32_t n_runs = (gb1 - gb0 + kMixRun - 1) / kMixRun;
            float acc = s_run[(g * runs_per_group) * 16u + f_local];
            for (uint32_t r = 1; r < n_runs; ++r) acc += s_run[(g * runs_per_group + r) * 16u + f_local];
            total = (m.root_add || g > 0) ? total + acc : acc;
        }
        if (m.granules != nullptr)                             // (mono: the host doubles the frame for a stereo caller)
            __hip_atomic_store(m.granules + f, ((unsigned long long)m.granule_tag << 32) | (unsigned long long)s2r_f2u(total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (sys) {
            if (m.stereo) { __hip_atomic_store(m.out + 2 * f, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            __hip_atomic_store(m.out + 2 * f + 1, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
            else __hip_atomic_store(m.out + f, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (m.stereo) { m.out[2 * f] = total; m.out[2 * f + 1] = total; }
        else m.out[f] = total;
    }
}

// ---------------------------------------------------------------------------------------
// One launch per fill (S2rMixTail, s2r_device.h).
// fused_heads: this workgroup's slice of the fill's records — mapped host memory, read at system scope: in the
// pool-resident kernel no kernel boundary stands between the host's writes and these loads — copied into HBM, and the
// chain heads of its own voices published (heads_body of s2r_aux.hip for one slice).  Write-through stores, a barrier: the
// lanes that follow their voices' chains read both `sc1` (tev_load / ov_load), whichever wave of the workgroup wrote them.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void fused_heads(const FillCtl &ctl) {
    // A lane moves HALF a record — 16 bytes, so that a wave reads one contiguous kilobyte of host memory (the link serves
    // eight 128-byte reads, not five hundred 4-byte ones: found as 27 us on the fill's busiest workgroups) — and the lane with
    // a record's first half (voice, frame, flags, pitch) publishes the chain head.
    // Four halves per thread with all four reads in flight: the busiest workgroups (a cohort of note-ons and its releases: ~450
    // records) pay ONE trip over the link, not one per 128 records.
    const uint32_t n_halves = (ctl.slice_hi - ctl.slice_lo) * 2u;
    for (uint32_t h0 = 0; h0 < n_halves; h0 += 4u * blockDim.x) {
        f4 v[4];
        const float *src[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t h = h0 + k * blockDim.x + threadIdx.x;
            const uint32_t hc = h < n_halves ? h : 0u;
            src[k] = reinterpret_cast<const float *>(ctl.tev_src + ctl.slice_lo + (hc >> 1)) + 4u * (hc & 1u);
        }
        // ONE statement for the four loads and their wait: an output of an asm statement is, to the compiler, a value that exists
        // when the statement ends — with the wait in a statement of its own, a copy or a spill of a load's destination placed
        // between the two would move a register the data has not reached yet (at -O0 it does exactly that)
        asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
                     "global_load_dwordx4 %1, %5, off sc0 sc1\n\t"
                     "global_load_dwordx4 %2, %6, off sc0 sc1\n\t"
                     "global_load_dwordx4 %3, %7, off sc0 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]) : "memory");
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t h = h0 + k * blockDim.x + threadIdx.x;
            if (h < n_halves) {
                const uint32_t i = ctl.slice_lo + (h >> 1);
                float *dst = reinterpret_cast<float *>(ctl.tev_copy + i) + 4u * (h & 1u);
                ov_store4(dst, v[k]);
                if ((h & 1u) == 0u && (s2r_f2u(v[k].z) & S2R_TEV_FIRST)) ov_store(ctl.heads + s2r_f2u(v[k].x), (int32_t)i);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// fused_tail: called by every thread of the workgroup once its partial row and its voices' state are stored.  s_run: LDS the
// workgroup no longer needs ([total runs][16] floats); s_word: one LDS word.
__device__ __forceinline__ void fused_tail(const S2rRenderParams &p, const FillCtl &ctl, float *s_run, uint32_t *s_word, unsigned long long *st = nullptr) {
    const S2rMixTail &mt = ctl.mt;
    if (mt.n_mixers == 0u) return;
#if defined(S2R_STAMPS)
#define S2R_TAIL_STAMP(k) do { if (st && (threadIdx.x & 63u) == 0u) st[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define S2R_TAIL_STAMP(k) do { (void)st; } while (0)
#endif
    // the row (stored sc1 by combine_groups) has left this compute unit's caches once every wave's stores are acknowledged
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) *s_word = __hip_atomic_fetch_add(ctl.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t pos = *s_word - (ctl.arrive_target - mt.n_blocks);    // this workgroup's place among the fill's arrivals
    __syncthreads();                                             // (s_word is written again below)
    S2R_TAIL_STAMP(4);
#if defined(S2R_STAMPS)
    if (st && (threadIdx.x & 63u) == 0u) st[9] = pos;
#endif
    if (pos + mt.n_mixers < mt.n_blocks) return;                 // (uniform over the workgroup) not one of the last
    const uint32_t mixer = pos - (mt.n_blocks - mt.n_mixers);
    if (threadIdx.x == 0 && !ov_wait(ctl.arrive, ctl.arrive_target)) ov_raise(ctl.fail, 2u);
    // A rank of a group of processes writes its row into the ROOT's memory, where each rows slot serves every other fill: the
    // root must have added up the slot's previous fill before this one's row goes over it.  Nothing else holds a rank back — its
    // own fills end when its rows are written — so a rank two fills ahead of the root waits here (word `consumed`, behind the
    // slot's counter: the target of the last fill the root has added up from it); bounded like every wait.
    if (threadIdx.x == 0 && mt.xmode == 2) {
        const uint32_t *consumed = mt.rows_done + 2;
        const uint32_t need = mt.rows_target - mt.n_rows;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((int32_t)(__hip_atomic_load(consumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - need) < 0) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) { ov_raise(ctl.fail, 3u); break; }
            __builtin_amdgcn_s_sleep(20);
        }
    }
    __syncthreads();
    S2R_TAIL_STAMP(5);
    S2rMixParams m{};
    m.block_partials = ctl.partials; m.n_blocks = mt.n_blocks; m.blocks_per_group = mt.blocks_per_group; m.n_groups = mt.n_groups;
    m.frames = ctl.frames; m.frames_stride = p.frames_stride; m.root_add = mt.root_add; m.stereo = mt.stereo; m.out = mt.out;
    m.granules = mt.granules; m.granule_tag = mt.granule_tag;
    const uint32_t n_fb = (ctl.frames + 15u) / 16u;
    for (uint32_t b = mixer; b < n_fb; b += mt.n_mixers) { mix_block(m, b, s_run, true, true); __syncthreads(); }
    if (mt.granules != nullptr) return;                          // (every frame carries the fill's tag: nothing to wait for, nothing to signal)
    // this mixer's part of the output is on its way; the last mixer to get here ends the fill
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    S2R_TAIL_STAMP(6);
    if (mt.n_mixers > 1u) {
        if (threadIdx.x == 0) {
            const uint32_t arrived = __hip_atomic_fetch_add(mt.done.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = arrived + 1u == mt.n_mixers;
            if (last) __hip_atomic_store(mt.done.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_word = last ? 1u : 0u;
        }
        __syncthreads();
        if (*s_word == 0u) return;
    }
    if (mt.rows_done == nullptr) {
        if (threadIdx.x == 0 && mt.done.flag != nullptr) __hip_atomic_store(mt.done.flag, mt.done.value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    // exchange: this shard's row is in the root's memory (system-scope stores, acknowledged); count in.
    //   xmode 0 (the shards of ONE process's device list): whoever counts in LAST, whichever shard that is, adds the rows in
    //     shard order.  Nobody waits for anybody: shards whose kernels share a hardware queue, or run one after the other, still
    //     finish.
    //   xmode 1 / 2 (one process per GPU; the output is host memory of the ROOT's process, which only the root's kernels
    //     reach): a rank other than the root (2) counts in and reports its own completion; the root's last mixer (1) counts in,
    //     WAITS for every rank — bounded; the ranks' kernels run on devices of their own — and adds the rows.
    if (threadIdx.x == 0) {
        const uint32_t before = __hip_atomic_fetch_add(mt.rows_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        uint32_t mine = (before + 1u == mt.rows_target) ? 1u : 0u;
        if (mt.xmode == 2) {
            if (mt.done.flag != nullptr) __hip_atomic_store(mt.done.flag, mt.done.value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            mine = 0u;
        } else if (mt.xmode == 1) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            mine = 1u;
            while ((int32_t)(__hip_atomic_load(mt.rows_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - mt.rows_target) < 0) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) { ov_raise(ctl.fail, 3u); break; }
                __builtin_amdgcn_s_sleep(20);
            }
        }
        *s_word = mine;
    }
    __syncthreads();
    if (*s_word == 0u) return;
    for (uint32_t f = threadIdx.x; f < ctl.frames; f += blockDim.x) {
        float total = 0.0f;                                      // accum = splat(0.0), synth.rs:176
        for (uint32_t r = 0; r < mt.n_rows; ++r) total += __hip_atomic_load(mt.rows + (size_t)r * mt.row_stride + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (mt.final_stereo) { __hip_atomic_store(mt.final_out + 2u * f, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                               __hip_atomic_store(mt.final_out + 2u * f + 1u, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        else __hip_atomic_store(mt.final_out + f, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        // (the rows have been read: the slot is the ranks' again)
        if (mt.xmode == 1) __hip_atomic_store(mt.rows_done + 2, mt.rows_target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (mt.final_done.flag != nullptr) __hip_atomic_store(mt.final_done.flag, mt.final_done.value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The pool-resident kernels' command loop (S2rPool, s2r_device.h): the shard's whole grid stays on the device; every workgroup
// renders fill after fill — `fill(ctl)`: a render kernel's per-fill function in its one-launch form (a fill's note events, timed
// or not, come as chains), or its two-stream form —
// as the host posts them, running ahead of the slower workgroups by as much as the fills in flight allow.  What happens to
// command last + 1 is decided once for the grid (word `decided`): run it, or — no command for idle_ticks — everybody leaves.
template <class Fill>
__device__ __forceinline__ void pool_loop(const S2rRenderArgs &a, const S2rPool &pl, Fill fill) {
    __shared__ uint32_t s_cmd[S2R_POOL_CMD_WORDS + 2];           // the command, then this workgroup's slice bounds
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint32_t last = pl.first_seq - 1u;
    for (;;) {
        if (tid < 64u) {
            const uint32_t want = last + 1u;
            const uint32_t slot = want % S2R_POOL_CMD_SLOTS;
            const uint32_t *c = pl.cmd + slot * S2R_POOL_CMD_WORDS;
            const uint32_t run_word = want << 1, bail_word = (want << 1) | 1u;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            uint32_t polls = 0;
            bool leave = false, decided_run = false;
            for (;;) {
                uint32_t w = 0u;
                // (the command lives behind the BAR in device memory, or in host memory: workgroup 0 looks at it every time
                // round, the others mostly at the decision word, which is local)
                const bool look = blockIdx.x == 0u || decided_run || gridDim.x <= 8u || (polls & 7u) == 0u;
                if (look && lane < S2R_POOL_CMD_WORDS) w = __hip_atomic_load(c + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const uint32_t s_first = (uint32_t)__builtin_amdgcn_readlane((int)w, 0), s_last = (uint32_t)__builtin_amdgcn_readlane((int)w, 15);
                const bool complete = look && s_first == want && s_last == want;
                uint32_t d = 0u;
                if (lane == 0u) {
                    d = __hip_atomic_load(pl.decided, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (d == (last << 1) && complete) {          // undecided, and the command is here: run it
                        uint32_t expected = last << 1;
                        d = __hip_atomic_compare_exchange_strong(pl.decided, &expected, run_word, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? run_word : expected;
                    }
                }
                d = (uint32_t)__builtin_amdgcn_readlane((int)d, 0);
                if (d == bail_word) { leave = true; break; }
                // Decisions only move forward, and a workgroup may be MORE than one behind the grid's: the one that adds up a fill's
                // rows for a group of processes waits for the other ranks' (tens of milliseconds at worst) while its neighbours run
                // the next command and — nothing further posted, patience 2 ms — decide to leave at the one after.  A decision about
                // a LATER command says this one was run.
                decided_run = d == run_word || (int32_t)((d >> 1) - want) > 0;
                if (decided_run && complete) {
                    if (lane < S2R_POOL_CMD_WORDS) s_cmd[lane] = w;
                    // (this workgroup's slice of the fill's records: a trip to host memory, spared when the fill brings none)
                    const uint32_t n_rec = (uint32_t)__builtin_amdgcn_readlane((int)w, 2);
                    if (lane >= 16u && lane < 18u)
                        s_cmd[lane] = n_rec == 0u ? 0u : __hip_atomic_load(pl.slices + (size_t)slot * pl.slices_stride + blockIdx.x + (lane - 16u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                ++polls;
                const bool out_of_patience = __builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)pl.idle_ticks || polls >= pl.max_polls;
                if (out_of_patience) {
                    if (decided_run) { leave = true; break; }    // (a command decided but never seen whole: cannot happen with a live host; leave rather than spin)
                    uint32_t won = 0u;
                    if (lane == 0u) {
                        uint32_t expected = last << 1;
                        won = __hip_atomic_compare_exchange_strong(pl.decided, &expected, bail_word, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
                        if (!won && expected == bail_word) won = 1u;
                    }
                    if (__builtin_amdgcn_readlane((int)won, 0) != 0) { leave = true; break; }
                    // (somebody has just decided to run it: look for the command a little longer)
                    decided_run = true;
                    polls = pl.max_polls > 4096u ? pl.max_polls - 4096u : 0u;
                }
                // (a caller in a loop comes back within microseconds: the first polls are close together, the rest leave the fabric alone)
                if (polls < 128u) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(8);
            }
            if (leave && lane == 0u) s_cmd[1] = S2R_POOL_FLAG_EXIT << 16;
        }
        __syncthreads();
        const uint32_t w1 = s_cmd[1];
        if ((w1 >> 16) & S2R_POOL_FLAG_EXIT) break;              // (uniform: every thread reads the same LDS word)
        last = s_cmd[0];
        const uint32_t frames = w1 & 0xffffu, sel = s_cmd[4] & 0xffu, par = s_cmd[6] & 1u, es = s_cmd[5] & 3u;
        const int32_t stereo = (int32_t)((s_cmd[4] >> 8) & 1u);
        FillCtl ctl = FillCtl{};                                  // (nothing of a launch's own fill: the kernel arguments' event words stay untouched)
        ctl.frames = frames; ctl.n_events = 0u;
        ctl.partials = par ? pl.partials[1] : pl.partials[0];
        ctl.heads = par ? pl.heads[1] : pl.heads[0];
        ctl.tev_copy = par ? pl.tev_copy[1] : pl.tev_copy[0];
        ctl.tev = ctl.tev_copy;
        ctl.tev_src = es == 0u ? pl.tev_src[0] : es == 1u ? pl.tev_src[1] : es == 2u ? pl.tev_src[2] : pl.tev_src[3];
        ctl.slice_lo = s_cmd[16]; ctl.slice_hi = s_cmd[17];
        ctl.fail = pl.fail;
        const bool two_streams = ((w1 >> 16) & S2R_POOL_FLAG_TWO_STREAMS) != 0u;
        if (two_streams) {
            // the two-stream form of a fill (S2rOverlapWords): the chain heads and the mix are a launch on the other stream, as for a
            // launch per fill — nothing of them on this workgroup's path; only the render launch is gone
            ctl.fused = false;
            ctl.tev_src = nullptr; ctl.slice_lo = 0u; ctl.slice_hi = 0u;
            ctl.arrive = nullptr; ctl.mt.n_mixers = 0u;
            ctl.ov_heads_counter = s_cmd[2] ? pl.ov_heads + par : nullptr; ctl.ov_heads_target = s_cmd[11];
            ctl.ov_render_counter = pl.ov_render + par;
        } else {
        ctl.fused = true;
        ctl.ov_heads_counter = nullptr; ctl.ov_render_counter = nullptr;
        ctl.arrive = pl.arrive + par; ctl.arrive_target = s_cmd[7];
        ctl.mt = pl.mt;
        ctl.mt.n_mixers = s_cmd[8];
        float *const o = sel == 0u ? pl.out[0] : sel == 1u ? pl.out[1] : pl.out[2];
        if (pl.mt.rows_done != nullptr) {                        // a shard of a device list / of a group of processes
            const uint32_t rs = s_cmd[10] & 1u;
            ctl.mt.out = rs ? pl.rows_mine[1] : pl.rows_mine[0];
            ctl.mt.rows = rs ? pl.rows[1] : pl.rows[0];
            ctl.mt.rows_done = pl.mt.rows_done + rs;
            ctl.mt.rows_target = s_cmd[9];
            ctl.mt.stereo = 0;
            ctl.mt.done = S2rDone{pl.mt.xmode == 2 ? pl.done_flag + sel : nullptr, s_cmd[3], pl.done_counter + sel};
            ctl.mt.final_out = sel == 0u ? pl.final_out[0] : sel == 1u ? pl.final_out[1] : pl.final_out[2];
            ctl.mt.final_done = S2rDone{pl.final_flag + sel, s_cmd[3], nullptr};
            ctl.mt.final_stereo = stereo;
        } else {
            ctl.mt.out = o;
            ctl.mt.stereo = stereo;
            ctl.mt.done = S2rDone{pl.done_flag + sel, s_cmd[3], pl.done_counter + sel};
            if (s_cmd[12] != 0u && pl.granules != nullptr) { ctl.mt.granules = pl.granules; ctl.mt.granule_tag = s_cmd[3]; }
        }
        }
        // (ONE call site: the fill function is the whole render kernel, and a second copy of it — or a call through the lambda's
        // address — costs the kernel arguments a copy in scratch)
        if (frames != 0u && frames <= a.p.frames) fill(ctl);
        __syncthreads();                                         // (the command words and the staging are reused)
    }
    if (blockIdx.x == 0u && tid == 0u) __hip_atomic_store(pl.exited, pl.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------
// The branch-free 16-frame chunk.  Measured (ablated builds): with the rare branches (envelope
// stage change, fmodf slow path, coefficient-source selection) inside the per-quad loop the
// SAME executed work takes almost twice as long — every one is a basic-block boundary the
// scheduler cannot move work across, and a taken branch is an instruction-fetch bubble for the
// single wave a SIMD holds.  So the decision is taken once per chunk, wave-uniformly, and the
// common case runs this straight-line code: 4 quads, closed-form part on 4-vectors, then the
// recurrence, all in one basic block.
//   Preconditions (checked by the caller for the whole wave): no envelope threshold inside the
//   chunk; period > 0 and 0 <= phase < 1 (then fmodf(period*phase, period) is `off` itself unless
//   off == period, where it is +0); no oscillator FM.
//   SRC: 0 = every voice flat (constant coefficient), 1 = coefficient tables (`tab`: this lane's entry for the chunk's
//        first frame, S2rTabRef), 2 = compute in-lane.
// ---------------------------------------------------------------------------------------
//   FILT != 0 (general kernel, flat stages only): the layer's filter is dsp_filters.rs' / the SVF with the
//   constant coefficients `fcoef`, state in *f2, instead of the one-pole.
//   FMV (with SRC == 1, oscillator FM): the tables' last plane is pow(2, mod * mod_env_to_osc_freq); the period
//   sr / (that * pitch) and its reciprocal are per-frame values (process.rs:146-147, units.rs:32-42, oscillators.rs:378).
//   NZ >= 1 "SMALL" (one-pole kernel): every lane's offsets of the chunk are below 2^24, so the f32 offsets are exact sums and
//   the noise hash works on their low 16 bits (hash_noise4_low16), and the patch's noise level is 0.0, so adding it
//   (process.rs:353-356) changes nothing: the noise value itself is never +-0 (v / 65535 == 0.5 has no integer
//   solution), and n + 0.0 == n for every other n.
//   NZ == 2 "ALIGNED" (on top of SMALL): the chunk starts on a multiple of 16 frames and rotl(seed, 5) has a zero low
//   nibble on every lane (the reference's seed is always 0, state.rs:18-21, and s2_bin renders 16 frames at a time,
//   main.rs:138-143).  Then (seed ^ (o + j)) & 0xffff == B + j with B = (seed ^ o) & 0xffff for j = 0..15, and the
//   hashed words of the chunk's frames are 16 consecutive integers: the noise comes from the device's noise table (below),
//   entry x = ((v / 65535) * 2) - 1 for v = (x * 0x9e3779b9) & 0xffff, evaluated with the two-operation quotient
//   h / 65535 = fma(h, 0x1.0001p-32, h * 2^-16) = fma(f, 0x1.0001p-16, f), f = h * 2^-16 (hashnoise.rs:33-51,57-68).
//   AFLAT (with SMALL): every started voice of the wave sits in an amplitude stage of slope +-0 (sustain, end) for
//   the whole run, so slope * (t - base) + y0 is (+-0) + y0 with the product's sign fixed by the slope's (t >= base
//   inside a stage): one evaluation per chunk, at its first frame, is every frame's value bit for bit.
template <int OSC, int SRC, int FILT = 0, bool FMV = false, int NZ = 0, bool AFLAT = false>
__device__ __forceinline__ void chunk_fast(const S2rRenderParams &p, VoiceRegs &r, const EnvRun &ea, const EnvRun &em,
                                           const FlatCache &fc, const OscK &k, uint32_t o_chunk, const float *tab,
                                           const uint64_t *sT, const float *sSin, bool live, uint32_t tile_m0,
                                           float *pv_dst, const FiltCoef *fcoef = nullptr, Filt2 *f2 = nullptr) {
    constexpr bool SMALL = NZ >= 1, ALIGNED = NZ == 2;
    tile_set_base(tile_m0);
    // table planes: the filter's coefficients (x and 1 - x for the one-pole, alpha / beta / gamma for the others), then
    // under FM pow(2, mod * amt_osc).  Four 16-byte loads per plane (dword-aligned: the index follows the voice's
    // offset), issued first and used last
    f4 xq[4], bq[4], gq[4], pq[4], iq[4];
    if (SRC == 1) {
        const uint32_t plane = p.tab.plane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            xq[q] = load_f4u(tab + 4 * q);
            // (the one-pole's 1 - x is two packed subtractions per quad: cheaper for a lone wave than a second load,
            // ~20 cycles of issue each)
            if (FILT != 0) { bq[q] = load_f4u(tab + plane + 4 * q); gq[q] = load_f4u(tab + 2u * plane + 4 * q); }
            if (FMV) pq[q] = load_f4u(tab + p.tab.fm_plane * plane + 4 * q);
        }
    }
    const double rcp_period = (OSC == S2R_OSC_SINE && !FMV) ? s2r_rcp_f64(k.period) : 0.0;   // constant over the run: hoisted by the compiler
    // a lane without a started voice: see a0 / ampq below.  Its filter history must be 0 for that (frames
    // of the general path, which selects per frame instead, leave a running value in it)
    // — and its amplitude line is 0 * (t - 0) + 0 = +0 and, where the coefficient is the run's constant, x = 1
    // (a0 = 1 - 1 = 0): selected once per chunk instead of per quad
    // (selects, not a branch: control flow here would split the chunk's basic block)
    r.last = live ? r.last : 0.0f;
    const float ea_slope = live ? ea.s0 : 0.0f, ea_base = live ? ea.s1 : 0.0f, ea_y0 = live ? ea.s2 : 0.0f;
    const float xc0 = live ? fc.xc : 1.0f;
    f4 amp[4], nz[4];
    const float t_chunk = (float)o_chunk;
    // SMALL: the low 16 bits of the chunk's first four offsets as two pairs, and of rotl(seed, 5) twice
    const uint32_t o_lo = o_chunk & 0xffffu;
    uint32_t o01 = pk_add_u16(o_lo | (o_lo << 16), 0x00010000u), o23 = pk_add_u16(o01, 0x00020002u);
    const uint32_t seed_pair = (r.seed_rot & 0xffffu) | (r.seed_rot << 16);
    f4 t_small = splat(t_chunk) + (f4){0.0f, 1.0f, 2.0f, 3.0f};      // exact below 2^24, as are the + 4 steps
    // ALIGNED: the chunk's 16 noise values are 16 consecutive entries of the device's noise table (S2rRenderParams.
    // noise_tab, s2r_noise_table_kernel: entry x = the noise of hashed word x, the expression above evaluated once per
    // 16-bit x).  (seed ^ (o + j)) & 0xffff == ((seed ^ o) & 0xffff) + j for j < 16 when both low nibbles are zero, so a
    // lane reads 64 contiguous, 64-byte-aligned bytes: four 16-byte loads from a 256 KiB table that lives in the L2 (a
    // cohort of voices started together reads the same 64 bytes), against 40 instructions of stepping and scaling.
    // (Asking for them one chunk AHEAD was measured too: the median wave's chunk 924 -> 816 cycles, but the 16 registers it
    // holds across the chunk come back as spills in the moving-stage variants, the waves that set the kernel's length:
    // the launch no shorter, 0.0446 against 0.0439 ms.)
    f4 nzt[4];
    if (ALIGNED) {
        const float *np = p.noise_tab + ((r.seed_rot ^ o_chunk) & 0xffffu);
#pragma unroll
        for (int q = 0; q < 4; ++q) nzt[q] = *reinterpret_cast<const f4 *>(np + 4 * q);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f4 t;
        if (ALIGNED) {
            t = t_small;
            t_small = t_small + splat(4.0f);
            nz[q] = nzt[q];
        } else if (SMALL) {
            // stepping (one inline constant, one scalar literal) instead of 4 q + k per quad (a scalar move per literal)
            t = t_small;
            nz[q] = hash_noise4_low16(seed_pair, o01, o23);
            t_small = t_small + splat(4.0f);
            o01 = pk_add_u16(o01, 0x00040004u); o23 = pk_add_u16(o23, 0x00040004u);
        } else {
            const u4 ou = (u4)(o_chunk + 4u * q) + (u4){0u, 1u, 2u, 3u};
            t = __builtin_convertvector(ou, f4);
            nz[q] = hash_noise4(r.seed_rot, t) + splat(p.noise_level);
        }
        if (AFLAT) amp[q] = splat(ea_slope * (t_chunk - ea_base) + ea_y0);
        else amp[q] = splat(ea_slope) * pk_add4(t, splat(-ea_base)) + splat(ea_y0);     // t - base, packed
        if (SRC == 0) xq[q] = splat(xc0);
        if (SRC == 1 && FMV) {
            const f4 f_osc = pq[q] * splat(r.pitch);             // pow(2, mod * amount) * freq   process.rs:146-147,231-250
            pq[q] = splat(p.sr) / f_osc;                         // units.rs:32-42
            iq[q] = splat(1.0f) / pq[q];                         // oscillators.rs:378
        }
        if (SRC == 2) {
            const f4 mod = splat(em.s0) * (t - splat(em.s1)) + splat(em.s2);
            const f4 f_lpf = pow2_sleef_core4(mod * splat(p.amt_lpf)) * splat(p.lpf_freq);
            const f4 num = splat(-2.0f * 3.14159274101257324f) * f_lpf;
            const f4 arg = p.fast_div_sr ? div_const_nocheck4(num, p.sr, p.rcp_sr) : (num / splat(p.sr));
            xq[q] = expf4(arg, sT);
        }
    }
    // ---- the recurrences.  Per quad: the four phases (serial: add + fract per frame), then everything that is
    //      feed-forward in them on 4-vectors (one wave per SIMD issues one instruction per ~4.7 cycles whether it is
    //      packed or not — tools/ubench/issue_rates3.hip — so a v_pk_* carries two frames for the price of one), then
    //      the filter (serial), then the gain and the four stores.
    //      fmodf(off, period) (oscillators.rs:66,105,154; lookup.rs:195) is `off` itself here: off = RN(period * ph)
    //      with 0 <= ph <= 1 - 2^-24 is strictly below period for every normal period (period * 2^-24 is at least half
    //      an ulp of period, so the product never rounds back up to it; checked exhaustively over ten binades of
    //      periods in tests/test_oracle_known_answers.py::test_phased_offset_stays_below_the_period). ----
    const uint32_t livemask = live ? 0xffffffffu : 0u;
    f4 outs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // filters.rs:23.  A lane without a started voice must put +0.0 into the mix (synth.rs:178 skips
        // it): with a0 = 0 and last = 0 its y is 0*s + x*0 = +0 for every finite s and x >= 0, and
        // (+0) * (amp = +0) = +0 — no select per frame
        f4 a0 = splat(1.0f) - xq[q];
        if (SRC == 2 && !live) a0 = splat(0.0f);                 // SRC 1: such a lane reads the tables' x = 1, 1 - x = 0 entries
        // the quad's oscillator constants: the run's (k), or under FM each frame's period and 1 / period (from the
        // tables' pow2 plane) with the rest by exact scalings (make_osck)
        const f4 period = FMV ? pq[q] : splat(k.period), inv_period = FMV ? iq[q] : splat(k.inv_period);
        f4 ph;
        float phc = r.phase;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ph[j] = phc;
            // oscillators.rs:377-381; 0 <= nx < 2: fmodf(nx, 1) for nx >= 0 is its fractional part, which is exact;
            // v_fract_f32 returns min(nx - floor(nx), 0x1.fffffep-1) and the clamp cannot bind for nx < 2^23
            phc = __builtin_amdgcn_fractf(phc + inv_period[j]);
        }
        r.phase = phc;
        const f4 x = period * ph;                                // fma(period, ph, +0) with a product >= +0; == off % period
        f4 osc;
        if (OSC == S2R_OSC_SAW) {
            const f4 ka = FMV ? splat(-2.0f) * inv_period : splat(k.a);
            osc = vfma(ka, x, splat(1.0f));                      // oscillators.rs:107-112
        } else if (OSC == S2R_OSC_SQUARE) {
            // x < period / 2 ? 1 : -1 (oscillators.rs:68-76) without a compare: x - half is negative exactly when
            // x < half (a difference of two distinct floats is never zero), so its sign bit over the bits of 1.0 is
            // MINUS the sample
            const f4 ka = FMV ? period * splat(0.5f) : splat(k.a);
            const u4 d = (u4)(x - ka);
            osc = -(f4)((d & 0x80000000u) | 0x3f800000u);
        } else if (OSC == S2R_OSC_TRIANGLE) {
            const f4 ka = FMV ? period * splat(0.5f) : splat(k.a);
            const f4 kb = FMV ? splat(-4.0f) * inv_period : splat(k.b), kc = FMV ? splat(4.0f) * inv_period : splat(k.c);
            const f4 first = vfma(kb, x, splat(1.0f)), second = vfma(kc, x - ka, splat(-1.0f));     // oscillators.rs:156-172
            const u4 lt = (u4)((i4)(u4)(x - ka) >> 31);          // all ones where x < half
            osc = (f4)(((u4)first & lt) | ((u4)second & ~lt));
        } else {
            osc = sine_quad<FMV>(x, period, rcp_period, sSin);
        }
        const f4 sv = pk_add4(osc + splat(p.osc_gain), nz[q]);   // process.rs:342-345 (ADD), :358
        f4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (FILT == 0) {
                y[j] = __builtin_fmaf(a0[j], sv[j], xq[q][j] * r.last);          // filters.rs:23-33
                r.last = y[j];
            } else {
                FiltCoef cj = *fcoef;                            // the run's constants, or this frame's from the tables
                if (SRC == 1) { cj.alpha = xq[q][j]; cj.beta = bq[q][j]; cj.gamma = gq[q][j]; }
                y[j] = dsp_filter_apply(FILT, cj, sv[j], *f2);
            }
        }
        f4 out = y * amp[q];                                     // process.rs:373-376
        // (a dead lane's dsp_filters.rs state may run away on its made-up input: its output is masked to +0.0 here)
        if (FILT != 0) out = (f4)((u4)out & livemask);
        tile_store4(q, out);
        outs[q] = out;
    }
    // the per-voice rows (tests, s2r_fill_voices): one branch behind the chunk's basic block, not one per quad
    if (pv_dst) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f4u *>(pv_dst + 4 * q) = outs[q];
    }
}

}  // namespace
