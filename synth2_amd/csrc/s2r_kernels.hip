// s2r_kernels.hip — gfx950 kernels of the voice-render path.
//
// One voice per lane.  Per-voice recurrence state (phase, LPF history, frame offset) is
// loaded coalesced from the SoA arrays in HBM into registers, the frames of the fill are
// walked serially (phase accumulation and the one-pole LPF are recurrences over time), and
// the cross-voice mixdown is an LDS transpose-and-add per wave (16 voices in index order, the
// reference's own order) -> LDS across the waves of a workgroup -> one partial row per
// workgroup in HBM -> a second tiny kernel that adds the rows in a fixed order.
// No MFMA: this is a scalar-per-voice recurrence.
//
// Everything arithmetic follows the reference op for op (citations relative to
// /root/reference/components/s2_lib/src/); this file must be compiled with
// -ffp-contract=off and without fast-math.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "s2r_device.h"
#include "s2r_math.h"

namespace {

__constant__ uint64_t c_exp2f_table[S2R_EXP2F_N] = S2R_EXP2F_TABLE_INIT;

constexpr int kChunk = 16;         // the reference's x16 chunk (synth.rs:158, process.rs:25)
constexpr uint32_t kSuperMax = 256; // frames between two cross-wave combines: 256 (small workgroups) or 64
constexpr int kP = 4;              // frames whose closed-form work one lane carries at once (ILP)

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

// DPP move (the quad_perm exchanges between the L lanes of a voice)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
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
struct EnvRun {
    float slope, base, y0, thr;
};

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
    s.slope = s0 ? e.slope_att : s1 ? e.slope_dec : s3 ? e.slope_rel : 0.0f;
    s.base  = s1 ? e.A : s3 ? ro : 0.0f;
    s.y0    = s1 ? 1.0f : (s2 || s3) ? e.S : 0.0f;
    s.thr   = s0 ? e.A : s1 ? e.sus_off : s2 ? ro : s3 ? end : __builtin_inff();
    return s;
}

__device__ __forceinline__ float env_value(const EnvRun &s, float t) {
    return s.slope * (t - s.base) + s.y0;       // mul then add, separately rounded (simdtest.rs:247-261)
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
constexpr int S2R_OSC_ANY = 4;

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

template <int OSC, bool FM>
__device__ __forceinline__ FlatCache refresh_flat(const S2rRenderParams &p, const VoiceRegs &r, const EnvRun em,
                                                  const uint64_t *sT, FlatCache fc) {
    if (em.slope == 0.0f) {
        const float mod = em.y0;                                          // 0 * (t - base) + y0 == y0
        const float f_lpf = s2r_pow2_sleef_core(mod * p.amt_lpf) * p.lpf_freq;
        fc.xc = s2r_expf(p.fast_div_sr ? lpf_arg<true>(p, f_lpf) : lpf_arg<false>(p, f_lpf), sT);
        if (FM) fc.k = make_osck<OSC>(p.sr / (s2r_pow2_sleef_core(mod * p.amt_osc) * r.pitch));
    }
    return fc;
}

// frames oi .. oi+3 of one voice
template <int OSC, bool FM>
__device__ __forceinline__ void closed_form_x4(const S2rRenderParams &p, const VoiceRegs &r, EnvRun &ea, EnvRun &em,
                                               float &thr_min, FlatCache &fc, uint32_t oi, const uint64_t *sT,
                                               bool have_stream, f4 stream_xc, FrameCF4 &cf, OscK4 &k) {
    const u4 ou = (u4)(oi) + (u4){0u, 1u, 2u, 3u};               // offsets_x16: wrapping u32 add (process.rs:213-219)
    const f4 t = __builtin_convertvector(ou, f4);                // offsets as f32 (simdtest.rs:277-279, process.rs:348)
    // fast path first: the active stages' lines for all four frames
    f4 amp = splat(ea.slope) * (t - splat(ea.base)) + splat(ea.y0);           // process.rs:144
    f4 mod = splat(em.slope) * (t - splat(em.base)) + splat(em.y0);           // process.rs:145
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
                    thr_min = __builtin_fminf(ea.thr, em.thr);                            \
                }                                                                         \
                amp.C = env_value(ea, tj);                                                \
                mod.C = env_value(em, tj);                                                \
            }
            S2R_ENV_STEP(x) S2R_ENV_STEP(y) S2R_ENV_STEP(z) S2R_ENV_STEP(w)
#undef S2R_ENV_STEP
            fc = refresh_flat<OSC, FM>(p, r, em, sT, fc);
        }
    }
    cf.amp = amp;
    cf.nz = hash_noise4(r.seed_rot, t) + splat(p.noise_level);   // process.rs:347-356 (ADD)
    if (have_stream) {             // wave-uniform: this 64-voice group's coefficients were computed ahead
        cf.xc = stream_xc;
        return;
    }
    moving = moving || em.slope != 0.0f;
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
                            const uint64_t *sT, const float *sSin, Filt2 *f2 = nullptr) {
    const float t = (float)oi;
    const float rel = r.released ? (float)r.release_u : 4294967296.0f;   // envelopes.rs:35
    const float amp = adsr_scalar(p.amp, t, rel);
    const float mod = adsr_scalar(p.mod, t, rel);
    const float f_osc = s2r_pow2_libm(mod * p.amt_osc, sT) * r.pitch;    // process.rs:221-229
    const float f_lpf = s2r_pow2_libm(mod * p.amt_lpf, sT) * p.lpf_freq;
    const OscK k = make_osck_any<OSC>(p.osc_kind, p.sr / f_osc);
    const float ph = r.phase;
    const float off = __builtin_fmaf(k.period, ph, 0.0f);                // oscillators.rs:212
    const float osc = osc_value_any<OSC>(p.osc_kind, k, off, sSin);
    r.phase = s2r_fmod1(ph + k.inv_period);                              // oscillators.rs:377-381
    const float osc_s = osc * p.osc_gain;                                // process.rs:287 (MULTIPLY)
    const float noise_s = (hash_noise(r.seed_rot, t)) * p.noise_level;   // process.rs:292 (MULTIPLY)
    const float s = osc_s + noise_s;
    if (DSPF) return dsp_filter_step(p.lpf_kind, p.lpf_damping, p.sr, f_lpf, s, *f2) * amp;
    const float x = lpf_coeff<false, P>(p, f_lpf, sT);
    const float y = lpf_apply(x, s, r.last);
    return y * amp;
}

// quad_perm broadcast of sub-lane K inside each group of L consecutive lanes
template <int L, int K>
__device__ __forceinline__ float bcast_sub(float v) {
    constexpr int ctrl = (L == 4) ? (K * 0x55) : ((K) | (K << 2) | ((2 + K) << 4) | ((2 + K) << 6));
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false));
}
template <int L, int K>
__device__ __forceinline__ f4 bcast_sub4(f4 v) {
    f4 r;
    r.x = bcast_sub<L, K>(v.x); r.y = bcast_sub<L, K>(v.y); r.z = bcast_sub<L, K>(v.z); r.w = bcast_sub<L, K>(v.w);
    return r;
}

// ---------------------------------------------------------------------------------------
// Coefficient stream.  The render kernel is one wave per SIMD at 64 k voices, so its duration is
// that of its SLOWEST wave: a few 64-voice groups whose mod envelope is moving (full pow/exp
// chain every frame) hold back a chip of waves that only reuse a constant.  The LPF coefficient
// is closed-form in the frame offset, so for exactly those groups it is computed ahead of time,
// spread over the whole GPU ((group, 16-quad chunk) work items, four frames per lane as in the
// render kernel), into an HBM/L2-resident stream that the render kernel then just reads.
//   s2r_classify_kernel: one wave per 64-voice group; a group gets a slot iff some live voice's
//     mod envelope is not provably flat for the whole fill.
//   s2r_coeff_kernel:    persistent-style grid over (slot, chunk of 16 quads).
// If more groups move than the buffer holds (count > capacity) nobody uses the stream and the
// render kernel computes in-lane as before.  Not used under oscillator FM.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) s2r_classify_kernel(const S2rRenderParams p) {
    const uint32_t group = blockIdx.x, lane = threadIdx.x;
    const uint32_t vi = group * 64u + lane;
    if (group == 0 && lane == 0) p.coeff_count[p.coeff_parity ^ 1u] = 0u;       // ready for the next fill
    bool moving = false, timed = false;
    if (vi < p.n_voices) {
        // a voice that is re-triggered or released inside this fill: its group streams, and the coefficient
        // pass follows the voice's event chain chunk by chunk
        if (p.tev != nullptr) timed = p.voice_ev_head[vi] >= 0;
        const uint32_t flags = p.v.flags[vi];
        if (flags & S2R_VF_STARTED) {
            const uint32_t offset = p.v.offset[vi];
            const float rel_f = (flags & S2R_VF_RELEASED) ? (float)p.v.release[vi] : 4294967296.0f;
            const float ro_m = __builtin_fmaxf(rel_f, p.mod.sus_off), end_m = ro_m + p.mod.R;
            const EnvRun e0 = env_stage_at(p.mod, ro_m, end_m, (float)offset);
            const float t_last = (float)(offset + (p.frames - 1u));
            moving = !(e0.slope == 0.0f && t_last < e0.thr);
        }
    }
    const bool any = __ballot(moving || timed) != 0ull;
    if (lane == 0) {
        int32_t slot = -1;
        if (any) {
            const uint32_t s = atomicAdd(&p.coeff_count[p.coeff_parity], 1u);
            if (s < p.coeff_capacity) { slot = (int32_t)s; p.slot_group[s] = group; }
        }
        p.group_slot_w[group] = slot;
    }
}

template <bool FASTDIV, bool FM, bool DSPF>
__global__ void __launch_bounds__(256) s2r_coeff_kernel(const S2rRenderParams p) {
    __shared__ uint64_t sT[S2R_EXP2F_N];
    if (threadIdx.x < S2R_EXP2F_N) sT[threadIdx.x] = c_exp2f_table[threadIdx.x];
    __syncthreads();
    const uint32_t count = p.coeff_count[p.coeff_parity];
    if (count == 0u || count > p.coeff_capacity) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_quads = (p.frames & ~15u) / kP;
    constexpr uint32_t kQuadsPerItem = 4;
    const uint32_t chunks = (n_quads + kQuadsPerItem - 1) / kQuadsPerItem;
    const uint32_t n_items = count * chunks;
    for (uint32_t it = blockIdx.x * 4u + wave; it < n_items; it += gridDim.x * 4u) {
        const uint32_t slot = it / chunks, chunk = it % chunks;
        const uint32_t vi = p.slot_group[slot] * 64u + lane;
        const bool in_range = vi < p.n_voices;
        uint32_t flags = in_range ? p.v.flags[vi] : 0u;
        uint32_t offset = in_range ? p.v.offset[vi] : 0u;
        uint32_t release = (flags & S2R_VF_RELEASED) ? p.v.release[vi] : 0u;
        float pitch = (FM && in_range && (flags & S2R_VF_STARTED)) ? p.v.pitch[vi] : 440.0f;
        if (p.tev != nullptr && in_range) {
            // note events inside this fill (they land on 16-frame boundaries == item boundaries): the voice's state
            // for this item's frames is its state at the start of the fill with every event up to the item's first
            // frame applied, exactly as the render kernel applies them (apply_events_at)
            const uint32_t first = chunk * kQuadsPerItem * kP;
            int32_t ei = p.voice_ev_head[vi];
            while (ei >= 0) {
                const S2rTimedEvent e = p.tev[ei];
                if (e.frame > first) break;
                if (e.flags & S2R_EV_RESTART) {
                    flags = S2R_VF_STARTED | ((e.flags & S2R_EV_RELEASE) ? S2R_VF_RELEASED : 0u);
                    offset = 0u - e.frame; release = 0u; pitch = e.pitch;
                } else if ((e.flags & S2R_EV_RELEASE) && (flags & S2R_VF_STARTED) && !(flags & S2R_VF_RELEASED)) {
                    flags |= S2R_VF_RELEASED; release = offset + e.frame;
                }
                ei = e.next;
            }
        }
        const float rel_f = (flags & S2R_VF_RELEASED) ? (float)release : 4294967296.0f;
        const float ro_m = __builtin_fmaxf(rel_f, p.mod.sus_off), end_m = ro_m + p.mod.R;
        EnvRun em = env_stage_at(p.mod, ro_m, end_m, 0.0f);
        float thr = -__builtin_inff();
        constexpr uint32_t kBase = DSPF ? 3u : 1u, kVec = kBase + (FM ? 2u : 0u);   // per quad: xc | alpha, beta, gamma [, period, 1 / period]
        f4 *dst = (f4 *)p.coeff + ((size_t)slot * n_quads + (size_t)chunk * kQuadsPerItem) * kVec * 64u + lane;
        const uint32_t q1 = (chunk + 1) * kQuadsPerItem < n_quads ? kQuadsPerItem : n_quads - chunk * kQuadsPerItem;
        for (uint32_t q = 0; q < q1; ++q) {
            const uint32_t oi = offset + (chunk * kQuadsPerItem + q) * kP;
            const u4 ou = (u4)(oi) + (u4){0u, 1u, 2u, 3u};
            const f4 t = __builtin_convertvector(ou, f4);
            f4 mod = splat(em.slope) * (t - splat(em.base)) + splat(em.y0);
            if (!(t.w < thr)) {          // a stage boundary inside (or before) these frames: frame by frame
#define S2R_STEP(C) { if (!(t.C < thr)) { em = env_stage_at(p.mod, ro_m, end_m, t.C); thr = em.thr; } mod.C = env_value(em, t.C); }
                S2R_STEP(x) S2R_STEP(y) S2R_STEP(z) S2R_STEP(w)
#undef S2R_STEP
            }
            const f4 f_lpf = pow2_sleef_core4(mod * splat(p.amt_lpf)) * splat(p.lpf_freq);
            const f4 num = splat(-2.0f * 3.14159274101257324f) * f_lpf;
            if (DSPF) {                                          // dsp_filters.rs / SVF coefficients at the modulated cutoff
                f4 al, be, ga;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const FiltCoef fc = dsp_filter_coef(p.lpf_kind, p.lpf_damping, p.sr, f_lpf[c]);
                    al[c] = fc.alpha; be[c] = fc.beta; ga[c] = fc.gamma;
                }
                dst[(size_t)(q * kVec) * 64u] = al;
                dst[(size_t)(q * kVec + 1u) * 64u] = be;
                dst[(size_t)(q * kVec + 2u) * 64u] = ga;
            } else {
                const f4 arg = FASTDIV ? div_const_nocheck4(num, p.sr, p.rcp_sr) : (num / splat(p.sr));
                // a lane without a started voice (for these frames) gets x = 1: the render kernel's branch-free chunk
                // then has a0 = 1 - 1 = 0 and x * last = 1 * 0 for it without a select of its own (chunk_fast)
                const f4 xv = expf4(arg, sT);
                dst[(size_t)(q * kVec) * 64u] = (flags & S2R_VF_STARTED) ? xv : splat(1.0f);
            }
            if (FM) {
                const f4 f_osc = pow2_sleef_core4(mod * splat(p.amt_osc)) * splat(pitch);       // process.rs:146-147,231-250
                const f4 period = splat(p.sr) / f_osc;                                             // units.rs:32-42
                dst[(size_t)(q * kVec + kBase) * 64u] = period;
                dst[(size_t)(q * kVec + kBase + 1u) * 64u] = splat(1.0f) / period;                 // oscillators.rs:378
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Events + classification in one launch for the common case (few untimed events, one-pole patch
// without oscillator FM): one 64-lane workgroup per 64-voice group.  The fill's note events — the
// host folds them to at most one record per voice — ride in the kernel arguments; the group
// picks out the ones that hit it (what s2r_events_kernel does through mapped host memory),
// writes the touched voices back, and classifies itself on the post-event state exactly like
// s2r_classify_kernel.  The coefficient pass follows as its own launch: it spreads the FEW moving
// groups over the whole chip, which a per-group launch shape cannot (measured: 20 us in-group
// vs 7 us).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) s2r_prep_kernel(const S2rPrepParams a) {
    const S2rRenderParams &p = a.p;
    const uint32_t group = blockIdx.x, lane = threadIdx.x;
    const uint32_t vi = group * 64u + lane;
    const bool in_range = vi < p.n_voices;
    if (group == 0 && lane == 0) p.coeff_count[p.coeff_parity ^ 1u] = 0u;       // ready for the next fill
    uint32_t flags = in_range ? p.v.flags[vi] : 0u;
    uint32_t offset = in_range ? p.v.offset[vi] : 0u;
    uint32_t release = in_range ? p.v.release[vi] : 0u;
    // ---- the fill's note events (synth.rs:61-80): 64 records per step, one per lane; the few that
    //      hit this group are handed to their lane ----
    uint32_t my_flags = 0u, my_pitch = 0u;
    // every step's records are fetched before the first is looked at: one trip to the kernel-argument
    // segment for the wave instead of one per 64 events
    constexpr uint32_t kSteps = (S2R_PREP_MAX_EVENTS + 63u) / 64u;
    uint32_t evv[kSteps], evf[kSteps], evp[kSteps];
#pragma unroll
    for (uint32_t k = 0; k < kSteps; ++k) {
        const uint32_t i = k * 64u + lane;
        const bool have = i < a.n_events;
        evv[k] = have ? a.ev[3u * i] : 0xffffffffu;
        evf[k] = have ? a.ev[3u * i + 1u] : 0u;
        evp[k] = have ? a.ev[3u * i + 2u] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < kSteps; ++k) {
        const uint32_t ev_voice = evv[k], ev_flags = evf[k], ev_pitch = evp[k];
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
    const bool restart = (my_flags & S2R_EV_RESTART) != 0u;
    if (restart) {                                               // *voice = Voice { .. }, synth.rs:63-69
        flags = S2R_VF_STARTED | ((my_flags & S2R_EV_RELEASE) ? S2R_VF_RELEASED : 0u);
        offset = 0u; release = 0u;
    } else if ((my_flags & S2R_EV_RELEASE) && (flags & S2R_VF_STARTED) && !(flags & S2R_VF_RELEASED)) {   // synth.rs:74-75
        release = offset; flags |= S2R_VF_RELEASED;
    }
    if (my_flags != 0u) {
        if (restart) {
            p.v.pitch[vi] = s2r_u2f(my_pitch);
            p.v.offset[vi] = 0u;
            p.v.phase[vi] = 0.0f;
            p.v.lpf_last[vi] = 0.0f;
            p.v.fx1[vi] = 0.0f; p.v.fx2[vi] = 0.0f; p.v.fy1[vi] = 0.0f; p.v.fy2[vi] = 0.0f;
            p.v.seed[vi] = 0u;                                   // this path carries no seed overrides
            p.v.program[vi] = my_flags >> S2R_EV_PROGRAM_SHIFT;
        }
        p.v.release[vi] = release;
        p.v.flags[vi] = flags;
    }
    // ---- does any mod envelope of the group move during this fill? (as s2r_classify_kernel) ----
    bool moving = false;
    if (flags & S2R_VF_STARTED) {
        const float rel_f = (flags & S2R_VF_RELEASED) ? (float)release : 4294967296.0f;
        const float ro_m = __builtin_fmaxf(rel_f, p.mod.sus_off), end_m = ro_m + p.mod.R;
        const EnvRun e0 = env_stage_at(p.mod, ro_m, end_m, (float)offset);
        const float t_last = (float)(offset + (p.frames - 1u));
        moving = !(e0.slope == 0.0f && t_last < e0.thr);
    }
    const bool any = __ballot(moving) != 0ull;
    if (lane == 0) {
        int32_t slot = -1;
        if (any) {
            const uint32_t sidx = atomicAdd(&p.coeff_count[p.coeff_parity], 1u);
            if (sidx < p.coeff_capacity) { slot = (int32_t)sidx; p.slot_group[sidx] = group; }
        }
        p.group_slot_w[group] = slot;
    }
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
//   SRC: 0 = every voice flat (constant coefficient), 1 = coefficient stream, 2 = compute in-lane.
// ---------------------------------------------------------------------------------------
//   FILT != 0 (general kernel, flat stages only): the layer's filter is dsp_filters.rs' / the SVF with the
//   constant coefficients `fcoef`, state in *f2, instead of the one-pole.
//   FMV (with SRC == 1, oscillator FM): the stream carries three vectors per quad — the LPF coefficient, the
//   oscillator period sr / f_osc and its reciprocal — and the period constants are per-frame values.
//   SMALL (one-pole kernel): every lane's offsets of the chunk are below 2^24, so the f32 offsets are exact sums and
//   the noise hash works on their low 16 bits (hash_noise4_low16), and the patch's noise level is 0.0, so adding it
//   (process.rs:353-356) changes nothing: the noise value itself is never +-0 (v / 65535 == 0.5 has no integer
//   solution), and n + 0.0 == n for every other n.
//   AFLAT (with SMALL): every started voice of the wave sits in an amplitude stage of slope +-0 (sustain, end) for
//   the whole run, so slope * (t - base) + y0 is (+-0) + y0 with the product's sign fixed by the slope's (t >= base
//   inside a stage): one evaluation per chunk, at its first frame, is every frame's value bit for bit.
template <int OSC, int SRC, int FILT = 0, bool FMV = false, bool SMALL = false, bool AFLAT = false>
__device__ __forceinline__ void chunk_fast(const S2rRenderParams &p, VoiceRegs &r, const EnvRun &ea, const EnvRun &em,
                                           const FlatCache &fc, const OscK &k, uint32_t o_chunk, const f4 *stream_q,
                                           const uint64_t *sT, const float *sSin, bool live, float *tile_col,
                                           uint32_t tile_stride, float *pv_dst,
                                           const FiltCoef *fcoef = nullptr, Filt2 *f2 = nullptr) {
    // stream layout per quad: the filter's coefficients (one vector for the one-pole, alpha / beta / gamma for the
    // others), then under FM the period and its reciprocal
    f4 xq[4], bq[4], gq[4], pq[4], iq[4];
    constexpr uint32_t kBase = FILT != 0 ? 3u : 1u, kVec = kBase + (FMV ? 2u : 0u);
    if (SRC == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {                            // coalesced 16-byte loads, used last
            xq[q] = stream_q[(size_t)(q * kVec) * 64u];
            if (FILT != 0) { bq[q] = stream_q[(size_t)(q * kVec + 1u) * 64u]; gq[q] = stream_q[(size_t)(q * kVec + 2u) * 64u]; }
            if (FMV) { pq[q] = stream_q[(size_t)(q * kVec + kBase) * 64u]; iq[q] = stream_q[(size_t)(q * kVec + kBase + 1u) * 64u]; }
        }
    }
    const double rcp_period = (OSC == S2R_OSC_SINE && !FMV) ? s2r_rcp_f64(k.period) : 0.0;   // constant over the run: hoisted by the compiler
    // a lane without a started voice: see a0 / ampq below.  Its filter history must be 0 for that (frames
    // of the general path, which selects per frame instead, leave a running value in it)
    // — and its amplitude line is 0 * (t - 0) + 0 = +0 and, where the coefficient is the run's constant, x = 1
    // (a0 = 1 - 1 = 0): selected once per chunk instead of per quad
    // (selects, not a branch: control flow here would split the chunk's basic block)
    r.last = live ? r.last : 0.0f;
    const float ea_slope = live ? ea.slope : 0.0f, ea_base = live ? ea.base : 0.0f, ea_y0 = live ? ea.y0 : 0.0f;
    const float xc0 = live ? fc.xc : 1.0f;
    f4 amp[4], nz[4];
    const float t_chunk = (float)o_chunk;
    // SMALL: the low 16 bits of the chunk's first four offsets as two pairs, and of rotl(seed, 5) twice
    const uint32_t o_lo = o_chunk & 0xffffu;
    uint32_t o01 = pk_add_u16(o_lo | (o_lo << 16), 0x00010000u), o23 = pk_add_u16(o01, 0x00020002u);
    const uint32_t seed_pair = (r.seed_rot & 0xffffu) | (r.seed_rot << 16);
    f4 t_small = splat(t_chunk) + (f4){0.0f, 1.0f, 2.0f, 3.0f};      // exact below 2^24, as are the + 4 steps
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f4 t;
        if (SMALL) {
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
        else amp[q] = splat(ea_slope) * (t - splat(ea_base)) + splat(ea_y0);
        if (SRC == 0) xq[q] = splat(xc0);
        if (SRC == 2) {
            const f4 mod = splat(em.slope) * (t - splat(em.base)) + splat(em.y0);
            const f4 f_lpf = pow2_sleef_core4(mod * splat(p.amt_lpf)) * splat(p.lpf_freq);
            const f4 num = splat(-2.0f * 3.14159274101257324f) * f_lpf;
            const f4 arg = p.fast_div_sr ? div_const_nocheck4(num, p.sr, p.rcp_sr) : (num / splat(p.sr));
            xq[q] = expf4(arg, sT);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // filters.rs:23.  A lane without a started voice must put +0.0 into the mix (synth.rs:178 skips
        // it): with a0 = 0 and last = 0 its y is 0*s + x*0 = +0 for every finite s and x >= 0, and
        // (+0) * (amp = +0) = +0 — no select per frame
        f4 a0 = splat(1.0f) - xq[q];
        const f4 ampq = amp[q];
        if (SRC == 2 && !live) a0 = splat(0.0f);                 // SRC 1: the stream holds x = 1 for such a lane (s2r_coeff_kernel)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // the frame's oscillator constants: the run's (k), or under FM the streamed period and 1/period with
            // the rest by exact scalings (make_osck)
            const float period = FMV ? pq[q][j] : k.period, inv_period = FMV ? iq[q][j] : k.inv_period;
            const float ka = !FMV ? k.a : (OSC == S2R_OSC_SAW ? -2.0f * inv_period : period * 0.5f);
            const float kb = !FMV ? k.b : -4.0f * inv_period, kc = !FMV ? k.c : 4.0f * inv_period;
            const float ph = r.phase;
            const float nx = ph + inv_period;                    // oscillators.rs:377-381; 0 <= nx < 2
            // fmodf(nx, 1) for nx >= 0 is its fractional part, which is exact; v_fract_f32 returns
            // min(nx - floor(nx), 0x1.fffffep-1) and the clamp cannot bind for nx < 2^23
            r.phase = __builtin_amdgcn_fractf(nx);
            const float off = period * ph;                       // fma(period, ph, +0) with a product >= +0
            // fmodf(off, period) on [0, period] (off == period -> 0): both are non-negative floats, so
            // bits(off) - bits(period) is negative exactly when off < period (the compiler turns this into an
            // integer compare + select on an SGPR pair, cheaper here than the float compare through VCC)
            const int32_t keep = ((int32_t)s2r_f2u(off) - (int32_t)s2r_f2u(period)) >> 31;
            const float x = s2r_u2f(s2r_f2u(off) & (uint32_t)keep);
            float osc;
            if (OSC == S2R_OSC_SAW) osc = __builtin_fmaf(ka, x, 1.0f);
            else if (OSC == S2R_OSC_SQUARE) osc = x < ka ? 1.0f : -1.0f;
            else if (OSC == S2R_OSC_TRIANGLE) {
                const float first = __builtin_fmaf(kb, x, 1.0f), second = __builtin_fmaf(kc, x - ka, -1.0f);
                osc = x < ka ? first : second;
            } else {
                // x * 1024 / period: through the run's double reciprocal, exactly (s2r_math.h); a true division when
                // the period changes every frame
                const float tv = FMV ? x * 1024.0f / period : s2r_div_by_rcp64(x * 1024.0f, rcp_period);
                const uint32_t i1 = s2r_f32_as_u32(tv), i2 = (i1 + 1u) & 1023u;
                const float2 pr = sin_pair(sSin, i2);
                const float s1 = i1 < 1024u ? pr.x : 0.0f, s2 = pr.y;
                osc = __builtin_fmaf((s2 - s1) / 1.0f, tv - (float)i1, s1);
            }
            const float s = (osc + p.osc_gain) + nz[q][j];
            float out;
            if (FILT == 0) {
                const float y = __builtin_fmaf(a0[j], s, xq[q][j] * r.last);
                r.last = y;
                out = y * ampq[j];
            } else {
                // (a dead lane's filter state may run away on its made-up input: select per frame here)
                FiltCoef cj = *fcoef;                            // the run's constants, or this frame's from the stream
                if (SRC == 1) { cj.alpha = xq[q][j]; cj.beta = bq[q][j]; cj.gamma = gq[q][j]; }
                const float y = dsp_filter_apply(FILT, cj, s, *f2);
                out = live ? y * amp[q][j] : 0.0f;
            }
            tile_col[(4 * q + j) * tile_stride] = out;
            if (pv_dst) pv_dst[4 * q + j] = out;
        }
    }
}

// ---------------------------------------------------------------------------------------
// render kernel.
//   * one voice per lane-group of L lanes (L = 1, 2 or 4);
//   * the x16 path walks the fill in groups of 4*L frames: each lane evaluates the closed-form
//     part of FOUR consecutive frames on 4-wide vectors (ILP), the L lanes of a voice take
//     consecutive quadruples, exchange results with quad_perm DPP moves, and every lane then
//     runs the short recurrence (phase, oscillator, LPF) for all 4*L frames;
//   * L > 1 multiplies the number of waves for the same voice count (64k voices are only one
//     wave per SIMD at L = 1) at the price of L x the recurrence work (~10% of a frame).
//   grid = ceil(n_voices / block_voices), blockDim.x = block_voices * L.
// ---------------------------------------------------------------------------------------
template <int OSC, bool FM, int MODE, int L, int MAXT>
__global__ void __launch_bounds__(MAXT) s2r_render_kernel(const S2rRenderParams p) {
    constexpr bool PV = MODE == 1;       // also store every voice's frames (mix-disabled debug/parity output)
    constexpr bool TEV = MODE == 2;      // note events that take effect inside the fill
    const uint32_t kSuper = p.super_frames;                      // 64 or 256, wave-uniform
    __shared__ uint64_t sT[S2R_EXP2F_N];
    __shared__ __attribute__((aligned(8))) float sSin[OSC == S2R_OSC_SINE ? 2048 : 2];
    extern __shared__ float s_dyn[];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6, n_waves = blockDim.x >> 6;
    // mixdown staging: per wave a [16 frames][VW voices + 1] tile that is written
    // voice-per-lane and read frame-per-lane, and per block the 16-voice group sums
    // sW[2][n_groups][64 frames] for the cross-wave combine
    constexpr uint32_t VW = 64 / L;                              // voices per wave
    constexpr uint32_t GW = VW / 16;                             // 16-voice groups per wave
    const uint32_t n_groups = n_waves * GW;
    float *const sW = s_dyn;                                     // [2][n_groups][kSuper]
    // two tiles per wave: the branch-free runs fill one while the previous chunk's is being added up
    constexpr uint32_t kTile = kChunk * (VW + 1);
    float *const tile = s_dyn + 2 * n_groups * kSuper + wave * (2 * kTile);
    const uint32_t sub = tid & (L - 1);                          // which quadruple of the group this lane prepares
    const uint32_t block_voices = blockDim.x / L;
    const uint32_t vi = blockIdx.x * block_voices + tid / L;

    if (tid < S2R_EXP2F_N) sT[tid] = c_exp2f_table[tid];
    if (OSC == S2R_OSC_SINE)
        for (uint32_t i = tid; i < 1024u; i += blockDim.x) { sSin[2u * i] = p.sin_table[(i + 1023u) & 1023u]; sSin[2u * i + 1u] = p.sin_table[i]; }

    // ---- load per-voice state (coalesced SoA reads; the L lanes of a voice read the same words) ----
    const bool in_range = vi < p.n_voices;
    const uint32_t flags = in_range ? p.v.flags[vi] : 0u;
    bool live = (flags & S2R_VF_STARTED) != 0u;                // synth.rs:178
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

    // mod_env_to_osc_freq == 0: pow(2, mod*0) == 1 exactly, so freq == 1.0 * pitch and the
    // period (and everything derived from it by correctly rounded divisions) is constant.
    OscK k_const = make_osck<OSC>(p.sr / (1.0f * r.pitch));

    // timed events: this voice's chain for the fill (taken over and cleared for the next fill)
    int32_t ev_idx = -1;
    uint32_t ev_frame = 0xffffffffu;
    bool ev_dirty = false, ev_restart = false;
    uint32_t seed_now = seed, program_now = 0u;
    if (TEV && in_range) {
        ev_idx = p.voice_ev_head[vi];
        if (ev_idx >= 0) {
            ev_frame = p.tev[ev_idx].frame;
            if (sub == 0) p.voice_ev_head[vi] = -1;
        }
    }

    // envelope stage registers: the cascade for the fill's first frame, run here once so that the first
    // chunk can already take the branch-free path (a timed event or a stage boundary resets thr_min to
    // -inf, which sends the next frame through the cascade again)
    EnvRun ea = env_stage_at(p.amp, r.ro_a, r.end_a, (float)r.offset);
    EnvRun em = env_stage_at(p.mod, r.ro_m, r.end_m, (float)r.offset);
    float thr_min = __builtin_fminf(ea.thr, em.thr);
    FlatCache fc;
    fc.xc = 0.0f; fc.k = k_const;

    __syncthreads();
    fc = refresh_flat<OSC, FM>(p, r, em, sT, fc);                // needs the exp2 table in LDS

    const bool wave_live = __ballot(live || ev_idx >= 0) != 0ull;
    const uint32_t x16_frames = p.frames & ~(uint32_t)(kChunk - 1);      // frames in full 16-chunks
    // preconditions of the branch-free chunk that hold for the whole fill once they hold at its start
    // (phase' = fmodf(phase + 1/period, 1) stays in [0,1) for a positive period)
    // Under oscillator FM the period constants are per-frame values — except while every lane's mod envelope
    // is flat, when they are the ones cached at stage entry (fc.k): that case is checked per run.
    const bool fast_ok = L == 1 && __ballot(!(r.phase >= 0.0f && r.phase < 1.0f)) == 0ull &&
        (FM || __ballot(!(k_const.period > 0.0f && k_const.period < __builtin_inff())) == 0ull);
    // coefficient stream for this wave's 64-voice group, if one was prepared (wave-uniform)
    int32_t slot = -1;
    if (p.use_coeff && p.coeff_count[p.coeff_parity] <= p.coeff_capacity)
        slot = p.group_slot[__builtin_amdgcn_readfirstlane((blockIdx.x * block_voices + tid / L) / 64u)];
    const bool have_stream = slot >= 0;
    // under oscillator FM the stream holds three vectors per quad and only the branch-free chunks read it; the
    // general path then computes in-lane as if there were none
    constexpr uint32_t kVec = FM ? 3u : 1u;
    const bool have_stream_gp = have_stream && !FM;
    const f4 *stream = (const f4 *)p.coeff + ((size_t)(have_stream ? slot : 0) * (x16_frames / kP)) * kVec * 64u + ((tid / L) & 63u);
    f4 xc_next = splat(0.0f);
    if (have_stream_gp && x16_frames) xc_next = stream[(size_t)sub * 64u];
    const size_t pv_base = (size_t)vi * p.frames;
    const bool pv_lane = in_range && sub == 0;
    float *bp = p.block_partials + (size_t)blockIdx.x * p.frames_stride;
    uint32_t buf = 0;
    constexpr uint32_t G = kP * L;                                       // frames per group

    const uint32_t col = lane / L;                                       // this voice's column in the wave's tile
    // after each 16-frame chunk lane j adds, for frame (j & 15), the 16 voices of group (j >> 4) in
    // index order (the reference's own order within the group, synth.rs:177-195) and files the
    // group sum for the cross-wave combine
    auto reduce_chunk = [&](uint32_t f_base, uint32_t n_frames) {
        const uint32_t f = lane & 15u, grp = lane >> 4;
        if (grp < GW && f < n_frames) {
            const float *src = tile + f * (VW + 1) + grp * 16u;
            float acc = src[0];
#pragma unroll
            for (int k = 1; k < 16; ++k) acc += src[k];
            sW[(buf * n_groups + wave * GW + grp) * kSuper + f_base + f] = acc;
        }
    };

    // note events that land on the 16-frame boundary `fpos` (synth.rs:61-80 applied between two
    // 16-frame sample() calls, main.rs:140-142): restart or release this lane's voice
    auto apply_events_at = [&](uint32_t fpos) {
        if (__ballot(ev_frame == fpos) == 0ull) return;
        while (ev_frame == fpos) {
            const S2rTimedEvent e = p.tev[ev_idx];
            if (e.flags & S2R_EV_RESTART) {                  // *voice = Voice { .. }, synth.rs:63-69
                r.pitch = e.pitch;
                r.offset = 0u - fpos;                        // current_frame_offset == 0 at frame fpos
                r.release_u = 0u;
                r.released = (e.flags & S2R_EV_RELEASE) != 0u;
                r.phase = 0.0f; r.last = 0.0f;
                seed_now = e.seed;
                r.seed_rot = (e.seed << 5) | (e.seed >> 27);
                program_now = e.program;
                live = true; ev_restart = true;
                k_const = make_osck<OSC>(p.sr / (1.0f * r.pitch));
            } else if ((e.flags & S2R_EV_RELEASE) && live && !r.released) {   // synth.rs:74-75
                r.released = true;
                r.release_u = r.offset + fpos;
            }
            ev_dirty = true;
            ev_idx = e.next;
            ev_frame = ev_idx >= 0 ? p.tev[ev_idx].frame : 0xffffffffu;
        }
        const float rf = r.released ? (float)r.release_u : 4294967296.0f;
        r.ro_a = __builtin_fmaxf(rf, p.amp.sus_off); r.end_a = r.ro_a + p.amp.R;
        r.ro_m = __builtin_fmaxf(rf, p.mod.sus_off); r.end_m = r.ro_m + p.mod.R;
        // the envelope cascade for the boundary's frame, so that the chunk can take the branch-free path at once
        const float t0 = (float)(r.offset + fpos);
        ea = env_stage_at(p.amp, r.ro_a, r.end_a, t0);
        em = env_stage_at(p.mod, r.ro_m, r.end_m, t0);
        thr_min = __builtin_fminf(ea.thr, em.thr);
        fc = refresh_flat<OSC, FM>(p, r, em, sT, fc);
    };

    // Frames are walked in super-chunks of 64 (one barrier and one cross-wave combine each), each
    // made of 16-frame chunks (one wave-local transpose-and-add each).
    for (uint32_t sc0 = 0; sc0 < p.frames; sc0 += kSuper) {
        const uint32_t n_sc = (p.frames - sc0 < kSuper) ? (p.frames - sc0) : kSuper;
        const uint32_t n_x16 = (x16_frames > sc0) ? ((x16_frames - sc0 < n_sc) ? (x16_frames - sc0) : n_sc) : 0u;
        if (wave_live) {
            for (uint32_t c16 = 0; c16 < n_x16; c16 += kChunk) {
                if (TEV) apply_events_at(sc0 + c16);
                const uint32_t o_chunk = r.offset + sc0 + c16;
                if (fast_ok) {
                    // How many 16-frame chunks can run branch-free from here?  Offsets only grow and the
                    // active stages' thresholds stay put on this path, so if the LAST frame of a run is below
                    // every lane's next threshold, every earlier frame is too: one wave-uniform decision per
                    // run (up to the rest of the super-chunk) instead of several per chunk — with one wave per
                    // SIMD every VALU->scalar decision and taken branch is a bubble nothing else fills.
                    const uint32_t left = (n_x16 - c16) / kChunk;
                    auto clear_for = [&](uint32_t n) {
                        return __ballot(!((float)(o_chunk + n * kChunk - 1u) < thr_min)) == 0ull;
                    };
                    uint32_t run = 0;
                    uint32_t most = left;                        // under timed events: up to the wave's next one
                    if (TEV) {
                        uint32_t nxt = ev_frame;                 // 0xffffffff: none
#pragma unroll
                        for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)nxt, sh); nxt = o < nxt ? o : nxt; }
                        nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt);
                        const uint32_t until = nxt == 0xffffffffu ? left : (nxt - (sc0 + c16)) / kChunk;
                        most = until < left ? until : left;      // (nxt > sc0 + c16: events at this boundary were just applied)
                    }
                    if (most == 0u) run = 0u;
                    else if (clear_for(most)) run = most;
                    else if (most > 4u && clear_for(4u)) run = 4u;
                    else if (clear_for(1u)) run = 1u;
                    if (FM && run && !have_stream) {
                        const bool fm_flat = !p.no_flat_shortcut &&
                            __ballot(em.slope != 0.0f || !(fc.k.period > 0.0f && fc.k.period < __builtin_inff())) == 0ull;
                        if (!fm_flat) run = 0;
                    }
                    if (run) {
                        // Software pipeline: chunk i lands in tile (i & 1); the 16 loads and the serial adds of
                        // chunk i-1's transpose-and-add are issued BEFORE chunk i's arithmetic and its group sum is
                        // stored after it, so their latency overlaps work instead of idling the wave's only SIMD.
                        run = (uint32_t)__builtin_amdgcn_readfirstlane((int)run);
                        const uint32_t rf = lane & 15u, rgrp = lane >> 4;
                        const bool r_on = L == 1 || rgrp < GW;
                        auto tile_sum = [&](uint32_t b) {
                            const float *src = tile + b * kTile + rf * (VW + 1) + (r_on ? rgrp : 0u) * 16u;
                            float acc = src[0];
#pragma unroll
                            for (int q = 1; q < 16; ++q) acc += src[q];
                            return acc;
                        };
                        float *const sw_row = sW + (buf * n_groups + wave * GW + (r_on ? rgrp : 0u)) * kSuper + rf;
                        auto run_chunks = [&](auto src_tag, auto small_tag, auto aflat_tag) {
                            constexpr int SRC = decltype(src_tag)::value;
                            constexpr bool SMALL = decltype(small_tag)::value;
                            constexpr bool AFLAT = decltype(aflat_tag)::value;
                            for (uint32_t i = 0; i < run; ++i) {
                                const uint32_t f0 = c16 + i * kChunk;            // frame inside the super-chunk
                                // unconditional, so that the loads and the serial adds sit in the chunk's basic block and
                                // the scheduler spreads them over it (before the run's first chunk the other tile holds
                                // stale or no data: read, never used)
                                const float prev = tile_sum((i - 1u) & 1u);
                                float *pvd = (PV && pv_lane) ? p.per_voice + pv_base + sc0 + f0 : nullptr;
                                const f4 *sq = stream + (size_t)((sc0 + f0) / kP) * kVec * 64u;
                                chunk_fast<OSC, SRC, 0, (FM && SRC == 1), SMALL, AFLAT>(p, r, ea, em, fc, FM ? fc.k : k_const, o_chunk + i * kChunk, sq, sT, sSin, live,
                                                                                 tile + (i & 1u) * kTile + col, VW + 1, pvd);
                                if (i && r_on) sw_row[f0 - kChunk] = prev;
                            }
                            const float last = tile_sum((run - 1u) & 1u);
                            if (r_on) sw_row[c16 + (run - 1u) * kChunk] = last;
                        };
                        // every offset of the run below 2^24 on every lane (349 s at 48 kHz; voices are never freed,
                        // synth.rs:196-199, so older ones exist) and a patch without noise (the default one): the cheaper
                        // offset and noise arithmetic of SMALL
                        const bool small = p.noise_level == 0.0f && __ballot(o_chunk + run * kChunk > (1u << 24)) == 0ull;
                        const bool flat = FM || (!p.no_flat_shortcut && __ballot(live && em.slope != 0.0f) == 0ull);   // FM: checked above
                        // ... and every started voice's amplitude envelope in a zero-slope stage (no threshold falls inside
                        // the run, so the stage holds): the amplitude is one value per chunk (AFLAT)
                        const bool aflat = small && __ballot(live && ea.slope != 0.0f) == 0ull;
                        using T = std::true_type; using F = std::false_type;
                        if (have_stream) {
                            if (aflat) run_chunks(std::integral_constant<int, 1>{}, T{}, T{});
                            else if (small) run_chunks(std::integral_constant<int, 1>{}, T{}, F{});
                            else run_chunks(std::integral_constant<int, 1>{}, F{}, F{});
                        } else if (flat) {
                            if (aflat) run_chunks(std::integral_constant<int, 0>{}, T{}, T{});
                            else if (small) run_chunks(std::integral_constant<int, 0>{}, T{}, F{});
                            else run_chunks(std::integral_constant<int, 0>{}, F{}, F{});
                        } else if (small) run_chunks(std::integral_constant<int, 2>{}, T{}, F{});
                        else run_chunks(std::integral_constant<int, 2>{}, F{}, F{});
                        c16 += (run - 1u) * kChunk;
                        if (have_stream_gp) {                    // keep the general path's one-ahead prefetch coherent
                            const uint32_t qn = (sc0 + c16 + kChunk) / kP + sub;
                            if (qn < x16_frames / kP) xc_next = stream[(size_t)qn * 64u];
                        }
                        continue;
                    }
                }
                for (uint32_t g = c16; g < c16 + kChunk; g += G) {
                    // closed-form work of frames sc0+g+4*sub .. +3 on this lane
                    FrameCF4 cf; OscK4 kf;
                    const f4 xc_now = xc_next;
                    if (have_stream_gp) {    // prefetch the next group's quadruple while this one is consumed
                        const uint32_t qn = (sc0 + g + G) / kP + sub;
                        if (qn < x16_frames / kP) xc_next = stream[(size_t)qn * 64u];
                    }
                    closed_form_x4<OSC, FM>(p, r, ea, em, thr_min, fc, r.offset + sc0 + g + kP * sub, sT,
                                                     have_stream_gp, xc_now, cf, kf);
                    // recurrence for the 4*L frames of the group, every lane of the voice alike
#define S2R_QUAD(Q)                                                                              \
                    if constexpr (Q < L) {                                                       \
                        FrameCF4 c4; OscK4 k4;                                                   \
                        if constexpr (L == 1) { c4 = cf; if (FM) k4 = kf; }                      \
                        else {                                                                   \
                            c4.amp = bcast_sub4<L, Q>(cf.amp); c4.xc = bcast_sub4<L, Q>(cf.xc);  \
                            c4.nz = bcast_sub4<L, Q>(cf.nz);                                     \
                            if (FM) {                                                            \
                                k4.period = bcast_sub4<L, Q>(kf.period);                         \
                                k4.inv_period = bcast_sub4<L, Q>(kf.inv_period);                 \
                                if (OSC != S2R_OSC_SINE) k4.a = bcast_sub4<L, Q>(kf.a);          \
                                if (OSC == S2R_OSC_TRIANGLE) { k4.b = bcast_sub4<L, Q>(kf.b); k4.c = bcast_sub4<L, Q>(kf.c); } \
                            }                                                                    \
                        }                                                                        \
                        f4 out4;                                                                 \
                        _Pragma("unroll")                                                        \
                        for (int j = 0; j < kP; ++j) {                                           \
                            FrameCF c1; c1.amp = c4.amp[j]; c1.xc = c4.xc[j]; c1.nz = c4.nz[j];  \
                            OscK k1 = k_const;                                                   \
                            if (FM) { k1.period = k4.period[j]; k1.inv_period = k4.inv_period[j]; \
                                      k1.a = k4.a[j]; k1.b = k4.b[j]; k1.c = k4.c[j]; }          \
                            const float o = recur_x16<OSC>(p, r, c1, k1, sSin);                  \
                            out4[j] = live ? o : 0.0f;                                           \
                        }                                                                        \
                        if (PV) { if (pv_lane) {                                        \
                            float *dst = p.per_voice + pv_base + sc0 + g + kP * Q;               \
                            dst[0] = out4.x; dst[1] = out4.y; dst[2] = out4.z; dst[3] = out4.w; } } \
                        float *trow = tile + ((g + kP * Q) & 15u) * (VW + 1) + col;              \
                        trow[0 * (VW + 1)] = out4.x; trow[1 * (VW + 1)] = out4.y;                \
                        trow[2 * (VW + 1)] = out4.z; trow[3 * (VW + 1)] = out4.w;                \
                    }
                    S2R_QUAD(0) S2R_QUAD(1) S2R_QUAD(2) S2R_QUAD(3)
#undef S2R_QUAD
                }
                reduce_chunk(c16, kChunk);
            }
            if (n_x16 < n_sc) {                                          // scalar tail (< 16 frames, last super-chunk)
                if (TEV) apply_events_at(sc0 + n_x16);
                for (uint32_t i = n_x16; i < n_sc; ++i) {
                    float out = frame_sisd<OSC>(p, r, r.offset + sc0 + i, sT, sSin);
                    out = live ? out : 0.0f;
                    if (PV) { if (pv_lane) p.per_voice[pv_base + sc0 + i] = out; }
                    tile[(i - n_x16) * (VW + 1) + col] = out;
                }
                reduce_chunk(n_x16, n_sc - n_x16);
            }
        } else {
            for (uint32_t i = lane; i < GW * kSuper; i += 64u)
                sW[(buf * n_groups + wave * GW + i / kSuper) * kSuper + (i % kSuper)] = 0.0f;
            if (PV) { if (pv_lane) for (uint32_t i = 0; i < n_sc; ++i) p.per_voice[pv_base + sc0 + i] = 0.0f; }
        }
        __syncthreads();
        for (uint32_t f = tid; f < n_sc; f += blockDim.x) {
            // the block's 16-voice group sums, in group (= voice index) order
            float acc = sW[(buf * n_groups + 0) * kSuper + f];
            for (uint32_t gq = 1; gq < n_groups; ++gq) acc += sW[(buf * n_groups + gq) * kSuper + f];
            bp[sc0 + f] = acc;
            if (p.direct_out) {                                  // one workgroup: this IS the mix
                const float total = 0.0f + acc;                  // accum = splat(0.0), synth.rs:176
                if (p.direct_stereo) { p.direct_out[2u * (sc0 + f)] = total; p.direct_out[2u * (sc0 + f) + 1u] = total; }
                else p.direct_out[sc0 + f] = total;
            }
        }
        buf ^= 1u;
    }

    // ---- write back the recurrence state ----
    if (live && sub == 0) {
        const uint32_t o = r.offset;
        // (a voice restarted at frame f carries offset = -f here, so the sum is frames - f)
        p.v.offset[vi] = (!ev_dirty && o > 0xffffffffu - p.frames) ? 0xffffffffu : o + p.frames;   // synth.rs:197
        p.v.phase[vi] = r.phase;
        p.v.lpf_last[vi] = r.last;
        if (ev_dirty) {
            p.v.pitch[vi] = r.pitch;
            p.v.release[vi] = r.release_u;
            p.v.flags[vi] = S2R_VF_STARTED | (r.released ? S2R_VF_RELEASED : 0u);
            p.v.seed[vi] = seed_now;
            if (ev_restart) {                                    // st::Layer::default(), and the program it was started with
                p.v.fx1[vi] = 0.0f; p.v.fx2[vi] = 0.0f; p.v.fy1[vi] = 0.0f; p.v.fy2[vi] = 0.0f;
                p.v.program[vi] = program_now;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// general render kernel: everything the tuned one-pole kernel above does not cover.
//   BANK = false: one patch (kernel arguments), filter = one of dsp_filters.rs (lpf.kind != onepole);
//   BANK = true : a bank of patches, every voice renders with bank[its program] — oscillator kind,
//                 filter kind and every level are per-lane values (OSC == S2R_OSC_ANY).
// Same voice-per-lane layout, state arrays, mixdown staging and timed events as s2r_render_kernel
// (L = 1), but the plain frame loop: per frame the envelope cascade, the oscillator and the filter
// step; the oscillator constants and the filter coefficients (pow2, exp or sin/cos/tan, the
// divisions) are recomputed only when some lane's mod-envelope value differs from the previous
// frame's (wave-uniform test).  DESIGN.md 4.6, 4.7.
// ---------------------------------------------------------------------------------------
struct LanePatch {           // field names as in S2rRenderParams: frame_sisd & co. take either
    int32_t osc_kind;
    float osc_gain, noise_level, lpf_freq, amt_osc, amt_lpf;
    int32_t lpf_kind;
    float lpf_damping;
    S2rEnv amp, mod;
    float sr, rcp_sr;
};

__device__ __forceinline__ LanePatch lane_patch_from_bank(const S2rBankEntry *bank, uint32_t n, uint32_t program, float sr) {
    const S2rBankEntry e = bank[program < n ? program : 0u];     // an index past the bank renders with patch 0
    LanePatch lp;
    lp.osc_kind = e.osc_kind; lp.osc_gain = e.osc_gain; lp.noise_level = e.noise_level; lp.lpf_freq = e.lpf_freq;
    lp.amt_osc = e.amt_osc; lp.amt_lpf = e.amt_lpf; lp.lpf_kind = e.lpf_kind; lp.lpf_damping = e.lpf_shape;
    lp.amp = e.amp; lp.mod = e.mod; lp.sr = sr; lp.rcp_sr = 0.0f;
    return lp;
}

template <int OSC, bool BANK, int MAXT>
__global__ void __launch_bounds__(MAXT) s2r_render_general_kernel(const S2rRenderParams p) {
    const uint32_t kSuper = p.super_frames;
    const bool PV = p.per_voice != nullptr, TEV = p.tev != nullptr;          // wave-uniform
    __shared__ uint64_t sT[S2R_EXP2F_N];
    __shared__ __attribute__((aligned(8))) float sSin[(OSC == S2R_OSC_SINE || OSC == S2R_OSC_ANY) ? 2048 : 2];
    extern __shared__ float s_dyn[];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6, n_waves = blockDim.x >> 6;
    constexpr uint32_t VW = 64, GW = 4;
    const uint32_t n_groups = n_waves * GW;
    float *const sW = s_dyn;                                     // [2][n_groups][kSuper]
    float *const tile = s_dyn + 2 * n_groups * kSuper + wave * (kChunk * (VW + 1));
    const uint32_t vi = blockIdx.x * blockDim.x + tid;

    if (tid < S2R_EXP2F_N) sT[tid] = c_exp2f_table[tid];
    if (OSC == S2R_OSC_SINE || OSC == S2R_OSC_ANY)
        for (uint32_t i = tid; i < 1024u; i += blockDim.x) { sSin[2u * i] = p.sin_table[(i + 1023u) & 1023u]; sSin[2u * i + 1u] = p.sin_table[i]; }

    const bool in_range = vi < p.n_voices;
    const uint32_t flags = in_range ? p.v.flags[vi] : 0u;
    bool live = (flags & S2R_VF_STARTED) != 0u;                  // synth.rs:178
    LanePatch lp;
    uint32_t program = 0u;
    if (BANK) {
        program = live ? p.v.program[vi] : 0u;
        lp = lane_patch_from_bank(p.bank, p.bank_size, program, p.sr);
    } else {
        lp.osc_kind = p.osc_kind; lp.osc_gain = p.osc_gain; lp.noise_level = p.noise_level; lp.lpf_freq = p.lpf_freq;
        lp.amt_osc = p.amt_osc; lp.amt_lpf = p.amt_lpf; lp.lpf_kind = p.lpf_kind; lp.lpf_damping = p.lpf_damping;
        lp.amp = p.amp; lp.mod = p.mod; lp.sr = p.sr; lp.rcp_sr = p.rcp_sr;
    }
    VoiceRegs r;
    r.pitch = live ? p.v.pitch[vi] : 440.0f;
    r.offset = live ? p.v.offset[vi] : 0u;
    r.release_u = live ? p.v.release[vi] : 0u;
    r.released = live && (flags & S2R_VF_RELEASED) != 0u;
    r.phase = live ? p.v.phase[vi] : 0.0f;
    r.last = live ? p.v.lpf_last[vi] : 0.0f;
    Filt2 f2;
    f2.x1 = live ? p.v.fx1[vi] : 0.0f; f2.x2 = live ? p.v.fx2[vi] : 0.0f;
    f2.y1 = live ? p.v.fy1[vi] : 0.0f; f2.y2 = live ? p.v.fy2[vi] : 0.0f;
    const uint32_t seed = live ? p.v.seed[vi] : 0u;
    r.seed_rot = (seed << 5) | (seed >> 27);
    auto set_release_thresholds = [&]() {
        const float rf = r.released ? (float)r.release_u : 4294967296.0f;    // u32::MAX as f32
        r.ro_a = __builtin_fmaxf(rf, lp.amp.sus_off); r.end_a = r.ro_a + lp.amp.R;
        r.ro_m = __builtin_fmaxf(rf, lp.mod.sus_off); r.end_m = r.ro_m + lp.mod.R;
    };
    set_release_thresholds();

    int32_t ev_idx = -1;
    uint32_t ev_frame = 0xffffffffu;
    bool ev_dirty = false;
    uint32_t seed_now = seed;
    EnvRun ea = env_stage_at(lp.amp, r.ro_a, r.end_a, 0.0f), em = env_stage_at(lp.mod, r.ro_m, r.end_m, 0.0f);
    float thr_min = -__builtin_inff();       // forces the cascade on this lane's first frame
    uint32_t mod_key = 0xffffffffu;          // bits of the mod value `k`, `xc`, `fc` were computed for (a NaN: never equal)
    OscK k = make_osck_any<OSC>(lp.osc_kind, lp.sr / r.pitch);
    float xc = 0.0f;                         // one-pole coefficient exp(-2 pi f / sr), filters.rs:21
    FiltCoef fc; fc.alpha = fc.beta = fc.gamma = 0.0f;
    fc.k = lp.lpf_kind >= S2R_FILT_SVF_LP ? 1.0f / lp.lpf_damping : 0.0f;   // the SVF's k = 1 / q does not depend on the cutoff
    if (TEV && in_range) {
        ev_idx = p.voice_ev_head[vi];
        if (ev_idx >= 0) { ev_frame = p.tev[ev_idx].frame; p.voice_ev_head[vi] = -1; }
    }
    __syncthreads();

    const bool wave_live = __ballot(live || ev_idx >= 0) != 0ull;
    const uint32_t x16_frames = p.frames & ~(uint32_t)(kChunk - 1);
    const size_t pv_base = (size_t)vi * p.frames;
    float *bp = p.block_partials + (size_t)blockIdx.x * p.frames_stride;
    uint32_t buf = 0;
    // coefficient stream of this wave's 64-voice group (single patch, 256-thread workgroups): alpha / beta / gamma
    // per frame, and under oscillator FM the period and its reciprocal
    constexpr bool kFast = !BANK && MAXT == 256;
    const bool fmv = p.amt_osc != 0.0f;                          // kernel argument: uniform
    int32_t slot = -1;
    if (kFast && p.use_coeff && p.coeff_count[p.coeff_parity] <= p.coeff_capacity)
        slot = p.group_slot[__builtin_amdgcn_readfirstlane(vi / 64u)];
    const bool have_stream = slot >= 0;
    const uint32_t stream_vecs = 3u + (fmv ? 2u : 0u);
    const f4 *stream = (const f4 *)p.coeff + ((size_t)(have_stream ? slot : 0) * (x16_frames / kP)) * stream_vecs * 64u + lane;

    auto reduce_chunk = [&](uint32_t f_base, uint32_t n_frames) {
        const uint32_t f = lane & 15u, grp = lane >> 4;
        if (f < n_frames) {
            const float *src = tile + f * (VW + 1) + grp * 16u;
            float acc = src[0];
#pragma unroll
            for (int q = 1; q < 16; ++q) acc += src[q];
            sW[(buf * n_groups + wave * GW + grp) * kSuper + f_base + f] = acc;
        }
    };
    auto apply_events_at = [&](uint32_t fpos) {
        if (__ballot(ev_frame == fpos) == 0ull) return;
        while (ev_frame == fpos) {
            const S2rTimedEvent e = p.tev[ev_idx];
            if (e.flags & S2R_EV_RESTART) {                      // *voice = Voice { .. }, synth.rs:63-69
                r.pitch = e.pitch;
                r.offset = 0u - fpos;
                r.release_u = 0u;
                r.released = (e.flags & S2R_EV_RELEASE) != 0u;
                r.phase = 0.0f; r.last = 0.0f;
                mod_key = 0xffffffffu;                           // the oscillator constants depend on the pitch
                f2.x1 = f2.x2 = f2.y1 = f2.y2 = 0.0f;
                seed_now = e.seed;
                r.seed_rot = (e.seed << 5) | (e.seed >> 27);
                if (BANK) { program = e.program; lp = lane_patch_from_bank(p.bank, p.bank_size, program, p.sr); }
                k = make_osck_any<OSC>(lp.osc_kind, lp.sr / r.pitch);    // the new pitch's constants (what FM-free frames use)
                live = true;
            } else if ((e.flags & S2R_EV_RELEASE) && live && !r.released) {   // synth.rs:74-75
                r.released = true;
                r.release_u = r.offset + fpos;
            }
            ev_dirty = true;
            ev_idx = e.next;
            ev_frame = ev_idx >= 0 ? p.tev[ev_idx].frame : 0xffffffffu;
        }
        set_release_thresholds();
        // the envelope cascade for the boundary's frame, so that the chunk can take the branch-free path at once
        const float t0 = (float)(r.offset + fpos);
        ea = env_stage_at(lp.amp, r.ro_a, r.end_a, t0);
        em = env_stage_at(lp.mod, r.ro_m, r.end_m, t0);
        thr_min = __builtin_fminf(ea.thr, em.thr);
    };

    // everything below `mod` that depends on it alone (and on the voice's pitch) is kept while no lane's
    // mod-envelope value changes
    auto refresh = [&](float mod) {
        if (__ballot(s2r_f2u(mod) != mod_key) == 0ull) return;
        const float f_osc = s2r_pow2_sleef_core(mod * lp.amt_osc) * r.pitch;     // process.rs:146-147,231-250
        const float f_lpf = s2r_pow2_sleef_core(mod * lp.amt_lpf) * lp.lpf_freq; // process.rs:148-152
        k = make_osck_any<OSC>(lp.osc_kind, lp.sr / f_osc);                      // units.rs:32-42
        if (lp.lpf_kind == S2R_FILT_ONEPOLE) xc = lpf_coeff<false, LanePatch>(lp, f_lpf, sT);
        else fc = dsp_filter_coef(lp.lpf_kind, lp.lpf_damping, lp.sr, f_lpf);
        mod_key = s2r_f2u(mod);
    };

    for (uint32_t sc0 = 0; sc0 < p.frames; sc0 += kSuper) {
        const uint32_t n_sc = (p.frames - sc0 < kSuper) ? (p.frames - sc0) : kSuper;
        const uint32_t n_x16 = (x16_frames > sc0) ? ((x16_frames - sc0 < n_sc) ? (x16_frames - sc0) : n_sc) : 0u;
        if (wave_live) {
            for (uint32_t c16 = 0; c16 < n_x16; c16 += kChunk) {
                if (TEV) apply_events_at(sc0 + c16);
                const bool calm = __ballot(!((float)(r.offset + sc0 + c16 + (kChunk - 1u)) < thr_min)) == 0ull;
                // every lane's mod envelope sits in a zero-slope stage for the whole chunk: its value (0 * (t -
                // base) + y0, the same bits on every frame) is looked at once instead of per frame
                const bool flat = calm && __ballot(em.slope != 0.0f) == 0ull;
                if (flat && !have_stream) refresh(env_value(em, (float)(r.offset + sc0 + c16)));
                if (kFast && calm && (flat || have_stream)) {
                    // One patch, every lane's mod envelope flat: oscillator constants and filter coefficients are
                    // constants of the run, so the branch-free chunk of the one-pole kernel applies with this
                    // patch's filter in its recurrence.  Same run rule: the LAST frame of the run is below every
                    // lane's next threshold.
                    const uint32_t left = (n_x16 - c16) / kChunk;
                    const uint32_t o_chunk = r.offset + sc0 + c16;
                    auto clear_for = [&](uint32_t n) {
                        return __ballot(!((float)(o_chunk + n * kChunk - 1u) < thr_min)) == 0ull;
                    };
                    uint32_t run = 1u;                           // `calm` already covers this chunk
                    uint32_t most = left;                        // under timed events: up to the wave's next one
                    if (TEV) {
                        uint32_t nxt = ev_frame;                 // 0xffffffff: none
#pragma unroll
                        for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)nxt, sh); nxt = o < nxt ? o : nxt; }
                        nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt);
                        const uint32_t until = nxt == 0xffffffffu ? left : (nxt - (sc0 + c16)) / kChunk;
                        most = until < left ? until : left;
                    }
                    if (most > 1u) { if (clear_for(most)) run = most; else if (most > 4u && clear_for(4u)) run = 4u; }
                    if (__ballot(!(((have_stream && fmv) || (k.period > 0.0f && k.period < __builtin_inff())) && r.phase >= 0.0f && r.phase < 1.0f)) != 0ull) run = 0u;
                    if (run) {
                        run = (uint32_t)__builtin_amdgcn_readfirstlane((int)run);
                        FlatCache fcx; fcx.xc = xc; fcx.k = k;
                        const FiltCoef fcc = fc;                 // the run's constants, by value
                        auto run_chunks = [&](auto filt_tag) {
                            constexpr int FILT = decltype(filt_tag)::value;
                            constexpr int O = OSC == S2R_OSC_ANY ? 0 : OSC;
                            auto loop = [&](auto src_tag, auto fmv_tag) {          // one tight loop per coefficient source
                                constexpr int SRC = decltype(src_tag)::value;
                                constexpr bool FMV = decltype(fmv_tag)::value;
                                for (uint32_t i = 0; i < run; ++i) {
                                    const uint32_t f0 = c16 + i * kChunk;
                                    float *pvd = (PV && in_range) ? p.per_voice + pv_base + sc0 + f0 : nullptr;
                                    const f4 *sq = stream + (size_t)((sc0 + f0) / kP) * stream_vecs * 64u;
                                    chunk_fast<O, SRC, FILT, FMV>(p, r, ea, em, fcx, k, o_chunk + i * kChunk, sq, sT, sSin, live, tile + lane, VW + 1, pvd, &fcc, &f2);
                                    reduce_chunk(f0, kChunk);
                                }
                            };
                            if (!have_stream) loop(std::integral_constant<int, 0>{}, std::false_type{});
                            else if (fmv) loop(std::integral_constant<int, 1>{}, std::true_type{});
                            else loop(std::integral_constant<int, 1>{}, std::false_type{});
                        };
                        switch (lp.lpf_kind) {                   // wave-uniform: the patch is a kernel argument here
                        case S2R_FILT_LP1: run_chunks(std::integral_constant<int, S2R_FILT_LP1>{}); break;
                        case S2R_FILT_HP1: run_chunks(std::integral_constant<int, S2R_FILT_HP1>{}); break;
                        case S2R_FILT_LP2: run_chunks(std::integral_constant<int, S2R_FILT_LP2>{}); break;
                        case S2R_FILT_HP2: run_chunks(std::integral_constant<int, S2R_FILT_HP2>{}); break;
                        case S2R_FILT_BP2: run_chunks(std::integral_constant<int, S2R_FILT_BP2>{}); break;
                        case S2R_FILT_SVF_LP: run_chunks(std::integral_constant<int, S2R_FILT_SVF_LP>{}); break;
                        case S2R_FILT_SVF_BP: run_chunks(std::integral_constant<int, S2R_FILT_SVF_BP>{}); break;
                        case S2R_FILT_SVF_HP: run_chunks(std::integral_constant<int, S2R_FILT_SVF_HP>{}); break;
                        default: break;                          // (the one-pole has its own kernel)
                        }
                        c16 += (run - 1u) * kChunk;
                        continue;
                    }
                }
                for (uint32_t j = 0; j < kChunk; ++j) {          // one frame of sample_voice_x16, process.rs:306-379
                    const uint32_t oi = r.offset + sc0 + c16 + j;            // wrapping u32 add (process.rs:213-219)
                    const float t = (float)oi;
                    // the active stages' lines stay valid until t reaches the earlier of their thresholds
                    // (EnvRun, above); `calm`: no lane gets there inside this chunk, so nobody even looks
                    if (!calm && !(t < thr_min)) {
                        ea = env_stage_at(lp.amp, r.ro_a, r.end_a, t);
                        em = env_stage_at(lp.mod, r.ro_m, r.end_m, t);
                        thr_min = __builtin_fminf(ea.thr, em.thr);
                    }
                    const float amp = env_value(ea, t);                                           // process.rs:144
                    const float mod = env_value(em, t);                                           // process.rs:145
                    if (!flat) refresh(mod);
                    const float nz = hash_noise(r.seed_rot, t) + lp.noise_level;                 // process.rs:347-356 (ADD)
                    const float ph = r.phase;                                                    // oscillators.rs:391-400
                    r.phase = s2r_fmod1(ph + k.inv_period);
                    const float off = __builtin_fmaf(k.period, ph, 0.0f);
                    const float osc = osc_value_any<OSC>(lp.osc_kind, k, off, sSin);
                    const float smp = (osc + lp.osc_gain) + nz;                                  // process.rs:342-345 (ADD), :358
                    float y;
                    if (lp.lpf_kind == S2R_FILT_ONEPOLE) y = lpf_apply(xc, smp, r.last);         // process.rs:363-371
                    else y = dsp_filter_apply(lp.lpf_kind, fc, smp, f2);
                    const float out = live ? y * amp : 0.0f;                                     // process.rs:373-376
                    if (PV && in_range) p.per_voice[pv_base + sc0 + c16 + j] = out;
                    tile[j * (VW + 1) + lane] = out;
                }
                reduce_chunk(c16, kChunk);
            }
            if (n_x16 < n_sc) {                                  // scalar tail (< 16 frames)
                if (TEV) apply_events_at(sc0 + n_x16);
                for (uint32_t i = n_x16; i < n_sc; ++i) {
                    float out;
                    if (lp.lpf_kind == S2R_FILT_ONEPOLE) out = frame_sisd<OSC, false, LanePatch>(lp, r, r.offset + sc0 + i, sT, sSin, nullptr);
                    else out = frame_sisd<OSC, true, LanePatch>(lp, r, r.offset + sc0 + i, sT, sSin, &f2);
                    out = live ? out : 0.0f;
                    if (PV && in_range) p.per_voice[pv_base + sc0 + i] = out;
                    tile[(i - n_x16) * (VW + 1) + lane] = out;
                }
                reduce_chunk(n_x16, n_sc - n_x16);
            }
        } else {
            for (uint32_t i = lane; i < GW * kSuper; i += 64u)
                sW[(buf * n_groups + wave * GW + i / kSuper) * kSuper + (i % kSuper)] = 0.0f;
            if (PV && in_range) for (uint32_t i = 0; i < n_sc; ++i) p.per_voice[pv_base + sc0 + i] = 0.0f;
        }
        __syncthreads();
        for (uint32_t f = tid; f < n_sc; f += blockDim.x) {
            float acc = sW[(buf * n_groups + 0) * kSuper + f];
            for (uint32_t gq = 1; gq < n_groups; ++gq) acc += sW[(buf * n_groups + gq) * kSuper + f];
            bp[sc0 + f] = acc;
            if (p.direct_out) {                                  // one workgroup: this IS the mix
                const float total = 0.0f + acc;                  // accum = splat(0.0), synth.rs:176
                if (p.direct_stereo) { p.direct_out[2u * (sc0 + f)] = total; p.direct_out[2u * (sc0 + f) + 1u] = total; }
                else p.direct_out[sc0 + f] = total;
            }
        }
        buf ^= 1u;
    }

    if (live) {
        const uint32_t o = r.offset;
        p.v.offset[vi] = (!ev_dirty && o > 0xffffffffu - p.frames) ? 0xffffffffu : o + p.frames;   // synth.rs:197
        p.v.phase[vi] = r.phase;
        p.v.lpf_last[vi] = r.last;
        p.v.fx1[vi] = f2.x1; p.v.fx2[vi] = f2.x2; p.v.fy1[vi] = f2.y1; p.v.fy2[vi] = f2.y2;
        if (ev_dirty) {
            p.v.pitch[vi] = r.pitch;
            p.v.release[vi] = r.release_u;
            p.v.flags[vi] = S2R_VF_STARTED | (r.released ? S2R_VF_RELEASED : 0u);
            p.v.seed[vi] = seed_now;
            if (BANK) p.v.program[vi] = program;
        }
    }
}

// ---------------------------------------------------------------------------------------
// mix kernel: adds the workgroup partial rows in the fixed order of DESIGN.md 4.3:
//   runs of 16 consecutive workgroups sequentially -> the run sums of a mix group sequentially
//   -> the mix groups sequentially -> root (+0.0) + total.
// One workgroup handles 16 frames: thread (slot, f) adds whole runs (16 independent loads in
// flight each), the run sums meet in LDS, 16 threads finish.  Runs never straddle a mix group.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kMixRun = 16;

__global__ void __launch_bounds__(256) s2r_mix_kernel(const S2rMixParams m) {
    extern __shared__ float s_run[];                             // [total runs][16 frames]
    const uint32_t f_local = threadIdx.x & 15u, slot = threadIdx.x >> 4;
    const uint32_t f = blockIdx.x * 16u + f_local;
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
            for (uint32_t j = 0; j < kMixRun; ++j) v[j] = (b0 + j < gb1) ? m.block_partials[(size_t)(b0 + j) * m.frames_stride + f] : 0.0f;
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
        if (m.stereo) { m.out[2 * f] = total; m.out[2 * f + 1] = total; }
        else m.out[f] = total;
    }
}

// out[i] = ((+0.0 + rows[0][i]) + rows[1][i]) + ...   (rank-order combine of shard partials)
__global__ void s2r_sum_rows_kernel(const float *rows, uint32_t n_rows, uint32_t frames, float *out) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames) return;
    float total = 0.0f;
    for (uint32_t r = 0; r < n_rows; ++r) total += rows[(size_t)r * frames + f];
    out[f] = total;
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
__global__ void s2r_tev_heads_kernel(int32_t *heads, const S2rTimedEvent *tev, S2rTimedEvent *tev_copy, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const S2rTimedEvent e = tev[i];
    tev_copy[i] = e;
    if (e.flags & S2R_TEV_FIRST) heads[e.voice] = (int32_t)i;
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
    } else if (e.flags & S2R_EV_RELEASE) {                       // synth.rs:74-75
        const uint32_t fl = v.flags[vi];
        if ((fl & S2R_VF_STARTED) && !(fl & S2R_VF_RELEASED)) {
            v.release[vi] = v.offset[vi];
            v.flags[vi] = fl | S2R_VF_RELEASED;
        }
    }
}

template <int OSC, bool FM, int MODE>
hipError_t launch_l(const S2rRenderParams &p0, uint32_t block_voices, uint32_t lanes, hipStream_t stream) {
    S2rRenderParams p = p0;
    const uint32_t grid = (p.n_voices + block_voices - 1) / block_voices;
    const dim3 block(block_voices * lanes);
    const uint32_t n_waves = block_voices * lanes / 64, vw = 64 / lanes, n_groups = n_waves * (vw / 16);
    // keeps the staging (group sums + two transpose tiles per wave) under the 160 KiB of LDS
    p.super_frames = n_groups <= 16 ? kSuperMax : (n_groups <= 32 ? 64u : 32u);
    const size_t lds = sizeof(float) * ((size_t)2 * n_groups * p.super_frames + (size_t)n_waves * 2 * kChunk * (vw + 1));
    // the launch bound is the register budget: 256-thread workgroups (one wave per SIMD) may use
    // the whole file, which the 4-frame vector code wants; bigger workgroups get what is left
    const uint32_t threads = block_voices * lanes;
    switch (lanes) {
    case 1:
        if (threads <= 256) hipLaunchKernelGGL((s2r_render_kernel<OSC, FM, MODE, 1, 256>), dim3(grid), block, lds, stream, p);
        else hipLaunchKernelGGL((s2r_render_kernel<OSC, FM, MODE, 1, 1024>), dim3(grid), block, lds, stream, p);
        break;
#if defined(S2R_WITH_LANE_VARIANTS)
    // 2 or 4 lanes per voice: bit-identical, slower at every pool size since the branch-free runs (which exist for
    // one lane per voice only), and 48 more kernels to compile — built only on request (synth2_amd/build.py)
    case 2:
        if (threads > 512) return hipErrorInvalidValue;
        hipLaunchKernelGGL((s2r_render_kernel<OSC, FM, MODE, 2, 512>), dim3(grid), block, lds, stream, p);
        break;
    case 4: hipLaunchKernelGGL((s2r_render_kernel<OSC, FM, MODE, 4, 1024>), dim3(grid), block, lds, stream, p); break;
#endif
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int OSC, bool BANK>
hipError_t launch_general(const S2rRenderParams &p0, uint32_t block_voices, hipStream_t stream) {
    S2rRenderParams p = p0;
    const uint32_t grid = (p.n_voices + block_voices - 1) / block_voices;
    const uint32_t n_waves = block_voices / 64, n_groups = n_waves * 4;
    p.super_frames = n_groups <= 16 ? kSuperMax : 64u;
    const size_t lds = sizeof(float) * ((size_t)2 * n_groups * p.super_frames + (size_t)n_waves * kChunk * 65);
    // the launch bound is the register budget (as for s2r_render_kernel): up to 256 threads get the whole file
    if (block_voices <= 256) hipLaunchKernelGGL((s2r_render_general_kernel<OSC, BANK, 256>), dim3(grid), dim3(block_voices), lds, stream, p);
    else hipLaunchKernelGGL((s2r_render_general_kernel<OSC, BANK, 1024>), dim3(grid), dim3(block_voices), lds, stream, p);
    return hipGetLastError();
}

template <int OSC>
hipError_t launch_osc(const S2rRenderParams &p, uint32_t block_voices, uint32_t lanes, hipStream_t stream) {
    if (p.lpf_kind != S2R_FILT_ONEPOLE) return launch_general<OSC, false>(p, block_voices, stream);
    // pow(2, mod * amount) == 1 exactly iff amount is +-0 (mod is always finite and >= 0)
    const bool fm = p.amt_osc != 0.0f;
    const int mode = p.per_voice != nullptr ? 1 : (p.tev != nullptr ? 2 : 0);
    if (p.per_voice != nullptr && p.tev != nullptr) return hipErrorInvalidValue;   // the host refuses this combination
    switch (mode) {
    case 1: return fm ? launch_l<OSC, true, 1>(p, block_voices, lanes, stream) : launch_l<OSC, false, 1>(p, block_voices, lanes, stream);
    case 2: return fm ? launch_l<OSC, true, 2>(p, block_voices, lanes, stream) : launch_l<OSC, false, 2>(p, block_voices, lanes, stream);
    default: return fm ? launch_l<OSC, true, 0>(p, block_voices, lanes, stream) : launch_l<OSC, false, 0>(p, block_voices, lanes, stream);
    }
}

}  // namespace

static void launch_coeff_pass(const S2rRenderParams &p, hipStream_t stream) {
    const bool fm = p.amt_osc != 0.0f;
    if (p.lpf_kind != S2R_FILT_ONEPOLE) {
        if (fm) hipLaunchKernelGGL((s2r_coeff_kernel<false, true, true>), dim3(2048), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((s2r_coeff_kernel<false, false, true>), dim3(2048), dim3(256), 0, stream, p);
    } else if (p.fast_div_sr) {
        if (fm) hipLaunchKernelGGL((s2r_coeff_kernel<true, true, false>), dim3(2048), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((s2r_coeff_kernel<true, false, false>), dim3(2048), dim3(256), 0, stream, p);
    } else {
        if (fm) hipLaunchKernelGGL((s2r_coeff_kernel<false, true, false>), dim3(2048), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((s2r_coeff_kernel<false, false, false>), dim3(2048), dim3(256), 0, stream, p);
    }
}

hipError_t s2r_launch_coeff(const S2rRenderParams &p, hipStream_t stream) {
    if (!p.use_coeff || p.n_voices == 0 || p.frames < 16u) return hipSuccess;
    const uint32_t n_groups64 = (p.n_voices + 63u) / 64u;
    hipLaunchKernelGGL(s2r_classify_kernel, dim3(n_groups64), dim3(64), 0, stream, p);
    launch_coeff_pass(p, stream);
    return hipGetLastError();
}

hipError_t s2r_launch_prep(const S2rPrepParams &a, hipStream_t stream) {
    if (a.p.n_voices == 0 || a.p.frames < 16u || !a.p.use_coeff) return hipErrorInvalidValue;
    const uint32_t n_groups64 = (a.p.n_voices + 63u) / 64u;
    hipLaunchKernelGGL(s2r_prep_kernel, dim3(n_groups64), dim3(64), 0, stream, a);
    launch_coeff_pass(a.p, stream);
    return hipGetLastError();
}

bool s2r_lane_variants_built() {
#if defined(S2R_WITH_LANE_VARIANTS)
    return true;
#else
    return false;
#endif
}

hipError_t s2r_launch_render(const S2rRenderParams &p, uint32_t block_voices, uint32_t lanes_per_voice, hipStream_t stream) {
    if (p.n_voices == 0 || p.frames == 0) return hipSuccess;
    if (block_voices < 64 || block_voices > 1024 || (block_voices & 63u)) return hipErrorInvalidValue;
    if (block_voices * lanes_per_voice > 1024 || (lanes_per_voice == 2 && block_voices > 256)) return hipErrorInvalidValue;
    if (p.bank_size > 1) return launch_general<S2R_OSC_ANY, true>(p, block_voices, stream);
    switch (p.osc_kind) {
    case S2R_OSC_SQUARE: return launch_osc<S2R_OSC_SQUARE>(p, block_voices, lanes_per_voice, stream);
    case S2R_OSC_SAW: return launch_osc<S2R_OSC_SAW>(p, block_voices, lanes_per_voice, stream);
    case S2R_OSC_TRIANGLE: return launch_osc<S2R_OSC_TRIANGLE>(p, block_voices, lanes_per_voice, stream);
    case S2R_OSC_SINE: return launch_osc<S2R_OSC_SINE>(p, block_voices, lanes_per_voice, stream);
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

hipError_t s2r_launch_events(const S2rVoiceArrays &v, const S2rVoiceEvent *dev_events, uint32_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(s2r_events_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, v, dev_events, n);
    return hipGetLastError();
}

hipError_t s2r_launch_tev_heads(int32_t *heads, const S2rTimedEvent *tev, S2rTimedEvent *tev_copy, uint32_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(s2r_tev_heads_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, heads, tev, tev_copy, n);
    return hipGetLastError();
}

hipError_t s2r_launch_decimate4(float *x_with_history, const float *taps, uint32_t n_out, float *out, hipStream_t stream) {
    if (n_out == 0) return hipSuccess;
    hipLaunchKernelGGL(s2r_decimate4_kernel, dim3((n_out + 255) / 256), dim3(256), 0, stream, x_with_history, taps, n_out, out);
    hipLaunchKernelGGL(s2r_decimate4_history_kernel, dim3(1), dim3(64), 0, stream, x_with_history, n_out);
    return hipGetLastError();
}

hipError_t s2r_launch_sum_rows(const float *rows, uint32_t n_rows, uint32_t frames, float *out, hipStream_t stream) {
    if (frames == 0) return hipSuccess;
    hipLaunchKernelGGL(s2r_sum_rows_kernel, dim3((frames + 255) / 256), dim3(256), 0, stream, rows, n_rows, frames, out);
    return hipGetLastError();
}
