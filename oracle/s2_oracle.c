/*
 * s2_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see s2_oracle.h).
 *
 * Restates, function by function, the reference's voice-render path.  Citations are
 * relative to /root/reference/components/s2_lib/src/.  The arithmetic follows the
 * reference's default build: cargo feature "fma" ON (components/s2_lib/Cargo.toml:6-8),
 * no contraction / fast-math otherwise.  Build with -ffp-contract=off.
 *
 * The x16 path (what runs for every buffer length that is a multiple of 16) and the
 * scalar "sisd" tail path are both here and deliberately DISAGREE exactly where the
 * reference does (process.rs:342-345,353-356 add the gains; process.rs:287,292 multiply).
 */
#define _GNU_SOURCE
#include "s2_oracle.h"
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ units.rs */

/* units.rs:44-53  Ms::as_samples: seconds = ms / 1000.0; samples = sample_rate * seconds */
float s2o_ms_as_samples(float ms, uint32_t sample_rate) {
    float sr = (float)sample_rate;
    float seconds = ms / 1000.0f;
    return sr * seconds;
}

/* units.rs:19-26 / 32-42  Hz::as_samples: sample_rate / hz */
static float hz_as_samples(float hz, uint32_t sample_rate) {
    return (float)sample_rate / hz;
}

/* ------------------------------------------------------------------ math.rs */

/* math.rs:11-19 (scalar) and :27-40 (x16), feature fma: slope = rise/run; fma(slope,x,y0) */
static float line_fma(float y_rise, float x_run, float x_value, float y_offset) {
    float slope = y_rise / x_run;
    return fmaf(slope, x_value, y_offset);
}

/* old/simdtest.rs:247-261: the x16 ADSR's own helper — divide, multiply, add (no fma) */
static float line_nofma(float y_rise, float x_run, float x_value, float y_offset) {
    float slope = y_rise / x_run;
    float y_value = slope * x_value;
    return y_value + y_offset;
}

/* Rust `as u32` / Simd::cast::<u32>() on f32: saturating, NaN -> 0 */
static uint32_t f32_as_u32(float f) {
    if (!(f > 0.0f)) return 0;            /* negative, -0, NaN */
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}

/* ------------------------------------------------------------------ envelopes */

/* old/simdtest.rs:270-331  AdsrX16::sample, one lane at a time */
void s2o_adsr_x16(float attack, float decay, float sustain, float release,
                  const uint32_t offset_u[16], int has_release, uint32_t release_offset_u,
                  float out[16]) {
    for (int i = 0; i < 16; i++) {
        float offset = (float)offset_u[i];                                  /* :277-279 */
        float decay_offset = attack;                                        /* :281 */
        float sustain_offset = attack + decay;                              /* :282 */
        float release_offset = (float)(has_release ? release_offset_u : 0xffffffffu); /* :283 */
        release_offset = fmaxf(release_offset, sustain_offset);             /* :285 simd_max */
        float end_offset = release_offset + release;                        /* :286 */

        int in_attack = offset < decay_offset;                              /* :288-292 */
        int in_decay = !in_attack && offset < sustain_offset;
        int in_sustain = !in_attack && !in_decay && offset < release_offset;
        int in_release = !in_attack && !in_decay && !in_sustain && offset < end_offset;
        int in_end = !in_attack && !in_decay && !in_sustain && !in_release;

        float attack_sample = line_nofma(1.0f, attack, offset, 0.0f);                         /* :294-300 */
        float decay_sample = line_nofma(sustain - 1.0f, decay, offset - decay_offset, 1.0f);  /* :302-308 */
        float sustain_sample = sustain;                                                       /* :310 */
        float release_sample = line_nofma(-sustain, release, offset - release_offset, sustain); /* :312-318 */
        float end_sample = 0.0f;

        float sample = 0.0f;                                                /* :322-327 */
        if (in_attack) sample = attack_sample;
        if (in_decay) sample = decay_sample;
        if (in_sustain) sample = sustain_sample;
        if (in_release) sample = release_sample;
        if (in_end) sample = end_sample;
        out[i] = sample;
    }
}

/* envelopes.rs:21-150  Adsr::sample (scalar path; release starts from the CURRENT level) */
float s2o_adsr_scalar(float attack, float decay, float sustain, float release,
                      uint32_t offset_u, int has_release, uint32_t release_offset_u) {
    float offset = (float)offset_u;                                          /* :32 */
    float decay_offset = attack;
    float sustain_offset = attack + decay;
    float release_offset = (float)(has_release ? release_offset_u : 0xffffffffu); /* :35 */
    float end_offset = release_offset + release;

    int in_release = offset >= release_offset && offset < end_offset;       /* :39-43 */
    int in_end = offset >= end_offset;
    int in_attack = !in_release && !in_end && offset < decay_offset;
    int in_decay = !in_release && !in_end && !in_attack && offset < sustain_offset;
    int in_sustain = !in_release && !in_end && !in_attack && !in_decay && offset < release_offset;

    /* :57-63 release_start_stage, :68-93 release_start_sample */
    float release_start_sample;
    if (release_offset < decay_offset)
        release_start_sample = line_fma(1.0f, attack, release_offset, 0.0f);
    else if (release_offset < sustain_offset)
        release_start_sample = line_fma(sustain - 1.0f, decay, release_offset - decay_offset, 1.0f);
    else
        release_start_sample = sustain;

    if (in_attack) return line_fma(1.0f, attack, offset, 0.0f);                            /* :96-107 */
    if (in_decay) return line_fma(sustain - 1.0f, decay, offset - decay_offset, 1.0f);     /* :108-119 */
    if (in_sustain) return sustain;                                                        /* :120-127 */
    if (in_release)                                                                        /* :128-139 */
        return line_fma(-release_start_sample, release, offset - release_offset, release_start_sample);
    return 0.0f;                                                                           /* :140-147 */
}

/* ------------------------------------------------------------------ hashnoise.rs */

#define SEED32 0x9e3779b9u                                                  /* :7 */

uint32_t s2o_hash_word(uint32_t start, uint32_t word) {                      /* :53-55 */
    uint32_t rot = (start << 5) | (start >> 27);
    return (rot ^ word) * SEED32;
}

void s2o_hash_word_x16(const uint32_t start[16], const uint32_t word[16], uint32_t out[16]) { /* :57-68 */
    for (int i = 0; i < 16; i++) {
        uint32_t leftshift = start[i] << 5;
        uint32_t rightshift = start[i] >> (32 - 5);
        uint32_t rotated = leftshift | rightshift;
        out[i] = (rotated ^ word[i]) * SEED32;
    }
}

float s2o_hash_noise(uint32_t seed, float offset) {                          /* :14-27 */
    uint32_t off = f32_as_u32(offset);
    uint32_t hash = s2o_hash_word(seed, off);
    uint16_t value16 = (uint16_t)hash;
    float value = (float)value16;
    float u16_max = 65535.0f;
    return value / u16_max * 2.0f - 1.0f;
}

void s2o_hash_noise_x16(uint32_t seed, const float offset[16], float out[16]) { /* :33-51 */
    uint32_t off[16], start[16], hash[16];
    for (int i = 0; i < 16; i++) { off[i] = f32_as_u32(offset[i]); start[i] = seed; }
    s2o_hash_word_x16(start, off, hash);
    for (int i = 0; i < 16; i++) {
        uint16_t v16 = (uint16_t)hash[i];        /* cast::<u16>() truncates */
        float value = (float)v16;
        out[i] = value / 65535.0f * 2.0f - 1.0f;
    }
}

/* ------------------------------------------------------------------ tables.rs / lookup.rs */

static float g_sin_table[1024];
static pthread_once_t g_sin_once = PTHREAD_ONCE_INIT;

/* components/s2_bin/src/tables.rs:6-10: i = i as f32 / 1024.0; i = i * PI * 2.0; sin(i).
 * The literals the reference ships (s2_lib try3/tables.rs:2-1025) equal the correctly
 * rounded f32 sine of that f32 argument everywhere except four entries that the author's
 * libm rounded one ULP further from zero; tests/test_sin_table.py diffs this against the
 * reference file whenever /root/reference is present and pins CRC32 0x55293b66. */
static void sin_table_init(void) {
    for (int k = 0; k < 1024; k++) {
        float i = (float)k / 1024.0f;
        i = i * 3.14159274101257324f * 2.0f;
        g_sin_table[k] = (float)sin((double)i);
    }
    static const int plus_one_ulp_in_magnitude[4] = { 395, 399, 610, 627 };
    for (int j = 0; j < 4; j++) {
        uint32_t u; memcpy(&u, &g_sin_table[plus_one_ulp_in_magnitude[j]], 4);
        u += 1;
        memcpy(&g_sin_table[plus_one_ulp_in_magnitude[j]], &u, 4);
    }
}
const float *s2o_sin_table(void) { pthread_once(&g_sin_once, sin_table_init); return g_sin_table; }

/* lookup.rs:10-44  table_lookup_exclusive */
float s2o_table_lookup_exclusive(const float *table, uint32_t len, float value, float range, int *panicked) {
    float table_length = (float)len;
    float table_value = value * table_length / range;                       /* :22 */
    uint32_t low = f32_as_u32(table_value);                                  /* :23 */
    uint32_t idx1 = low;
    uint32_t idx2 = (idx1 + 1u) % len;                                       /* :26 (u32 +1: debug-overflow aside) */
    float low_f = (float)low;
    if (idx1 >= len) { if (panicked) *panicked = 1; return 0.0f; }           /* :31 index panic */
    float sample1 = table[idx1], sample2 = table[idx2];
    return line_fma(sample2 - sample1, 1.0f, table_value - low_f, sample1);  /* :34-42 */
}

/* lookup.rs:92-130  table_lookup_inclusive */
float s2o_table_lookup_inclusive(const float *table, uint32_t len, float value, float range, int *panicked) {
    float table_length = (float)(len ? len - 1u : 0u);                       /* :106-108 */
    float table_value = value * table_length / range;
    uint32_t low = f32_as_u32(table_value);
    uint32_t idx1 = low;
    uint32_t idx2 = (idx1 + 1u) % len;
    float low_f = (float)low;
    if (idx1 >= len) { if (panicked) *panicked = 1; return 0.0f; }
    float sample1 = table[idx1], sample2 = table[idx2];
    return line_fma(sample2 - sample1, 1.0f, table_value - low_f, sample1);
}

static void lookup_x16(const float *table, uint32_t len, float table_length,
                       const float value[16], const float range[16], float out[16]) {
    for (int i = 0; i < 16; i++) {
        float table_value = value[i] * table_length / range[i];              /* :63 / :150 */
        uint32_t low = f32_as_u32(table_value);                              /* :64 cast::<u32>() */
        uint32_t idx1 = low;
        uint32_t idx2 = (idx1 + 1u) % len;                                   /* :67 wrapping simd add */
        float low_f = (float)low;
        float sample1 = idx1 < len ? table[idx1] : 0.0f;                     /* :72 gather_or_default */
        float sample2 = idx2 < len ? table[idx2] : 0.0f;
        out[i] = line_fma(sample2 - sample1, 1.0f, table_value - low_f, sample1); /* :75-84 */
    }
}

/* lookup.rs:46-85 */
void s2o_table_lookup_exclusive_x16(const float *table, uint32_t len, const float value[16], const float range[16], float out[16]) {
    lookup_x16(table, len, (float)len, value, range, out);
}
/* lookup.rs:132-172 */
void s2o_table_lookup_inclusive_x16(const float *table, uint32_t len, const float value[16], const float range[16], float out[16]) {
    lookup_x16(table, len, (float)(len ? len - 1u : 0u), value, range, out);
}
/* lookup.rs:187-199 */
void s2o_table_lookup_periodic_x16(const float *table, uint32_t len, const float value[16], const float range[16], float out[16]) {
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = fmodf(value[i], range[i]);
    s2o_table_lookup_exclusive_x16(table, len, v, range, out);
}

/* ------------------------------------------------------------------ filters.rs */

/* filters.rs:16-34  LowPassFilter::process (feature fma) */
float s2o_lpf_process(float *last, uint32_t sample_rate_u, float freq, float input) {
    float sample_rate = (float)sample_rate_u;
    const float pi = 3.14159274101257324f;
    float x = expf(-2.0f * pi * freq / sample_rate);        /* ((-2*pi)*freq)/sr ; Rust f32::exp = libm expf */
    float a0 = 1.0f - x;
    float b1 = -x;
    float out = fmaf(a0, input, -b1 * *last);
    *last = out;
    return out;
}

/* ------------------------------------------------------------------ build-defined 4x decimator */

void s2o_decim4_taps(float *h) {
    const double PI = 3.14159265358979323846, fc = 0.115;
    double d[S2O_DECIM_TAPS], sum = 0.0;
    for (int k = 0; k < S2O_DECIM_TAPS; k++) {
        const double t = (double)(k - (S2O_DECIM_TAPS - 1) / 2);
        const double ideal = t == 0.0 ? 2.0 * fc : sin(2.0 * PI * fc * t) / (PI * t);
        const double w = 0.42 - 0.5 * cos(2.0 * PI * k / (S2O_DECIM_TAPS - 1)) + 0.08 * cos(4.0 * PI * k / (S2O_DECIM_TAPS - 1));
        d[k] = ideal * w;
        sum += d[k];
    }
    for (int k = 0; k < S2O_DECIM_TAPS; k++) h[k] = (float)(d[k] / sum);
}

void s2o_decimate4(const float *x, size_t n_out, const float *h, float *out) {
    for (size_t n = 0; n < n_out; n++) {
        float acc = 0.0f;
        for (int k = 0; k < S2O_DECIM_TAPS; k++) acc = acc + h[k] * x[4 * n + (size_t)k];   /* x[0] is sample 4n-62 of the stream */
        out[n] = acc;
    }
}

/* ------------------------------------------------------------------ dsp_filters.rs */

/* dsp_filters.rs:25-45 (LP1), :60-80 (HP1), :99-130 (LP2), :149-180 (HP2), :199-230 (BP2).  No
 * `fma` feature switch in that file: every operation is rounded separately, in Rust's evaluation
 * order.  sin/cos/tan are Rust f32::sin/cos/tan = the host libm's sinf/cosf/tanf. */
float s2o_dsp_filter_process(int kind, float *x1p, float *x2p, float *y1p, float *y2p,
                             uint32_t sample_rate_u, float cutoff_freq, float damping_factor, float input) {
    const float PI = 3.14159274101257324f;
    float sample_rate = (float)sample_rate_u;
    float theta_cutoff = 2.0f * PI * cutoff_freq / sample_rate;
    float x1 = *x1p, x2 = *x2p, y1 = *y1p, y2 = *y2p;
    float x = input, y;
    if (kind == S2O_FILT_LP1 || kind == S2O_FILT_HP1) {
        float gamma = cosf(theta_cutoff) / (1.0f + sinf(theta_cutoff));
        if (kind == S2O_FILT_LP1) {
            float alpha = (1.0f - gamma) / 2.0f;
            y = alpha * (x + x1) + gamma * y1;
        } else {
            float alpha = (1.0f + gamma) / 2.0f;
            y = alpha * (x - x1) + gamma * y1;
        }
        *x1p = x; *y1p = y;
        return y;
    }
    if (kind >= S2O_FILT_SVF_LP) {
        /* BUILD-DEFINED (self-oracle): trapezoidal state-variable filter.  The
         * definition is this sequence of separately rounded f32 operations; x1/x2 hold the two
         * integrator states. */
        float q = damping_factor;
        float fc = fminf(cutoff_freq, 0.49f * sample_rate);     /* below Nyquist: g > 0, unconditionally stable */
        float g = tanf(PI * fc / sample_rate);
        float k = 1.0f / q;
        float a1 = 1.0f / (1.0f + g * (g + k));
        float a2 = g * a1;
        float a3 = g * a2;
        float v3 = x - x2;
        float v1 = a1 * x1 + a2 * v3;
        float v2 = x2 + a2 * x1 + a3 * v3;
        *x1p = 2.0f * v1 - x1;
        *x2p = 2.0f * v2 - x2;
        if (kind == S2O_FILT_SVF_LP) return v2;
        if (kind == S2O_FILT_SVF_BP) return v1;
        return x - k * v1 - v2;
    }
    if (kind == S2O_FILT_BP2) {                                  /* theta_center, quality_factor */
        float quality_factor = damping_factor;
        float tq = tanf(theta_cutoff / (2.0f * quality_factor));
        float beta = (1.0f / 2.0f) * ((1.0f - tq) / (1.0f + tq));
        float gamma = (1.0f / 2.0f + beta) * cosf(theta_cutoff);
        float alpha = (1.0f / 2.0f - beta) / 2.0f;
        y = 2.0f * (alpha * (x - x2) + gamma * y1 - beta * y2);
        *x2p = x1; *x1p = x; *y2p = y1; *y1p = y;
        return y;
    }
    float s = sinf(theta_cutoff);
    float beta = (1.0f / 2.0f) * ((1.0f - damping_factor / 2.0f * s) / (1.0f + damping_factor / 2.0f * s));
    float gamma = (1.0f / 2.0f + beta) * cosf(theta_cutoff);
    if (kind == S2O_FILT_LP2) {
        float alpha = (1.0f / 2.0f + beta - gamma) / 4.0f;
        y = 2.0f * (alpha * (x + 2.0f * x1 + x2) + gamma * y1 - beta * y2);
    } else {
        float alpha = (1.0f / 2.0f + beta + gamma) / 4.0f;
        y = 2.0f * (alpha * (x - 2.0f * x1 + x2) + gamma * y1 - beta * y2);
    }
    *x2p = x1; *x1p = x; *y2p = y1; *y1p = y;
    return y;
}

static float layer_filter(const s2o_layer_cfg *c, s2o_layer_state *st, uint32_t sr, float freq, float input) {
    if (c->lpf_kind == S2O_FILT_ONEPOLE) return s2o_lpf_process(&st->lpf_last, sr, freq, input);
    return s2o_dsp_filter_process(c->lpf_kind, &st->x1, &st->x2, &st->y1, &st->y2, sr, freq,
                                  c->lpf_kind >= S2O_FILT_BP2 ? c->lpf_q : c->lpf_damping, input);
}

/* ------------------------------------------------------------------ oscillators.rs */

/* oscillators.rs:377-381  accum_phase */
static float accum_phase(float phase, float period) {
    float phase_delta = 1.0f / period;
    return fmodf(phase + phase_delta, 1.0f);
}

/* oscillators.rs:391-400  accum_phase_x16 */
static float accum_phase_x16(float phase0, const float period[16], float phase[16]) {
    float acc = phase0;
    for (int i = 0; i < 16; i++) phase[i] = phase0;
    for (int i = 1; i < 16; i++) { acc = accum_phase(acc, period[i - 1]); phase[i] = acc; }
    acc = accum_phase(acc, period[15]);
    return acc;
}

/* oscillators.rs:207-215 / 217-239 (feature fma): period.mul_add(phase, offset) */
static float phased_offset(float period, float phase, float offset) { return fmaf(period, phase, offset); }

/* oscillators.rs:47-58 / 60-80 */
static float basic_square(float period, float offset) {
    offset = fmodf(offset, period);
    float half_period = period / 2.0f;
    return offset < half_period ? 1.0f : -1.0f;
}
/* oscillators.rs:82-97 / 99-119 */
static float basic_saw(float period, float offset) {
    offset = fmodf(offset, period);
    return line_fma(-2.0f, period, offset, 1.0f);
}
/* oscillators.rs:121-146 / 148-183 (x16 computes both halves and selects: same value) */
static float basic_triangle(float period, float offset) {
    offset = fmodf(offset, period);
    float half_period = period / 2.0f;
    if (offset < half_period) return line_fma(-2.0f, half_period, offset, 1.0f);
    return line_fma(2.0f, half_period, offset - half_period, -1.0f);
}

/* Build-defined alias-suppressed oscillators (the reference only links to the literature, notes.md:32,79):
 * differentiated polynomial waveforms.  s = the naive saw's sample at this phase (oscillators.rs:99-119: 1 at phase 0,
 * falling to -1); the naive shapes are functions of s — saw s, square sign(s), triangle 2|s| - 1 (== oscillators.rs:60-80,
 * 148-183) — and F is each one's integral over s, continuous across the phase wrap: saw s^2 / 2, square |s|, triangle
 * s (|s| - 1).  The output is F's first difference divided by s's step per frame, -2 / period:
 *     y[n] = (-0.5 * period) * (F(s[n]) - F(s[n-1]))
 * every operation a separately rounded f32 one, in this order; before a voice's first frame the memory holds that
 * frame's own F (y[0] = 0).  What the differencing buys: the shapes' discontinuities (in value for saw and square, in
 * slope for the triangle) become ones of a higher derivative of F, so their aliases fall 6 dB per octave faster. */
static float dpw_sample(int kind, float period, float phase, s2o_layer_state *st) {
    float offset = phased_offset(period, phase, 0.0f);
    float x = fmodf(offset, period);
    float s = line_fma(-2.0f, period, x, 1.0f);
    float F;
    if (kind == S2O_OSC_DPW_SAW) F = 0.5f * (s * s);
    else if (kind == S2O_OSC_DPW_SQUARE) F = fabsf(s);
    else F = s * (fabsf(s) - 1.0f);
    float z = st->has_z ? st->dpw_z : F;
    float c = -0.5f * period;
    st->has_z = 1; st->dpw_z = F;
    return c * (F - z);
}

static float osc_sample(int kind, float period, float phase, int x16, int *panicked) {
    float offset = phased_offset(period, phase, 0.0f);
    switch (kind) {
    case S2O_OSC_SQUARE: return basic_square(period, offset);
    case S2O_OSC_SAW: return basic_saw(period, offset);
    case S2O_OSC_TRIANGLE: return basic_triangle(period, offset);
    default: {
        const float *T = s2o_sin_table();
        if (x16) {   /* oscillators.rs:191-199 -> lookup.rs:187-199; evaluated lane-wise */
            float v[16], r[16], o[16];
            for (int i = 0; i < 16; i++) { v[i] = offset; r[i] = period; }
            s2o_table_lookup_periodic_x16(T, 1024, v, r, o);
            return o[0];
        }
        /* oscillators.rs:185-189 -> lookup.rs:179-185 */
        return s2o_table_lookup_exclusive(T, 1024, fmodf(offset, period), period, panicked);
    }
    }
}

/* ------------------------------------------------------------------ process.rs */

/* process.rs:231-250 */
void s2o_modulate_freq_unipolar_x16(float freq, const float mod[16], float amount, float out[16]) {
    for (int i = 0; i < 16; i++) {
        float modulation_amount_ = mod[i] * amount;
        out[i] = s2o_sleef_powf(2.0f, modulation_amount_) * freq;
    }
}
/* process.rs:221-229 */
float s2o_modulate_freq_unipolar(float freq, float mod, float amount) {
    float modulation_amount_ = mod * amount;
    volatile float two = 2.0f;   /* keep the call a real libm powf, not a folded exp2f */
    return powf(two, modulation_amount_) * freq;
}

typedef struct { float periods[16]; float lpf_freqs[16]; float gains[16]; } plan_x16;

/* process.rs:137-174 prepare_frame_x16 (+ :191-219 sample_envelope_x16 / offsets_x16) */
static void prepare_frame_x16(const s2o_layer_cfg *c, float pitch, uint32_t sr, uint32_t offset,
                              int has_release, uint32_t release_offset, plan_x16 *p) {
    uint32_t offsets[16];
    for (int i = 0; i < 16; i++) offsets[i] = offset + (uint32_t)i;          /* wrapping simd add */
    float mod_env[16], osc_freqs[16];
    s2o_adsr_x16(s2o_ms_as_samples(c->amp_env.attack_ms, sr), s2o_ms_as_samples(c->amp_env.decay_ms, sr),
                 c->amp_env.sustain, s2o_ms_as_samples(c->amp_env.release_ms, sr),
                 offsets, has_release, release_offset, p->gains);
    s2o_adsr_x16(s2o_ms_as_samples(c->mod_env.attack_ms, sr), s2o_ms_as_samples(c->mod_env.decay_ms, sr),
                 c->mod_env.sustain, s2o_ms_as_samples(c->mod_env.release_ms, sr),
                 offsets, has_release, release_offset, mod_env);
    s2o_modulate_freq_unipolar_x16(pitch, mod_env, c->mod_env_to_osc_freq, osc_freqs);
    s2o_modulate_freq_unipolar_x16(c->lpf_freq, mod_env, c->mod_env_to_lpf_freq, p->lpf_freqs);
    for (int i = 0; i < 16; i++) p->periods[i] = hz_as_samples(osc_freqs[i], sr);
}

/* process.rs:306-379 sample_voice_x16 */
static void sample_voice_x16(const s2o_layer_cfg *c, const plan_x16 *p, s2o_layer_state *st,
                             uint32_t sr, uint32_t offset, float out[16], int *panicked) {
    float init_phase = st->has_phase ? st->phase_accum : 0.0f;               /* oscillators.rs:483 */
    float phase[16];
    float next = accum_phase_x16(init_phase, p->periods, phase);
    float samples[16];
    for (int i = 0; i < 16; i++) {
        float osc = c->osc_kind >= S2O_OSC_DPW_SAW ? dpw_sample(c->osc_kind, p->periods[i], phase[i], st)
                                                   : osc_sample(c->osc_kind, p->periods[i], phase[i], 1, panicked);
        float osc_plus = osc + c->osc_gain;                                  /* :342-345  ADD */
        float off_f = (float)(offset + (uint32_t)i);                         /* :347-348 */
        float noise = s2o_hash_noise(st->seed, off_f);
        float noise_plus = noise + c->noise;                                 /* :353-356  ADD */
        samples[i] = osc_plus + noise_plus;                                  /* :358 */
    }
    st->has_phase = 1; st->phase_accum = next;                               /* oscillators.rs:492 */
    for (int i = 0; i < 16; i++)                                             /* :363-371 sequential */
        samples[i] = layer_filter(c, st, sr, p->lpf_freqs[i], samples[i]);
    for (int i = 0; i < 16; i++) out[i] = samples[i] * p->gains[i];          /* :373-376 */
}

/* process.rs:101-135 + 252-304: scalar prepare_frame + sample_voice */
static float process_layer(const s2o_layer_cfg *c, s2o_layer_state *st, float pitch, uint32_t sr,
                           uint32_t offset, int has_release, uint32_t release_offset, int *panicked) {
    float amp = s2o_adsr_scalar(s2o_ms_as_samples(c->amp_env.attack_ms, sr), s2o_ms_as_samples(c->amp_env.decay_ms, sr),
                                c->amp_env.sustain, s2o_ms_as_samples(c->amp_env.release_ms, sr),
                                offset, has_release, release_offset);
    float mod = s2o_adsr_scalar(s2o_ms_as_samples(c->mod_env.attack_ms, sr), s2o_ms_as_samples(c->mod_env.decay_ms, sr),
                                c->mod_env.sustain, s2o_ms_as_samples(c->mod_env.release_ms, sr),
                                offset, has_release, release_offset);
    float osc_freq = s2o_modulate_freq_unipolar(pitch, mod, c->mod_env_to_osc_freq);
    float lpf_freq = s2o_modulate_freq_unipolar(c->lpf_freq, mod, c->mod_env_to_lpf_freq);
    float period = hz_as_samples(osc_freq, sr);

    float phase = st->has_phase ? st->phase_accum : 0.0f;                    /* oscillators.rs:461 */
    float osc = c->osc_kind >= S2O_OSC_DPW_SAW ? dpw_sample(c->osc_kind, period, phase, st)
                                               : osc_sample(c->osc_kind, period, phase, 0, panicked);
    st->has_phase = 1; st->phase_accum = accum_phase(phase, period);         /* oscillators.rs:469 */
    float osc_sample_ = osc * c->osc_gain;                                   /* :287  MULTIPLY */
    float noise = s2o_hash_noise(st->seed, (float)offset);                   /* :289-291 */
    float noise_sample = noise * c->noise;                                   /* :292  MULTIPLY */
    float sample = osc_sample_ + noise_sample;
    sample = layer_filter(c, st, sr, lpf_freq, sample);
    return sample * amp;                                                     /* :302 */
}

/* process.rs:14-49 process_layer_buf_simd, :51-74 process_layer_buf_sisd.  Returns 0, or
 * -1 where the reference panics (`checked_add(..).expect("overflow")`, :36,:71). */
int s2o_process_layer_buf_simd(const s2o_layer_cfg *cfg, s2o_layer_state *st, float pitch,
                               uint32_t sr, uint32_t offset, int has_release,
                               uint32_t release_offset, float *buf, size_t len) {
    int panicked = 0;
    size_t i = 0;
    for (; i + 16 <= len; i += 16) {
        plan_x16 p;
        prepare_frame_x16(cfg, pitch, sr, offset, has_release, release_offset, &p);
        sample_voice_x16(cfg, &p, st, sr, offset, buf + i, &panicked);
        if (offset > 0xffffffffu - 16u) return -1;
        offset += 16;
    }
    for (; i < len; i++) {
        buf[i] = process_layer(cfg, st, pitch, sr, offset, has_release, release_offset, &panicked);
        if (offset == 0xffffffffu) return -1;
        offset += 1;
    }
    return panicked ? -1 : 0;
}

/* ------------------------------------------------------------------ synth.rs */

/* synth.rs:208-212 */
float s2o_note_to_pitch(uint8_t note_u) {
    float note = (float)note_u;
    volatile float two = 2.0f;
    return 440.0f * powf(two, (note - 69.0f) / 12.0f);
}

/* synth.rs:125-152 */
s2o_layer_cfg s2o_default_config(void) {
    s2o_layer_cfg c;
    c.osc_kind = S2O_OSC_SAW; c.osc_gain = 1.0f;
    c.noise = 0.0f;
    c.lpf_freq = 200.0f;
    c.amp_env.attack_ms = 100.0f; c.amp_env.decay_ms = 100.0f; c.amp_env.sustain = 0.5f; c.amp_env.release_ms = 100.0f;
    c.mod_env.attack_ms = 0.0f; c.mod_env.decay_ms = 200.0f; c.mod_env.sustain = 0.0f; c.mod_env.release_ms = 0.0f;
    c.mod_env_to_osc_freq = 0.0f;
    c.mod_env_to_lpf_freq = 10.0f;
    c.lpf_kind = S2O_FILT_ONEPOLE;
    c.lpf_damping = 1.41421354f;     /* "sqrt(2) is neutral", dsp_filters.rs:95 */
    c.lpf_q = 3.0f;                  /* "3 is neutral", dsp_filters.rs:194 */
    return c;
}

s2o_synth *s2o_synth_new(uint32_t num_voices) {
    s2o_synth *s = (s2o_synth *)calloc(1, sizeof *s);
    s->config = s2o_default_config();
    s->num_voices = num_voices;
    s->voices = (s2o_voice *)calloc(num_voices ? num_voices : 1, sizeof(s2o_voice)); /* Voice::default(), :41-51 */
    return s;
}
void s2o_synth_free(s2o_synth *s) { if (s) { free(s->voices); free(s->bank); free(s); } }

void s2o_set_bank(s2o_synth *s, const s2o_layer_cfg *cfgs, uint32_t n) {
    free(s->bank);
    s->bank = NULL; s->bank_size = 0;
    if (n) {
        s->bank = (s2o_layer_cfg *)malloc(sizeof(s2o_layer_cfg) * n);
        memcpy(s->bank, cfgs, sizeof(s2o_layer_cfg) * n);
        s->bank_size = n;
    }
    if (s->current_program >= n) s->current_program = 0;
}
void s2o_program_change(s2o_synth *s, uint32_t program) { s->current_program = program; }

static const s2o_layer_cfg *voice_config(const s2o_synth *s, const s2o_voice *v) {
    if (!s->bank_size) return &s->config;
    return &s->bank[v->program < s->bank_size ? v->program : 0];
}

/* synth.rs:101-120: the voice with the greatest current_frame_offset (None = u32::MAX),
 * first such index on ties (strict `>`). */
uint32_t s2o_next_voice_index(const s2o_synth *s) {
    uint32_t oldest_index = 0, oldest_off = 0;
    for (uint32_t i = 0; i < s->num_voices; i++) {
        uint32_t this_off = s->voices[i].has_current ? s->voices[i].current_frame_offset : 0xffffffffu;
        if (i == 0) { oldest_off = this_off; oldest_index = 0; }
        else if (this_off > oldest_off) { oldest_off = this_off; oldest_index = i; }
    }
    return oldest_index;
}

/* synth.rs:61-70 */
void s2o_note_on(s2o_synth *s, uint8_t note, float velocity) {
    s2o_voice *v = &s->voices[s2o_next_voice_index(s)];
    memset(v, 0, sizeof *v);                      /* state: st::Layer::default() */
    v->note = note; v->velocity = velocity;
    v->has_current = 1; v->current_frame_offset = 0;
    v->has_release = 0;
    v->program = s->current_program;
}

/* synth.rs:72-96: LAST index whose note matches and which is active */
void s2o_note_off(s2o_synth *s, uint8_t note) {
    int64_t found = -1;
    for (uint32_t i = 0; i < s->num_voices; i++) {
        const s2o_voice *v = &s->voices[i];
        if (v->note == note && v->has_current && !v->has_release) found = i;
    }
    if (found < 0) return;
    s2o_voice *v = &s->voices[found];
    if (!v->has_release) { v->has_release = 1; v->release_frame_offset = v->current_frame_offset; }
    else s->double_release++;                     /* unreachable, as in the reference (:74-78) */
}

/* One voice through Synth::sample's chunking (synth.rs:158-168, 171-198): 16-frame
 * chunks, then one tail chunk; offset advances by saturating_add per chunk (:197). */
static void render_one_voice(s2o_synth *s, s2o_voice *v, float *row, size_t frames, uint32_t sr) {
    if (!v->has_current) { for (size_t i = 0; i < frames; i++) row[i] = 0.0f; return; }
    float pitch = s2o_note_to_pitch(v->note);                                /* :179 */
    size_t done = 0;
    while (done < frames) {
        size_t n = frames - done < 16 ? frames - done : 16;
        float buf[16] = {0};
        if (s2o_process_layer_buf_simd(voice_config(s, v), &v->state, pitch, sr, v->current_frame_offset,
                                       v->has_release, v->release_frame_offset, buf, n) != 0)
            s->panicked = 1;
        memcpy(row + done, buf, n * sizeof(float));
        uint32_t o = v->current_frame_offset;
        v->current_frame_offset = (o > 0xffffffffu - (uint32_t)n) ? 0xffffffffu : o + (uint32_t)n; /* :197 */
        done += n;
    }
}

void s2o_render_voices(s2o_synth *s, float *per_voice, size_t frames, uint32_t sr) {
    for (uint32_t v = 0; v < s->num_voices; v++)
        render_one_voice(s, &s->voices[v], per_voice + (size_t)v * frames, frames, sr);
}

/* ------------------------------------------------------------------ multi-threaded drivers
 * (timing legs and the full-size parity tests).  Voices are independent until the add (synth.rs:195), so they are
 * sharded over a PERSISTENT pool of worker threads: created on first use, parked on a condition variable between
 * jobs, never joined per buffer; per-thread row and accumulator buffers are kept and grown, not allocated per call. */
typedef void (*pool_fn)(void *ctx, int t, int n_threads);
static struct {
    pthread_mutex_t mu; pthread_cond_t go, done;
    pthread_t *th; int n_workers;            /* workers 1..n (the caller is thread 0) */
    pool_fn fn; void *ctx; int n_active; unsigned long long gen; int remaining;
    float **scratch; size_t *scratch_cap;    /* per thread: grown on demand, kept */
    int scratch_n;
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, 0, NULL, NULL, 0, 0, 0, NULL, NULL, 0 };

/* timing legs: worker t runs on CPU g_pin[t] (one per physical core, chosen by the caller); n == 0: unpinned */
static int g_pin[256], g_pin_n = 0;
static unsigned long long g_pin_gen = 0;
void s2o_pool_pin(const int *cpus, int n) {
    pthread_mutex_lock(&g_pool.mu);
    g_pin_n = n < 0 ? 0 : (n > 256 ? 256 : n);
    for (int i = 0; i < g_pin_n; i++) g_pin[i] = cpus[i];
    g_pin_gen++;
    pthread_mutex_unlock(&g_pool.mu);
}

static void *pool_worker(void *arg) {
    const int me = (int)(intptr_t)arg;
    unsigned long long seen = 0, pin_seen = 0;
    pthread_mutex_lock(&g_pool.mu);
    for (;;) {
        /* (jobs follow each other within microseconds in the event-driven renderer — one per 16-frame stretch — so a
         * worker looks for the next one for a little while before it sleeps; the pool's threads are pinned one per core
         * in the timing legs) */
        if (g_pool.gen == seen) {
            pthread_mutex_unlock(&g_pool.mu);
            for (int spin = 0; spin < 20000 && __atomic_load_n(&g_pool.gen, __ATOMIC_ACQUIRE) == seen; spin++) __builtin_ia32_pause();
            pthread_mutex_lock(&g_pool.mu);
        }
        while (g_pool.gen == seen) pthread_cond_wait(&g_pool.go, &g_pool.mu);
        seen = g_pool.gen;
        if (pin_seen != g_pin_gen) {
            pin_seen = g_pin_gen;
            cpu_set_t set; CPU_ZERO(&set);
            if (me < g_pin_n) CPU_SET(g_pin[me], &set);
            else for (int c = 0; c < CPU_SETSIZE; c++) CPU_SET(c, &set);
            pthread_setaffinity_np(pthread_self(), sizeof set, &set);       /* (an error leaves the thread where it is) */
        }
        if (me < g_pool.n_active) {
            pool_fn fn = g_pool.fn; void *ctx = g_pool.ctx; const int n = g_pool.n_active;
            pthread_mutex_unlock(&g_pool.mu);
            fn(ctx, me, n);
            pthread_mutex_lock(&g_pool.mu);
            if (--g_pool.remaining == 0) pthread_cond_signal(&g_pool.done);
        }
    }
    return NULL;
}

static float *pool_scratch(int t, size_t floats) {
    if (g_pool.scratch_cap[t] < floats) {
        free(g_pool.scratch[t]);
        g_pool.scratch[t] = (float *)malloc(sizeof(float) * floats);
        g_pool.scratch_cap[t] = floats;
    }
    return g_pool.scratch[t];
}

/* runs fn(ctx, t, threads) for t = 0..threads-1, the caller being thread 0; returns when all are done */
static void pool_run(int threads, pool_fn fn, void *ctx) {
    if (threads < 1) threads = 1;
    pthread_mutex_lock(&g_pool.mu);
    if (g_pool.scratch_n < threads) {
        g_pool.scratch = (float **)realloc(g_pool.scratch, sizeof(float *) * threads);
        g_pool.scratch_cap = (size_t *)realloc(g_pool.scratch_cap, sizeof(size_t) * threads);
        for (int t = g_pool.scratch_n; t < threads; t++) { g_pool.scratch[t] = NULL; g_pool.scratch_cap[t] = 0; }
        g_pool.scratch_n = threads;
    }
    if (g_pool.n_workers < threads - 1) {
        g_pool.th = (pthread_t *)realloc(g_pool.th, sizeof(pthread_t) * (threads - 1));
        for (int t = g_pool.n_workers; t < threads - 1; t++)
            pthread_create(&g_pool.th[t], NULL, pool_worker, (void *)(intptr_t)(t + 1));
        g_pool.n_workers = threads - 1;
    }
    g_pool.fn = fn; g_pool.ctx = ctx; g_pool.n_active = threads; g_pool.remaining = threads - 1;
    g_pool.gen++;
    pthread_cond_broadcast(&g_pool.go);
    pthread_mutex_unlock(&g_pool.mu);
    fn(ctx, 0, threads);
    for (int spin = 0; spin < 20000 && __atomic_load_n(&g_pool.remaining, __ATOMIC_ACQUIRE) != 0; spin++) __builtin_ia32_pause();
    pthread_mutex_lock(&g_pool.mu);
    while (g_pool.remaining) pthread_cond_wait(&g_pool.done, &g_pool.mu);
    pthread_mutex_unlock(&g_pool.mu);
}

typedef struct { s2o_synth *s; float *pv; size_t frames, stride, col0; uint32_t sr; float *acc; } mt_job;

/* thread t of n takes the 64-voice blocks b with b % n == t: voices that were started together (and are in the same,
 * possibly dearer, envelope stages: pow(2, 0) is an early exit) sit next to each other in the pool, so contiguous
 * ranges would hand some threads all of them */
#define MT_BLOCK 64u
static void mt_render(void *ctx, int t, int n) {
    mt_job *j = (mt_job *)ctx;
    const uint32_t nv = j->s->num_voices;
    for (uint32_t b0 = (uint32_t)t * MT_BLOCK; b0 < nv; b0 += (uint32_t)n * MT_BLOCK)
        for (uint32_t v = b0; v < b0 + MT_BLOCK && v < nv; v++)
            render_one_voice(j->s, &j->s->voices[v], j->pv + (size_t)v * j->stride + j->col0, j->frames, j->sr);
}
void s2o_render_voices_mt(s2o_synth *s, float *per_voice, size_t frames, uint32_t sr, int threads) {
    mt_job j = { s, per_voice, frames, frames, 0, sr, NULL };
    pool_run(threads, mt_render, &j);
}

/* thread t's voices (the same blocks) rendered and added in index order into its accumulator acc[t][frames_stride] at column col0 */
static void mt_sample(void *ctx, int t, int n) {
    mt_job *j = (mt_job *)ctx;
    const uint32_t nv = j->s->num_voices;
    float row[16];
    float *acc = j->acc + (size_t)t * j->stride + j->col0;
    for (size_t i = 0; i < j->frames; i++) acc[i] = 0.0f;
    for (uint32_t b0 = (uint32_t)t * MT_BLOCK; b0 < nv; b0 += (uint32_t)n * MT_BLOCK)
        for (uint32_t v = b0; v < b0 + MT_BLOCK && v < nv; v++) {
            if (!j->s->voices[v].has_current) continue;
            for (size_t d = 0; d < j->frames; d += 16) {
                const size_t m = j->frames - d < 16 ? j->frames - d : 16;
                render_one_voice(j->s, &j->s->voices[v], row, m, j->sr);
                for (size_t i = 0; i < m; i++) acc[d + i] += row[i];
            }
        }
}
static void sum_thread_partials(const float *acc, int threads, size_t stride, size_t frames, float *buffer) {
    for (size_t i = 0; i < frames; i++) {
        float a = 0.0f;
        for (int t = 0; t < threads; t++) a += acc[(size_t)t * stride + i];
        buffer[i] = a;
    }
}
void s2o_sample_mt(s2o_synth *s, float *buffer, size_t frames, uint32_t sr, int threads) {
    if (threads < 1) threads = 1;
    pthread_mutex_lock(&g_pool.mu);                   /* (the accumulators: thread 0's scratch, kept between calls) */
    if (g_pool.scratch_n < 1) { g_pool.scratch = (float **)calloc(1, sizeof(float *)); g_pool.scratch_cap = (size_t *)calloc(1, sizeof(size_t)); g_pool.scratch_n = 1; }
    float *acc = pool_scratch(0, frames * (size_t)threads);
    pthread_mutex_unlock(&g_pool.mu);
    mt_job j = { s, NULL, frames, frames, 0, sr, acc };
    pool_run(threads, mt_sample, &j);
    sum_thread_partials(acc, threads, frames, frames, buffer);
}

/* One buffer in the reference caller's pattern (s2_bin/src/main.rs:138-147): for every 16-frame chunk, the note
 * events stamped with that frame are applied (in order), then Synth::sample renders the chunk.  Events: the layout of
 * libs2r's s2r_note_event {u8 kind (0 off, 1 on, 2 program change), u8 note, u16 frame, f32 velocity}, sorted by
 * frame; frames must be multiples of 16.  per_voice != NULL: every voice's frames, [num_voices][frames];
 * mix != NULL: the voices of each thread added in index order, thread partials in thread order (timing leg). */
static double g_events_seconds[2];       /* [0] inside note_on / note_off (the reference's O(V) scans), [1] rendering */
void s2o_events_seconds(double out[2], int reset) {
    out[0] = g_events_seconds[0]; out[1] = g_events_seconds[1];
    if (reset) g_events_seconds[0] = g_events_seconds[1] = 0.0;
}
void s2o_render_events_mt(s2o_synth *s, const s2o_note_event *ev, size_t n_ev, float *per_voice, float *mix,
                          size_t frames, uint32_t sr, int threads) {
    if (threads < 1) threads = 1;
    float *acc = NULL;
    if (mix) {
        pthread_mutex_lock(&g_pool.mu);
        if (g_pool.scratch_n < 1) { g_pool.scratch = (float **)calloc(1, sizeof(float *)); g_pool.scratch_cap = (size_t *)calloc(1, sizeof(size_t)); g_pool.scratch_n = 1; }
        acc = pool_scratch(0, frames * (size_t)threads);
        pthread_mutex_unlock(&g_pool.mu);
    }
    size_t k = 0, c = 0;
    struct timespec ta, tb, tc;
    while (c < frames) {
        clock_gettime(CLOCK_MONOTONIC, &ta);
        for (; k < n_ev && ev[k].frame <= c; k++) {          /* apply_all_midi_messages, main.rs:170-187 */
            if (ev[k].kind == 1) s2o_note_on(s, ev[k].note, ev[k].velocity);
            else if (ev[k].kind == 0) s2o_note_off(s, ev[k].note);
            else s2o_program_change(s, ev[k].note);
        }
        clock_gettime(CLOCK_MONOTONIC, &tb);
        /* the stretch up to the next event (a whole number of 16-frame sample() calls: the chunking restarts per
         * call, synth.rs:158, and 16-frame calls of full chunks compose to one longer call of full chunks) */
        size_t next = frames;
        if (k < n_ev && ev[k].frame < frames) next = ev[k].frame;
        const size_t n = next - c;
        if (per_voice) { mt_job j = { s, per_voice, n, frames, c, sr, NULL }; pool_run(threads, mt_render, &j); }
        else { mt_job j = { s, NULL, n, frames, c, sr, acc }; pool_run(threads, mt_sample, &j); }
        clock_gettime(CLOCK_MONOTONIC, &tc);
        g_events_seconds[0] += (double)(tb.tv_sec - ta.tv_sec) + 1e-9 * (double)(tb.tv_nsec - ta.tv_nsec);
        g_events_seconds[1] += (double)(tc.tv_sec - tb.tv_sec) + 1e-9 * (double)(tc.tv_nsec - tb.tv_nsec);
        c = next;
    }
    if (mix) sum_thread_partials(acc, threads, frames, frames, mix);
}

/* synth.rs:171-203: accum = 0; for voices in index order (started ones only) accum += buf */
void s2o_mix_sequential(const float *per_voice, uint32_t voices, size_t frames, float *out) {
    for (size_t i = 0; i < frames; i++) {
        float accum = 0.0f;
        for (uint32_t v = 0; v < voices; v++) accum += per_voice[(size_t)v * frames + i];
        out[i] = accum;
    }
}

void s2o_sample(s2o_synth *s, float *buffer, size_t frames, uint32_t sr) {
    /* Rendering voice-by-voice over the whole buffer is the same computation as the
     * reference's chunk-outer / voice-inner loop: voices do not interact before the add,
     * and the per-frame additions happen in the same (voice index) order.  Never-started
     * voices are skipped by the reference (:178); adding their +0.0 rows is bit-identical
     * because accum starts at +0.0 (x + 0.0 == x for every x that can occur, incl. -0.0
     * after the first add: (+0.0) + (-0.0) = +0.0). */
    float *pv = (float *)malloc(sizeof(float) * (size_t)s->num_voices * (frames ? frames : 1));
    s2o_render_voices(s, pv, frames, sr);
    s2o_mix_sequential(pv, s->num_voices, frames, buffer);
    free(pv);
}

/* ------------------------------------------------------------------ GPU mix tree */

/* 16 consecutive voices added in index order (the reference's own order inside the group) */
static float group16(const float *per_voice, uint32_t voices, size_t frames, uint32_t v0, size_t i) {
    float acc = (v0 < voices) ? per_voice[(size_t)v0 * frames + i] : 0.0f;
    for (uint32_t k = 1; k < 16; k++) acc += (v0 + k < voices) ? per_voice[(size_t)(v0 + k) * frames + i] : 0.0f;
    return acc;
}

/* a workgroup's `block_voices` voices: its 16-voice group sums added in group order */
static float block_partial(const float *per_voice, uint32_t voices, size_t frames, uint32_t b, uint32_t block_voices, size_t i) {
    uint32_t v0 = b * block_voices;
    float acc = group16(per_voice, voices, frames, v0, i);
    for (uint32_t g = 1; g < block_voices / 16; g++) acc += group16(per_voice, voices, frames, v0 + 16 * g, i);
    return acc;
}

/* blocks [b0, b1): runs of 16 consecutive blocks added sequentially, then the run sums */
static float blocks_sum(const float *per_voice, uint32_t voices, size_t frames, uint32_t b0, uint32_t b1, uint32_t block_voices, size_t i) {
    float total = 0.0f;
    for (uint32_t r0 = b0; r0 < b1; r0 += 16) {
        uint32_t r1 = r0 + 16 < b1 ? r0 + 16 : b1;
        float acc = block_partial(per_voice, voices, frames, r0, block_voices, i);
        for (uint32_t b = r0 + 1; b < r1; b++) acc += block_partial(per_voice, voices, frames, b, block_voices, i);
        total = (r0 == b0) ? acc : total + acc;
    }
    return total;
}

/* The same tree a row at a time (every addition of the scalar form above, in its order, on whole rows of `frames`
 * values): the voice rows are read once, front to back, instead of once per frame with a stride of a whole row — what
 * makes the 65 536-voice checks affordable.  tests/test_oracle_known_answers.py holds the two forms against each other. */
static void rows_blocks_sum(const float *per_voice, uint32_t voices, size_t frames, uint32_t b0, uint32_t b1, uint32_t block_voices,
                            float *total, float *run, float *blk, float *grp) {
    for (uint32_t r0 = b0; r0 < b1; r0 += 16) {
        const uint32_t r1 = r0 + 16 < b1 ? r0 + 16 : b1;
        for (uint32_t b = r0; b < r1; b++) {
            const uint32_t vb = b * block_voices;
            for (uint32_t g = 0; g < block_voices / 16; g++) {
                const uint32_t v0 = vb + 16 * g;
                for (size_t i = 0; i < frames; i++) grp[i] = (v0 < voices) ? per_voice[(size_t)v0 * frames + i] : 0.0f;
                for (uint32_t k = 1; k < 16; k++) {
                    if (v0 + k < voices) { const float *row = per_voice + (size_t)(v0 + k) * frames; for (size_t i = 0; i < frames; i++) grp[i] += row[i]; }
                    else for (size_t i = 0; i < frames; i++) grp[i] += 0.0f;
                }
                if (g == 0) memcpy(blk, grp, frames * sizeof(float));
                else for (size_t i = 0; i < frames; i++) blk[i] += grp[i];
            }
            if (b == r0) memcpy(run, blk, frames * sizeof(float));
            else for (size_t i = 0; i < frames; i++) run[i] += blk[i];
        }
        if (r0 == b0) memcpy(total, run, frames * sizeof(float));
        else for (size_t i = 0; i < frames; i++) total[i] += run[i];
    }
}

void s2o_mix_tree_partial(const float *per_voice, uint32_t voices, size_t frames, uint32_t block_voices, float *out) {
    const uint32_t nblocks = (voices + block_voices - 1) / block_voices;
    float *tmp = (float *)malloc(sizeof(float) * 3 * (frames ? frames : 1));
    rows_blocks_sum(per_voice, voices, frames, 0, nblocks, block_voices, out, tmp, tmp + frames, tmp + 2 * frames);
    free(tmp);
}

void s2o_mix_tree(const float *per_voice, uint32_t voices, size_t frames, s2o_tree tree, float *out) {
    const uint32_t nblocks = (voices + tree.block_voices - 1) / tree.block_voices;
    const uint32_t groups = tree.groups ? tree.groups : 1;
    const uint32_t per_group = (nblocks + groups - 1) / groups;
    float *tmp = (float *)malloc(sizeof(float) * 4 * (frames ? frames : 1));
    for (size_t i = 0; i < frames; i++) out[i] = 0.0f;                /* accum = splat(0.0), synth.rs:176 */
    for (uint32_t g = 0; g < groups; g++) {
        const uint32_t b0 = g * per_group, b1 = b0 + per_group < nblocks ? b0 + per_group : nblocks;
        if (b0 >= b1) continue;
        rows_blocks_sum(per_voice, voices, frames, b0, b1, tree.block_voices, tmp, tmp + frames, tmp + 2 * frames, tmp + 3 * frames);
        for (size_t i = 0; i < frames; i++) out[i] += tmp[i];
    }
    free(tmp);
}

/* the scalar form (one frame at a time), kept as the statement of the tree the row form is checked against */
void s2o_mix_tree_partial_scalar(const float *per_voice, uint32_t voices, size_t frames, uint32_t block_voices, float *out) {
    uint32_t nblocks = (voices + block_voices - 1) / block_voices;
    for (size_t i = 0; i < frames; i++) out[i] = blocks_sum(per_voice, voices, frames, 0, nblocks, block_voices, i);
}

void s2o_mix_tree_scalar(const float *per_voice, uint32_t voices, size_t frames, s2o_tree tree, float *out) {
    uint32_t nblocks = (voices + tree.block_voices - 1) / tree.block_voices;
    uint32_t groups = tree.groups ? tree.groups : 1;
    uint32_t per_group = (nblocks + groups - 1) / groups;
    for (size_t i = 0; i < frames; i++) {
        float total = 0.0f;                                   /* accum = splat(0.0), synth.rs:176 */
        for (uint32_t g = 0; g < groups; g++) {
            uint32_t b0 = g * per_group, b1 = b0 + per_group < nblocks ? b0 + per_group : nblocks;
            if (b0 >= b1) continue;
            total += blocks_sum(per_voice, voices, frames, b0, b1, tree.block_voices, i);
        }
        out[i] = total;
    }
}
