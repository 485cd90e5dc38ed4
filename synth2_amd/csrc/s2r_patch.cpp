// s2r_patch.cpp — `.synth2` loader (see s2r_patch.h).
//
// Grammar (whitespace-insensitive; `#` and `//` start a comment that runs to end of line):
//
//   file   := "synth" IDENT "{" { entry } "}"
//   entry  := KEY "=" VALUE [";" | ","]
//   KEY    := osc.kind | osc.gain | noise | lpf.freq | lpf.kind | lpf.damping | lpf.q
//           | amp_env.attack | amp_env.decay | amp_env.sustain | amp_env.release
//           | mod_env.attack | mod_env.decay | mod_env.sustain | mod_env.release
//           | mod_env_to_osc_freq | mod_env_to_lpf_freq
//   VALUE  := number | square | saw | triangle | sine | dpw_saw | dpw_square | dpw_triangle   (kind names only for osc.kind)
//           | onepole | lp1 | hp1 | lp2 | hp2 | bp2 | svf_lp | svf_bp | svf_hp   (only for lpf.kind)
//
// Units follow static_config.rs: *.attack/decay/release in ms (Ms), lpf.freq in Hz,
// gains/levels/sustain Unipolar<1>, the two modulation amounts Bipolar<10>.
#include "s2r_patch.h"
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>

void s2r_default_patch(s2r_patch *p) {      // synth.rs:125-152
    p->osc_kind = S2R_OSC_SAW;
    p->osc_gain = 1.0f;
    p->noise = 0.0f;
    p->lpf_freq = 200.0f;
    p->amp_env = {100.0f, 100.0f, 0.5f, 100.0f};
    p->mod_env = {0.0f, 200.0f, 0.0f, 0.0f};
    p->mod_env_to_osc_freq = 0.0f;
    p->mod_env_to_lpf_freq = 10.0f;
    p->lpf_kind = S2R_FILT_ONEPOLE;          // filters.rs: the only filter the reference wires up
    p->lpf_damping = 1.41421354f;            // dsp_filters.rs:95 "sqrt(2) is neutral"
    p->lpf_q = 3.0f;                         // dsp_filters.rs:194 "3 is neutral"
}

namespace {

struct Lexer {
    const char *s; size_t n, i = 0; int line = 1;
    void skip() {
        for (;;) {
            while (i < n && std::isspace((unsigned char)s[i])) { if (s[i] == '\n') line++; i++; }
            if (i < n && s[i] == '#') { while (i < n && s[i] != '\n') i++; continue; }
            if (i + 1 < n && s[i] == '/' && s[i + 1] == '/') { while (i < n && s[i] != '\n') i++; continue; }
            break;
        }
    }
    bool eof() { skip(); return i >= n; }
    bool punct(char c) { skip(); if (i < n && s[i] == c) { i++; return true; } return false; }
    // identifiers may contain '.', digits and '_'
    bool ident(std::string *out) {
        skip();
        size_t b = i;
        if (i < n && (std::isalpha((unsigned char)s[i]) || s[i] == '_')) {
            while (i < n && (std::isalnum((unsigned char)s[i]) || s[i] == '_' || s[i] == '.')) i++;
            out->assign(s + b, i - b);
            return true;
        }
        return false;
    }
    bool number(float *out) {
        skip();
        std::string tmp(s + i, std::min<size_t>(n - i, 64));
        char *end = nullptr;
        float v = std::strtof(tmp.c_str(), &end);
        if (end == tmp.c_str()) return false;
        i += (size_t)(end - tmp.c_str());
        *out = v;
        return true;
    }
};

bool in_range(float v, float lo, float hi) { return std::isfinite(v) && v >= lo && v <= hi; }

int fail(std::string *err, int code, const std::string &msg) { if (err) *err = msg; return code; }

}  // namespace

int s2r_validate_patch(const s2r_patch *p, std::string *err) {
    if (p->osc_kind < S2R_OSC_SQUARE || p->osc_kind > S2R_OSC_DPW_TRIANGLE) return fail(err, S2R_ERR_PATCH_RANGE, "osc.kind out of range");
    if (!in_range(p->osc_gain, 0.0f, 1.0f)) return fail(err, S2R_ERR_PATCH_RANGE, "osc.gain outside Unipolar<1> [0,1]");
    if (!in_range(p->noise, 0.0f, 1.0f)) return fail(err, S2R_ERR_PATCH_RANGE, "noise outside Unipolar<1> [0,1]");
    if (!std::isfinite(p->lpf_freq) || p->lpf_freq < 0.0f) return fail(err, S2R_ERR_PATCH_RANGE, "lpf.freq must be a finite, non-negative Hz value");
    const s2r_adsr *envs[2] = {&p->amp_env, &p->mod_env};
    const char *names[2] = {"amp_env", "mod_env"};
    for (int k = 0; k < 2; k++) {
        const s2r_adsr *e = envs[k];
        if (!std::isfinite(e->attack_ms) || e->attack_ms < 0 || !std::isfinite(e->decay_ms) || e->decay_ms < 0 ||
            !std::isfinite(e->release_ms) || e->release_ms < 0)
            return fail(err, S2R_ERR_PATCH_RANGE, std::string(names[k]) + " times must be finite, non-negative ms");
        if (!in_range(e->sustain, 0.0f, 1.0f)) return fail(err, S2R_ERR_PATCH_RANGE, std::string(names[k]) + ".sustain outside Unipolar<1> [0,1]");
    }
    if (!in_range(p->mod_env_to_osc_freq, -10.0f, 10.0f)) return fail(err, S2R_ERR_PATCH_RANGE, "mod_env_to_osc_freq outside Bipolar<10> [-10,10]");
    if (!in_range(p->mod_env_to_lpf_freq, -10.0f, 10.0f)) return fail(err, S2R_ERR_PATCH_RANGE, "mod_env_to_lpf_freq outside Bipolar<10> [-10,10]");
    if (p->lpf_kind < S2R_FILT_ONEPOLE || p->lpf_kind > S2R_FILT_SVF_HP) return fail(err, S2R_ERR_PATCH_RANGE, "lpf.kind out of range");
    if (!in_range(p->lpf_damping, 0.0f, 10.0f)) return fail(err, S2R_ERR_PATCH_RANGE, "lpf.damping outside Unipolar<10> [0,10]");
    if (!in_range(p->lpf_q, 0.0f, 10.0f)) return fail(err, S2R_ERR_PATCH_RANGE, "lpf.q outside Unipolar<10> [0,10]");
    return S2R_OK;
}

int s2r_parse_patch(const char *text, size_t len, s2r_patch *out, std::string *name, std::string *err) {
    Lexer lx{text, len};
    s2r_patch p;
    s2r_default_patch(&p);
    std::string tok;
    auto at = [&](const std::string &m) { return "line " + std::to_string(lx.line) + ": " + m; };
    if (!lx.ident(&tok) || tok != "synth") return fail(err, S2R_ERR_PATCH_SYNTAX, at("expected `synth`"));
    if (!lx.ident(&tok)) return fail(err, S2R_ERR_PATCH_SYNTAX, at("expected a synth name"));
    if (name) *name = tok;
    if (!lx.punct('{')) return fail(err, S2R_ERR_PATCH_SYNTAX, at("expected `{`"));
    for (;;) {
        if (lx.punct('}')) break;
        std::string key;
        if (!lx.ident(&key)) return fail(err, S2R_ERR_PATCH_SYNTAX, at("expected a key or `}`"));
        if (!lx.punct('=')) return fail(err, S2R_ERR_PATCH_SYNTAX, at("expected `=` after " + key));
        if (key == "osc.kind") {
            std::string kind; float num;
            if (lx.ident(&kind)) {
                if (kind == "square") p.osc_kind = S2R_OSC_SQUARE;
                else if (kind == "saw") p.osc_kind = S2R_OSC_SAW;
                else if (kind == "triangle") p.osc_kind = S2R_OSC_TRIANGLE;
                else if (kind == "sine") p.osc_kind = S2R_OSC_SINE;
                else if (kind == "dpw_saw") p.osc_kind = S2R_OSC_DPW_SAW;
                else if (kind == "dpw_square") p.osc_kind = S2R_OSC_DPW_SQUARE;
                else if (kind == "dpw_triangle") p.osc_kind = S2R_OSC_DPW_TRIANGLE;
                else return fail(err, S2R_ERR_PATCH_SYNTAX, at("unknown oscillator kind " + kind));
            } else if (lx.number(&num)) {
                p.osc_kind = (int32_t)num;
            } else return fail(err, S2R_ERR_PATCH_SYNTAX, at("expected an oscillator kind"));
        } else if (key == "lpf.kind") {
            std::string kind; float num;
            if (lx.ident(&kind)) {
                if (kind == "onepole") p.lpf_kind = S2R_FILT_ONEPOLE;
                else if (kind == "lp1") p.lpf_kind = S2R_FILT_LP1;
                else if (kind == "hp1") p.lpf_kind = S2R_FILT_HP1;
                else if (kind == "lp2") p.lpf_kind = S2R_FILT_LP2;
                else if (kind == "hp2") p.lpf_kind = S2R_FILT_HP2;
                else if (kind == "bp2") p.lpf_kind = S2R_FILT_BP2;
                else if (kind == "svf_lp") p.lpf_kind = S2R_FILT_SVF_LP;
                else if (kind == "svf_bp") p.lpf_kind = S2R_FILT_SVF_BP;
                else if (kind == "svf_hp") p.lpf_kind = S2R_FILT_SVF_HP;
                else return fail(err, S2R_ERR_PATCH_SYNTAX, at("unknown filter kind " + kind));
            } else if (lx.number(&num)) {
                p.lpf_kind = (int32_t)num;
            } else return fail(err, S2R_ERR_PATCH_SYNTAX, at("expected a filter kind"));
        } else {
            float v;
            if (!lx.number(&v)) return fail(err, S2R_ERR_PATCH_SYNTAX, at("expected a number for " + key));
            if (key == "osc.gain") p.osc_gain = v;
            else if (key == "noise") p.noise = v;
            else if (key == "lpf.freq") p.lpf_freq = v;
            else if (key == "lpf.damping") p.lpf_damping = v;
            else if (key == "lpf.q") p.lpf_q = v;
            else if (key == "amp_env.attack") p.amp_env.attack_ms = v;
            else if (key == "amp_env.decay") p.amp_env.decay_ms = v;
            else if (key == "amp_env.sustain") p.amp_env.sustain = v;
            else if (key == "amp_env.release") p.amp_env.release_ms = v;
            else if (key == "mod_env.attack") p.mod_env.attack_ms = v;
            else if (key == "mod_env.decay") p.mod_env.decay_ms = v;
            else if (key == "mod_env.sustain") p.mod_env.sustain = v;
            else if (key == "mod_env.release") p.mod_env.release_ms = v;
            else if (key == "mod_env_to_osc_freq") p.mod_env_to_osc_freq = v;
            else if (key == "mod_env_to_lpf_freq") p.mod_env_to_lpf_freq = v;
            else return fail(err, S2R_ERR_PATCH_SYNTAX, at("unknown key " + key));
        }
        if (!lx.punct(';')) lx.punct(',');
    }
    if (!lx.eof()) return fail(err, S2R_ERR_PATCH_SYNTAX, at("trailing text after `}`"));
    int rc = s2r_validate_patch(&p, err);
    if (rc != S2R_OK) return rc;
    *out = p;
    return S2R_OK;
}
