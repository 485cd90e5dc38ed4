// s2r_patch.h — the `.synth2` patch text -> s2r_patch.
//
// The reference ships only a stub of the format (example.synth2:1-3: `synth mySynth {` / `}`)
// and no loader; its patch is the literal in Synth::default_config (synth.rs:125-152).
// This loader therefore DEFINES the body grammar (DESIGN.md §.synth2): an empty body is
// exactly default_config(), and every key names one field of static_config::Layer
// (static_config.rs:4-44).  Values are range-checked the way the unit newtypes do
// (units.rs:55-65: Unipolar<N> in [0,N]; Bipolar<N> is taken as [-N,N]).
#pragma once
#include <string>
#include "s2r.h"

// Returns S2R_OK, S2R_ERR_PATCH_SYNTAX or S2R_ERR_PATCH_RANGE; on error `err` explains.
int s2r_parse_patch(const char *text, size_t len, s2r_patch *out, std::string *name, std::string *err);
// Range/finite checks shared with s2r_set_patch.
int s2r_validate_patch(const s2r_patch *p, std::string *err);
