// the one-pole render kernel for the square oscillator (static_config.rs:26-32); see s2r_render_onepole.inc
#define S2R_TU_OSC 0
#include "s2r_render_onepole.inc"
