// the general render kernel (dsp_filters.rs kinds, SVF) for the triangle oscillator; see s2r_render_general.inc
#define S2R_TU_OSC 2
#include "s2r_render_general.inc"
