// the general render kernel (dsp_filters.rs kinds, SVF) for the square oscillator; see s2r_render_general.inc
#define S2R_TU_OSC 0
#include "s2r_render_general.inc"
