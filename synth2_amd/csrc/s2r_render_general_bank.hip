// the general render kernel for patch banks: oscillator and filter kind per lane (S2R_OSC_ANY); see s2r_render_general.inc
#define S2R_TU_OSC 15
#include "s2r_render_general.inc"
