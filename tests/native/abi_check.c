/* include/s2r.h consumed from plain C (C99, -pedantic): the header must not need C++, and the entry
 * points that work without a GPU must link and behave.  Built and run by tests/test_abi_c.py. */
#include <stdio.h>
#include <string.h>
#include "s2r.h"

int main(void) {
    s2r_patch p;
    char err[128];
    char json[64];
    float frame[3] = {0.5f, -1.0f, 0.0f};
    s2r_voice_pool *pool;
    s2r_voice_state vs;
    s2r_config cfg;
    s2r_note_event ev;

    if (s2r_abi_version() != S2R_ABI_VERSION) return 1;
    s2r_default_patch(&p);
    if (p.osc_kind != S2R_OSC_SAW || p.lpf_kind != S2R_FILT_ONEPOLE) return 2;
    if (s2r_parse_patch_text("synth x { lpf.kind = bp2 }", 26, &p, err, sizeof err) != S2R_OK || p.lpf_kind != S2R_FILT_BP2) return 3;
    if (s2r_parse_patch_text("synth x {", 9, &p, err, sizeof err) != S2R_ERR_PATCH_SYNTAX || !err[0]) return 4;
    if (s2r_stream_frame_json(frame, 3, json, sizeof json) != strlen("[0.5,-1.0,0.0]") || strcmp(json, "[0.5,-1.0,0.0]")) return 5;
    pool = s2r_voice_pool_create(8);
    if (!pool) return 6;
    if (s2r_voice_pool_note_on(pool, 60, 1.0f) != 0 || s2r_voice_pool_note_on(pool, 60, 1.0f) != 1) return 7;
    if (s2r_voice_pool_note_off(pool, 60) != 1) return 8;                     /* LAST active match, synth.rs:72-96 */
    s2r_voice_pool_advance(pool, 1024);
    if (s2r_voice_pool_query(pool, 1, &vs) != S2R_OK || !vs.released || vs.current_frame_offset != 1024) return 9;
    s2r_voice_pool_destroy(pool);
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (uint32_t)sizeof cfg;
    ev.kind = S2R_PROGRAM_CHANGE; ev.note = 0; ev.frame = 0; ev.velocity = 0.0f;
    (void)ev;
    printf("abi ok: version %u, sizeof(s2r_patch) %u, sizeof(s2r_voice_state) %u, sizeof(s2r_config) %u\n",
           (unsigned)s2r_abi_version(), (unsigned)sizeof(s2r_patch), (unsigned)sizeof(s2r_voice_state), (unsigned)sizeof(s2r_config));
    return 0;
}
