"""Times one fill of 65536 voices with a DPW oscillator (the per-lane-kind kernel): saw + one-pole, and BASELINE config
[4]'s patch shape (DPW saw + SVF low-pass)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2

V = 65536
for label, osc, filt in (("dpw_saw + one-pole", s2.OSC_DPW_SAW, s2.FILT_ONEPOLE), ("dpw_square + one-pole", s2.OSC_DPW_SQUARE, s2.FILT_ONEPOLE),
                         ("dpw_saw + svf_lp", s2.OSC_DPW_SAW, s2.FILT_SVF_LP)):
    p = s2.default_patch(); p.osc_kind = osc; p.lpf_kind = filt
    s = s2.Synth(V, max_frames=1024)
    s.set_patch(p)
    ev = np.zeros(V, dtype=s2.NOTE_EVENT_DTYPE); ev["kind"] = 1; ev["note"] = (np.arange(V) * 13) % 100 + 20; ev["velocity"] = 1.0
    s.note_events(ev)
    s.set_timing(True)
    buf = np.empty(1024, dtype=np.float32)
    ts = []
    for k in range(24):
        s.sample(buf); ts.append(s.last_render_ms())
    print("%-24s first %.3f ms  settled %.3f ms" % (label, ts[0], np.mean(ts[12:])))
