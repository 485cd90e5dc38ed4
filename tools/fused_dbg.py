#!/usr/bin/env python3
"""development aid: the one-launch fill against the three-launch form (S2R_FUSED=0) on the same events, case by case"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth2_amd as s2

def mk(voices, fused):
    os.environ["S2R_FUSED"] = fused
    return s2.Synth(voices, max_frames=1024)

def batch(rng, n, timed, frames=1024, ons_only=False):
    ev = np.zeros(n, dtype=s2.NOTE_EVENT_DTYPE)
    ev["kind"] = 1 if ons_only else rng.randint(0, 2, n)
    ev["note"] = rng.randint(36, 97, n); ev["velocity"] = 1.0
    if timed:
        ev["frame"] = np.sort(rng.randint(0, frames // 16, n)) * 16
    return ev

for voices in (512, 2048):
    for name, n, timed, ons in (("few untimed ons", 100, False, True), ("many untimed ons", 1000, False, True), ("many untimed mixed", 3000, False, False),
                                ("timed mixed", 700, True, False)):
        rng = np.random.RandomState(1)
        a, b = mk(voices, "1"), mk(voices, "0")
        ev = batch(rng, n, timed, ons_only=ons)
        for k in range(2):
            if k == 0:
                a.note_events(ev); b.note_events(ev)
            ga = a.sample(np.empty(1024, dtype=np.float32)).copy(); gb = b.sample(np.empty(1024, dtype=np.float32)).copy()
            same = np.array_equal(ga.view(np.uint32), gb.view(np.uint32))
            print(voices, name, "buffer", k, "same" if same else "DIFF sum %.4f vs %.4f" % (np.abs(ga).sum(), np.abs(gb).sum()), flush=True)
        sa, sb = a.export_state(), b.export_state()
        print("   started", int(sa["started"].sum()), int(sb["started"].sum()), "offsets equal", np.array_equal(sa["current_frame_offset"], sb["current_frame_offset"]),
              "first diff voice", (np.nonzero(sa["started"] != sb["started"])[0][:8]).tolist())

print("---- detail: 512 voices, 600 untimed ons")
rng = np.random.RandomState(1)
a, b = mk(512, "1"), mk(512, "0")
ev = batch(rng, 600, False, ons_only=True)
a.note_events(ev); b.note_events(ev)
ga = a.sample(np.empty(1024, dtype=np.float32)).copy(); gb = b.sample(np.empty(1024, dtype=np.float32)).copy()
sa, sb = a.export_state(), b.export_state()
for f in ("current_frame_offset", "started", "released", "pitch_hz", "phase_accum", "lpf_last", "noise_seed", "program"):
    d = np.nonzero(sa[f] != sb[f])[0]
    print(f, "differs at", d.size, "voices; first", d[:10].tolist(), "fused", sa[f][d[:5]].tolist(), "plain", sb[f][d[:5]].tolist())
u, c = np.unique(sa["current_frame_offset"], return_counts=True)
print("fused offsets", dict(zip(u.tolist(), c.tolist())))
