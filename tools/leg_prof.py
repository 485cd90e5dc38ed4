#!/usr/bin/env python3
"""One of bench.py's config legs on its own (tools/profile_leg.sh runs it under rocprofv3): the same function, the same
schedule, the same call sequence as the leg on the bench line — `c2` = config_legs[0] (65 536 voices, saw + ADSR + SVF, two
buffers in flight), `c4` = config_legs[1] (32 768 voices, DPW saw + SVF, 4x oversampled).  Prints the leg as JSON.
usage: tools/leg_prof.py c2|c4 [steps] [warmup]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

LEGS = {
    "c2": dict(name="config[2]", voices=65536, patch_text="synth c2 { lpf.kind = svf_lp; lpf.q = 1.4 }", oversampled=False, bytes_per_voice=36 + 20),
    "c4": dict(name="config[4] share", voices=32768, patch_text="synth c4 { osc.kind = dpw_saw; lpf.kind = svf_lp; lpf.q = 1.4 }", oversampled=True,
               bytes_per_voice=40 + 24),
}


def main():
    which = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    leg, _s = bench.run_config_leg(steps=steps, warmup=warmup, **LEGS[which])
    print(json.dumps(leg))


if __name__ == "__main__":
    main()
