"""bench.py against another build of libs2r.so: S2R_AB_LIB=<path> python tools/ab_bench.py [bench.py flags]."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synth2_amd.build as _b
if os.environ.get("S2R_AB_LIB"):
    _b.LIB = os.environ["S2R_AB_LIB"]
    _b.needs_build = lambda: False
import bench
bench.main()
