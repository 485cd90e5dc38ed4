import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import synth2_amd as s2
from helpers import Pair, make_patch, oracle_cfg_from_patch, ulp_diff
def swap(pr,p):
    pr.gpu.set_patch(p); pr.cpu.config=oracle_cfg_from_patch(p)
A=make_patch(osc_kind=1); D=make_patch(osc_kind=4); D2=make_patch(osc_kind=5,lpf_kind=3)
for seq in ([A,D,A,D],[D,A,D],[D,D2,D],[A,D2,A,D2]):
  for frames in (256,17,1):
    pr=Pair(64,seq[0],max_frames=1024)
    for v in range(20): pr.note_on(40+v)
    for k,pt in enumerate(seq):
        if k: swap(pr,pt)
        g,o=pr.render_voices(frames)
        bad=np.nonzero(g.view(np.uint32)!=o.view(np.uint32))
        print([p.osc_kind for p in seq],frames,"step",k,"kind",pt.osc_kind,"mismatches",bad[0].size, (bad[0][0],bad[1][0],g[bad[0][0],bad[1][0]],o[bad[0][0],bad[1][0]]) if bad[0].size else "")
        if k==1: pr.note_on(70); pr.note_off(45)
