#!/bin/bash
# development aid: builds ablated variants of libs2r (results are WRONG by construction) and
# times the render kernel for each in turn.  usage (on the GPU box): tools/ablate.sh
set -u
cd /root/repo
SRC="synth2_amd/csrc/s2r_kernels.hip synth2_amd/csrc/s2r_host.cpp synth2_amd/csrc/s2r_patch.cpp"
cp synth2_amd/libs2r.so /tmp/libs2r_good.so
for V in ${ABLATE_LIST:-NONE MIX RECUR NOISE BARRIER}; do V=${V//+/ -DS2R_ABLATE_};
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -Isynth2_amd/csrc -DS2R_ABLATE_$V -o synth2_amd/libs2r.so $SRC 2>/dev/null
  touch synth2_amd/libs2r.so
  echo "ablate $V:"; python tools/frames_sweep.py 2>&1 | grep "voices  65536"
done
cp /tmp/libs2r_good.so synth2_amd/libs2r.so
