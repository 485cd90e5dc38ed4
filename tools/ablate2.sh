#!/bin/bash
# development aid: builds variants of libs2r with one piece of the branch-free chunk removed
# (results are WRONG by construction) so that tools/ablate2_run.sh can time them on the GPU box.
# usage (in the build container): tools/ablate2.sh ; then gpurun tools/ablate2_run.sh
set -u
cd /root/repo
SRC="synth2_amd/csrc/s2r_kernels.hip synth2_amd/csrc/s2r_host.cpp synth2_amd/csrc/s2r_patch.cpp"
mkdir -p tools/ubench/_build
for V in ${ABLATE_LIST:-NONE NOISE AMP SEL LIVE TILE PHASE}; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -shared -Iinclude -Isynth2_amd/csrc -DS2R_ABL_$V -o tools/ubench/_build/libs2r_$V.so $SRC 2>/dev/null; echo built $V ) &
done
wait
