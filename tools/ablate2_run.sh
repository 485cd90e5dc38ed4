#!/bin/bash
# on the GPU box: time the settled one-pole kernel for every variant built by tools/ablate2.sh
cd /root/repo
cp synth2_amd/libs2r.so /tmp/libs2r_good.so
for f in tools/ubench/_build/libs2r_*.so; do
  cp $f synth2_amd/libs2r.so
  echo "$(basename $f): $(V=65536 KINDS=0 python tools/dspf_time.py 2>&1 | grep 'kind 0')"
done
cp /tmp/libs2r_good.so synth2_amd/libs2r.so
