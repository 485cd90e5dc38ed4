#!/bin/bash
# development aid: A/B timing of two builds of libs2r on ONE GPU box, alternating so that clock drift hits both alike.
#   tools/ab_bench.sh synth2_amd/libs2r_A.so [rounds] [bench args...]      (B = the product library)
# prints ms_per_step and the render kernel's ms (HIP events) per run
A=$1; N=${2:-3}; shift; shift
for i in $(seq 1 $N); do
  for which in A B; do
    if [ $which = A ]; then export S2R_AB_LIB=$A; else unset S2R_AB_LIB; fi
    python bench.py --no-config-legs --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$which', 'ms_per_step %.5f' % d['ms_per_step'], 'kernel_ms %.5f' % d['roofline']['kernel_ms'], 'value %.4g' % d['value'])" || exit 1
  done
done
