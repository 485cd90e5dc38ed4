#!/bin/bash
# development aid: bench.py with the fills of s2r_fill_begin on one stream (S2R_OVERLAP=0) and on two (default), alternating
N=${1:-3}; shift
for i in $(seq 1 $N); do
  for v in 0 1; do
    S2R_OVERLAP=$v python bench.py --no-config-legs --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S2R_OVERLAP=$v', 'ms_per_step %.5f' % d['ms_per_step'], 'kernel_ms %.5f' % d['roofline']['kernel_ms'], 'value %.4g' % d['value'], 'host', d['host_time_per_step'])" || exit 1
  done
done
