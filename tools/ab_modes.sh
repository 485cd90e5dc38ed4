#!/bin/bash
# development aid: the headline step in its three forms on ONE GPU box, alternating so that clock drift hits all alike:
#   R  pool-resident kernel (s2r_set_resident: posted commands)        S2R_BENCH_RESIDENT=1
#   F  one launch per fill (in-kernel heads + ticket mix)              S2R_BENCH_RESIDENT=0 S2R_FUSED=2
#   T  two streams (render | mix + heads), two launches per fill       S2R_BENCH_RESIDENT=0 S2R_FUSED=1
# usage: tools/ab_modes.sh [rounds] [bench args...]
N=${1:-3}; shift
for i in $(seq 1 $N); do
  for which in R F T; do
    case $which in
      R) export S2R_BENCH_RESIDENT=1 S2R_FUSED=1;;
      F) export S2R_BENCH_RESIDENT=0 S2R_FUSED=2;;
      T) export S2R_BENCH_RESIDENT=0 S2R_FUSED=1;;
    esac
    python bench.py --no-config-legs --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['host_time_per_step']
print('$which', 'ms_per_step %.5f' % d['ms_per_step'], 'kernel_ms %.5f' % d['roofline']['kernel_ms'], 'value %.4g' % d['value'],
      'host note_events %.1f fill_begin %.1f fill_end %.1f fence %.1f us' % (h['note_events_us'], h['fill_begin_us'], h['fill_end_us'], h['final_fence_us']), 'sync %.3g' % d['value_host_api_sync'])" || exit 1
  done
done
