#!/bin/bash
# development aid: bench.py's busy wait between the fence and the timed region (S2R_BENCH_PREWAIT_MS), alternating values on one box
for i in 1 2 3 4 5; do for w in 5 2 1; do S2R_BENCH_PREWAIT_MS=$w python bench.py --no-config-legs --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('prewait $w ms:', 'ms_per_step %.5f' % d['ms_per_step'], 'host', ' '.join('%.1f'%d['host_time_per_step'][k] for k in ('note_events_us','fill_begin_us','fill_end_us','final_fence_us')))"; done; done
