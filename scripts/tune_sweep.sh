#!/bin/bash
# One-knob-at-a-time sweep of eigensolver tunings on the bench (GPU box): scripts/tune_sweep.sh "k=v" "k=v k2=v2" ...
for t in "$@"; do
  args=""; for kv in $t; do args="$args --tune $kv"; done
  python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline $args 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=j['config']['eig']
print('%-40s %7.2f ms  eig %6.2f  it %d prod %d' % ('$t', j['ms_per_step'], j['stage_ms_per_step']['eig'], e['outer_iterations'], e['g_products']))"
done
