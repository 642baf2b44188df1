#!/bin/bash
# scripts/tune_one.sh "<bench workload args>" "k=v ..." ...   (GPU box)
cfg="$1"; shift
for t in "$@"; do
  args=""; for kv in $t; do args="$args --tune $kv"; done
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline $cfg $args 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=j['config']['eig']
print('%-60s %7.2f ms  eig %6.2f  it %d prod %d' % ('$t', j['ms_per_step'], j['stage_ms_per_step']['eig'], e['outer_iterations'], e['g_products']))"
done
