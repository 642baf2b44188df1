#!/bin/bash
# GPU box: the eigensolver's defaults against a variant (knobs given as arguments) over seven workloads.
# usage: tune_compare.sh eig_landing=0 gemm_tile64_below=0 ...
OLD=""
for kv in "$@"; do OLD="$OLD --tune $kv"; done
run() { python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=j['config']['eig']
print('   %7.2f ms  eig %6.2f  it %d prod %d' % (j['ms_per_step'], j['stage_ms_per_step']['eig'], e['outer_iterations'], e['g_products']))"; }
for cfg in "--n 1000000 --d 16 --s 5000 --r 10 --K 200" "--n 100000 --d 3 --s 2000 --r 5 --K 100" "--n 300000 --d 8 --s 3000 --r 8 --K 150" "--n 500000 --d 16 --s 4000 --r 6 --K 300" "--n 200000 --d 4 --s 1000 --r 4 --K 50" "--n 400000 --d 32 --s 6000 --r 12 --K 100" "--n 1000000 --d 16 --s 5000 --r 10 --K 400"; do
  echo "$cfg"; echo -n " defaults"; run $cfg; echo -n " variant "; run $cfg $OLD
done
