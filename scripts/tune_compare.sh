#!/bin/bash
# old vs new eigensolver defaults over several workloads (GPU box)
OLD="--tune eig_guard_sweeps=2 --tune eig_cut_pct=100 --tune eig_amp_exp=7 --tune eig_overlap_below_e6=1000"
run() { python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=j['config']['eig']
print('   %7.2f ms  eig %6.2f  it %d prod %d' % (j['ms_per_step'], j['stage_ms_per_step']['eig'], e['outer_iterations'], e['g_products']))"; }
for cfg in "--n 1000000 --d 16 --s 5000 --r 10 --K 200" "--n 100000 --d 3 --s 2000 --r 5 --K 100" "--n 300000 --d 8 --s 3000 --r 8 --K 150" "--n 500000 --d 16 --s 4000 --r 6 --K 300" "--n 200000 --d 4 --s 1000 --r 4 --K 50" "--n 400000 --d 32 --s 6000 --r 12 --K 100" "--n 1000000 --d 16 --s 5000 --r 10 --K 400"; do
  echo "$cfg"; echo -n " new"; run $cfg; echo -n " old"; run $cfg $OLD
done
