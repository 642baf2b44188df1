#!/bin/bash
# GPU box: the eigensolver at BASELINE configs[2] over several start blocks (eig_start_stream = 0..N-1) under the given knobs:
# iterations, products and time per start block, and their means.  usage: eig_seeds.sh <nseeds> [knob=value ...]
n=$1; shift
for sd in $(seq 0 $((n-1))); do
  python3 scripts/eig_trace.py 3 eig_start_stream=$sd "$@" 2>&1 | tail -1
done | python3 -c "
import sys, re
ts=[]; its=[]; pr=[]
for line in sys.stdin:
    m = re.search(r'eig ms: ([\d. ]+) \{.outer_iterations.: (\d+), .g_products.: (\d+)', line)
    if not m: print(line.strip()); continue
    t = min(float(x) for x in m.group(1).split()); ts.append(t); its.append(int(m.group(2))); pr.append(int(m.group(3)))
print('$*', 'times', ' '.join('%.2f' % t for t in ts), '| iterations', its, '| mean %.2f ms, %.1f iterations, %.1f products' % (sum(ts)/len(ts), sum(its)/len(its), sum(pr)/len(pr)))
"
