#!/bin/bash
# GPU box: time the C3 eigensolve under several tuning-knob sets.  usage: eig_tune.sh "k=v k=v" "k=v" ...
for t in "$@"; do
  printf "%-70s " "$t"
  python3 scripts/eig_trace.py 4 $t 2>&1 | tail -1 | sed -e "s/'dense.*top/top/"
done
