#!/bin/bash
# GPU box: time the panel kernel's variants.  Usage: hk_var.sh <tag> "<tuning a>" "<tuning b>" ...
tag=${1:-hkv}; shift
o=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  timeout -k 10 120 python3 scripts/run_hk.py 1000000 1000 200 12 $v >> $o/hk_var.txt 2>&1 < /dev/null
  timeout -k 10 120 python3 scripts/run_hk.py 1000000 100 200 12 $v >> $o/hk_var.txt 2>&1 < /dev/null
done
grep "^hk " $o/hk_var.txt
