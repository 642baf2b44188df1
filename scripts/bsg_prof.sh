#!/bin/bash
# GPU box: kernel statistics of the eigensolver at BASELINE configs[2] for a list of knob settings.
# usage: bsg_prof.sh <outdir> "<knobs of run 1>" "<knobs of run 2>" ...
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for knobs in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/p$i -- python3 $GRAFT_REPO_ROOT/scripts/eig_trace.py 4 $knobs > $GRAFT_REPO_ROOT/$out/p$i.log 2>&1 < /dev/null
  f=$(find $GRAFT_REPO_ROOT/$out/p$i -name '*kernel_stats.csv' | head -1)
  echo "== $knobs"; grep "eig ms" $GRAFT_REPO_ROOT/$out/p$i.log
  [ -n "$f" ] && grep -E "bsg_gemm|bsg_pre|jac_block|small_gemm|gemm_f64" "$f" | cut -c1-40,60-200 | awk -F, '{print $1, $2, $3, $4, $5}'
  find $GRAFT_REPO_ROOT/$out/p$i -name '*.csv' ! -name '*kernel_stats.csv' -delete
done
