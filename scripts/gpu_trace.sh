#!/bin/bash
# GPU box: kernel-trace one run of scripts/eig_trace.py and summarise the last solve.  usage: gpu_trace.sh <tag> [eig_trace args]
set -e -o pipefail
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scripts/eig_trace.py 5 "${@:2}" > $out/eig_plain.log 2>&1 || { tail -5 $out/eig_plain.log; exit 1; }
tail -1 $out/eig_plain.log
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 scripts/eig_trace.py "$@" > $out/eig_traced.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_timeline.py $f $out/timeline.txt
rm -rf $out/trace
head -${HEADN:-30} $out/timeline.txt
