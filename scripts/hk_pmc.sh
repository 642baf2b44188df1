#!/bin/bash
# GPU box: MFMA-busy / clock counters of the H contraction for each tuning variant given.  Usage: hk_pmc.sh <tag> "<tuning a>" ...
tag=${1:-hkp}; shift
o=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for v in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $o/pmc_$i -- python3 scripts/run_hk.py 1000000 1000 200 6 $v > $o/pmc_$i.log 2>&1 < /dev/null
  f=$(find $o/pmc_$i -name "*counter_collection.csv" | head -1)
  echo "== $v" >> $o/hk_pmc.txt
  [ -n "$f" ] && python3 scripts/pmc_mfma_summary.py $f | grep -v "^kernel" >> $o/hk_pmc.txt
  rm -rf $o/pmc_$i
done
cat $o/hk_pmc.txt
