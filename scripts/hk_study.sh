#!/bin/bash
# GPU box: timings of the H contraction (panel kernel vs tiled GEMM, several shapes) and the MFMA / wait counters of the
# panel kernel.  Usage: hk_study.sh <tag> [extra tuning key=value ...]
set -o pipefail
tag=${1:-hk}; shift
o=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $o/hk_time.txt
for args in "1000000 1000 200 10" "1000000 1000 200 10 hk_panel=0" "1000000 100 200 10" "1000000 100 200 10 hk_panel=0" "1000000 1000 96 10" "1000000 1000 96 10 hk_panel=0" "1000000 2000 200 5" "1000000 2000 200 5 hk_panel=0"; do
  timeout -k 10 120 python3 scripts/run_hk.py $args "$@" >> $o/hk_time.txt 2>&1 < /dev/null
done
grep "^hk " $o/hk_time.txt
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $o/pmc_mfma -- python3 scripts/run_hk.py 1000000 1000 200 3 "$@" > $o/pmc.log 2>&1 < /dev/null
f=$(find $o/pmc_mfma -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 scripts/pmc_mfma_summary.py $f > $o/hk_pmc_mfma.csv && cat $o/hk_pmc_mfma.csv
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $o/pmc_wait -- python3 scripts/run_hk.py 1000000 1000 200 3 "$@" > $o/pmc2.log 2>&1 < /dev/null
f=$(find $o/pmc_wait -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" > $o/hk_pmc_wait.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
for name, c in agg.items():
    if "hk_panel" in name or "gemm_f64" in name:
        print(name, {k: "%.4g" % v for k, v in c.items()})
PY
cat $o/hk_pmc_wait.txt
rm -rf $o/pmc_mfma $o/pmc_wait
