#!/bin/bash
# GPU box: clock and wait counters of the solver-shaped products (scripts/gemm_time.py).  usage: gemm_pmc.sh <tag>
set -e -o pipefail
tag=$1; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python3 scripts/gemm_time.py > $out/run$i.log 2>&1 || { tail -5 $out/run$i.log; exit 1; }
done
python3 - "$out" <<'P'
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int)); dur = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gemm_f64_kernel" not in k: continue
        key = (k.split("(")[0], row["Grid_Size"])
        acc[key][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[key][row["Counter_Name"]] += 1
        if "Start_Timestamp" in row: dur[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
for key in sorted(acc):
    print(key, "mean duration us %.1f" % (sum(dur[key]) / max(len(dur[key]), 1) / 1e3))
    for c in sorted(acc[key]): print("   %-28s %.4g" % (c, acc[key][c] / cnt[key][c]))
P
rm -rf $out/p*
