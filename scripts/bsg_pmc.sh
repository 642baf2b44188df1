#!/bin/bash
# GPU box: wait / LDS / MFMA / cache counters of the block-sparse product kernels inside the eigensolver at BASELINE configs[2].
# usage: bsg_pmc.sh <tag> [knob=value ...]
set -o pipefail
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python3 scripts/eig_trace.py 1 "$@" > $out/run$i.log 2>&1 < /dev/null || { echo "group $i failed"; tail -3 $out/run$i.log; }
done
python3 - "$out" <<'P'
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int)); dur = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "bsg_gemm_kernel" not in k and "bsg_pre_kernel" not in k: continue
        key = k.split("(")[0]
        acc[key][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[key][row["Counter_Name"]] += 1
        if "Start_Timestamp" in row: dur[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
for key in sorted(acc):
    print(key, "mean duration us %.1f" % (sum(dur[key]) / max(len(dur[key]), 1) / 1e3))
    for c in sorted(acc[key]): print("   %-32s %.5g" % (c, acc[key][c] / cnt[key][c]))
P
rm -rf $out/p*
