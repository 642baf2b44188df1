#!/bin/bash
# GPU box: SQ counters of the k-NN kernel (scripts/knn_time.py), one counter group per run.  usage: knn_pmc.sh <tag> [knob=value ...]
set -e -o pipefail
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python3 scripts/knn_time.py "$@" > $out/run$i.log 2>&1 || { tail -5 $out/run$i.log; exit 1; }
done
python3 - "$out" <<'P'
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "knn" not in k: continue
        k = k.split("<")[0].split("(")[0]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
with open(out + "/knn_pmc.txt", "w") as fh:
    for k in acc:
        for c in sorted(acc[k]):
            line = f"{k} {c} per-launch {acc[k][c] / cnt[k][c]:.4g} (launches {cnt[k][c]})"
            print(line); fh.write(line + "\n")
P
rm -rf $out/p*
