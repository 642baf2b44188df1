#!/bin/bash
# GPU box: L2 / HBM counters of the solver's rotation (scripts/rot_once.py).  usage: rot_pmc.sh <tag> [rot_once args]
set -e -o pipefail
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python3 scripts/rot_once.py "$@" > $out/run$i.log 2>&1 || { tail -5 $out/run$i.log; echo "(group $grp failed)"; continue; }
done
python3 - "$out" <<'P'
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gemm_f64_kernel" not in k and "rot_panel" not in k: continue
        key = k.split("(")[0]
        acc[key][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[key][row["Counter_Name"]] += 1
for key in sorted(acc):
    print(key)
    for c in sorted(acc[key]): print("   %-32s %.5g per launch" % (c, acc[key][c] / cnt[key][c]))
P
rm -rf $out/p*
