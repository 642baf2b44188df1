#!/bin/bash
# Collects the evidence set of one round on the GPU box: bench line, rocprofv3 kernel stats of the same command, and the
# PMC passes (separate runs, kernel-trace only, as MI355X_MICROARCH.md prescribes), then the per-kernel summaries.
# Usage: collect_profiles.sh <tag>      (results under gpurun_out/<tag>/; copy what is to be judged into profiles/)
set -e -o pipefail
tag=${1:-r02x}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench.json 2> $out/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-e2e > $out/bench_under_rocprof.json 2> $out/stats.err
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > $out/pmc_fetch.json 2> $out/pmc_fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > $out/pmc_write.json 2> $out/pmc_write.err
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > $out/pmc_mfma.json 2> $out/pmc_mfma.err
echo "mfma done"
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
f=$(find $out/pmc_fetch -name "*counter_collection.csv" | head -1); w=$(find $out/pmc_write -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_summary.py $f $w > $out/pmc_hbm_traffic.csv
python3 scripts/pmc_mfma_summary.py $(find $out/pmc_mfma -name "*counter_collection.csv" | head -1) > $out/pmc_mfma_util.csv
rm -rf $out/stats $out/pmc_fetch $out/pmc_write $out/pmc_mfma
head -12 $out/kernel_stats.csv | cut -c1-150
head -12 $out/pmc_hbm_traffic.csv
head -8 $out/pmc_mfma_util.csv
