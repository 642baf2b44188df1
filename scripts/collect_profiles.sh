#!/bin/bash
# Collects the evidence set of one round on the GPU box: bench line, rocprofv3 kernel stats of the same command, and the
# PMC passes (separate runs, kernel-trace only, as MI355X_MICROARCH.md prescribes).  Usage: collect_profiles.sh <tag>
set -e -o pipefail
tag=${1:-r01x}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench.json 2> $out/bench.err
echo "bench done" 
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.err
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_fetch.json 2> $out/pmc_fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_write.json 2> $out/pmc_write.err
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_mfma.json 2> $out/pmc_mfma.err
echo "mfma done"
find $out -name "*.csv" | head -20
