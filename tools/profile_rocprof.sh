#!/bin/bash
# Collect the rocprofv3 evidence for one bench run (run on the GPU box via gpurun from the repo root):
#   bash tools/profile_rocprof.sh <tag> [extra bench.py arguments, e.g. --config c3]
# Writes gpurun_out/prof_<tag>/: kernel-trace stats, and two separate PMC passes (FETCH_SIZE, WRITE_SIZE)
# as MI355X_MICROARCH.md prescribes. Copy the summaries you want judged into profiles/.
set -e
TAG=${1:-r01}
shift || true
EXTRA="$@"
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-second-mode --layer-file $OUT/layers.txt $EXTRA > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-second-mode $EXTRA > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-second-mode $EXTRA > $OUT/bench_pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
# MFMA utilisation of the same command (own pass)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $OUT/pmc_mfma -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-second-mode $EXTRA > $OUT/bench_pmc_mfma.log 2>&1 || true
