#!/bin/bash
# SQ / GRBM counters for one bench step (separate PMC pass, kernel-trace-free):
#   bash tools/profile_sq.sh <tag>
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/sq_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-second-mode --layer-file $OUT/layers.txt > $OUT/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM --output-format csv -d $OUT/b -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-second-mode > $OUT/b.log 2>&1 || true
tail -n 2 $OUT/a.log; tail -n 2 $OUT/b.log
