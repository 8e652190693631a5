# A/B of the half-buffered halo kernel (HCTR_HALFHALO=1) against halo4, per layer:
#   bash tools/profile_halfhalo_ab.sh [trace|sq]     (on the GPU box; writes gpurun_out/hh_<mode>_h{0,1}/)
set -e
MODE=${1:-trace}
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for V in 0 1; do
  OUT=$REPO/gpurun_out/hh_${MODE}_h$V
  rm -rf $OUT; mkdir -p $OUT
  if [ $MODE = sq ]; then
    HCTR_HALFHALO=$V rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-second-mode --no-extra-configs --no-oracle-check --layer-file $OUT/layers.txt > $OUT/a.log 2>&1
  else
    HCTR_HALFHALO=$V rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-second-mode --no-extra-configs --no-oracle-check --layer-file $OUT/layers.txt > $OUT/bench_stats.log 2>&1
  fi
done
