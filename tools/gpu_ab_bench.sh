#!/bin/bash
# Same-box interleaved A/B of library builds (run on the GPU box from the repo root):
#   bash tools/gpu_ab_bench.sh <outdir> <rounds> <name>[=ab_libs/<file>.so] ...      ("default" = the in-tree library)
# First checks every alternative build bit for bit against the default over tools/gpu_shape_sweep.py's 30 cases.
set -e
OUT=$1; R=$2; shift 2
mkdir -p $OUT
timeout -k 10 300 python tools/gpu_shape_sweep.py dump $OUT/default.npz > $OUT/dump_default.log 2>&1
for V in "$@"; do
  [ $V = default ] && continue
  HCTR_LIB_PATH=ab_libs/$V.so timeout -k 10 300 python tools/gpu_shape_sweep.py dump $OUT/$V.npz > $OUT/dump_$V.log 2>&1
  python tools/gpu_shape_sweep.py identical $OUT/default.npz $OUT/$V.npz
done
rm -f $OUT/*.npz
for i in $(seq 1 $R); do
  for V in "$@"; do
    if [ $V = default ]; then unset HCTR_LIB_PATH; else export HCTR_LIB_PATH=ab_libs/$V.so; fi
    timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-mode --no-extra-configs --no-oracle-check > $OUT/b_${V}_$i.json 2> $OUT/b_${V}_$i.err
    python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['frac'])" $OUT/b_${V}_$i.json
  done
done
