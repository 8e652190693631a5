#!/bin/bash
# Build an alternative libhctr_hip.so with extra compile flags for a same-box A/B (run here, no GPU needed):
#   bash tools/ab_build.sh <name> -DNOPRIO=1     ->  gpurun_out/ab/<name>.so     (travels with the snapshot? no:
# gpurun_out/ is not sent - the library is written to ab_libs/<name>.so, which is git-ignored via *.so)
# Use on the GPU box:  HCTR_LIB_PATH=ab_libs/<name>.so python bench.py ...
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/handwritten-chinese-ocr-samples_amd/csrc
OUT=$ROOT/ab_libs; mkdir -p $OUT/obj_$NAME
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
$HIPCC $COMMON "$@" -c $SRC/kernels.hip -o $OUT/obj_$NAME/kernels.o
$HIPCC $COMMON -ffp-contract=off -c $SRC/preprocess.hip -o $OUT/obj_$NAME/preprocess.o
$HIPCC $COMMON "$@" -x hip -c $SRC/engine.cpp -o $OUT/obj_$NAME/engine.o
$HIPCC $COMMON -ffp-contract=off -c $SRC/beam_search.cpp -o $OUT/obj_$NAME/beam_search.o
$HIPCC $COMMON -ffp-contract=off -c $SRC/ngram_lm.cpp -o $OUT/obj_$NAME/ngram_lm.o
$HIPCC $COMMON -x hip -c $SRC/gather.cpp -o $OUT/obj_$NAME/gather.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so $OUT/obj_$NAME/*.o -lpthread -ldl
echo $OUT/$NAME.so
