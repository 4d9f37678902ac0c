#!/bin/bash
# Round 6: the intermediate raster of the 256 x 256 tile on the write-bound 1x1 expansions -- an XCD keeps
# g cout tiles resident and walks its pixel tiles (EMP_CONV256_NGROUP=g) against "all cout tiles of a pixel tile side by side"
# (g = 0): per-shape time (HIP events, batch 32) and FETCH_SIZE / WRITE_SIZE per launch (separate PMC passes).
#   bash tools/conv_raster_ab.sh    (through gpurun, from the repo root) -> gpurun_out/conv_raster.txt
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/conv_raster.txt
for g in 0 2 4; do
  echo "==== EMP_CONV256_NGROUP=$g: ms / TFLOP/s (tools/conv_bench.py 32: the 256x256 column)" >> $OUT/conv_raster.txt
  for sh in "l3.conv3" "l4.conv3" "l4.conv1" "l3.conv1"; do
    EMP_CONV256_NGROUP=$g python3 $REPO/tools/conv_bench.py 32 "$sh" 2>/dev/null | grep -v amdgpu >> $OUT/conv_raster.txt
  done
  rm -rf /tmp/ctf /tmp/ctw
  EMP_CONV256_NGROUP=$g rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/ctf -o f -- python3 $REPO/tools/conv_traffic.py run 32 > /tmp/ctf.log 2>&1
  EMP_CONV256_NGROUP=$g rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/ctw -o w -- python3 $REPO/tools/conv_traffic.py run 32 > /tmp/ctw.log 2>&1
  echo "---- traffic per launch, EMP_CONV256_NGROUP=$g" >> $OUT/conv_raster.txt
  python3 $REPO/tools/conv_traffic.py join /tmp/ctf /tmp/ctw 32 2>&1 | grep -E "shape|l3.conv3|l4.conv3|l4.conv1|l3.conv1" >> $OUT/conv_raster.txt
done
cat $OUT/conv_raster.txt
