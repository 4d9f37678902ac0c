#!/bin/bash
# Per-layer roofline of the PanopticBiFPN forward (gpurun from the repo root) -> gpurun_out/profiles/bifpn_layer_roofline.csv
REPO=$(pwd); OUT=$REPO/gpurun_out/profiles; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp EMP_MODEL=bifpn EMP_LAYER_LOG=/tmp/bif_layers.log
rocprofv3 --kernel-trace --output-format csv -d /tmp/bifl -o bifl -- python3 $REPO/tools/layer_roofline.py run 32 1024 > /tmp/bifl.log 2>&1
python3 $REPO/tools/layer_roofline.py join /tmp/bifl /tmp/bif_layers.log $OUT/bifpn_layer_roofline.csv | tail -1
