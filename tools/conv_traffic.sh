#!/bin/bash
# Per-shape HBM traffic of the conv kernels (profiles/<round>_conv_traffic.txt): two separate PMC passes, joined with the
# algorithmic bytes.   bash tools/conv_traffic.sh   (through gpurun, from the repo root)
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ctf /tmp/ctw
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/ctf -o f -- python3 $REPO/tools/conv_traffic.py run 32 > /tmp/ctf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/ctw -o w -- python3 $REPO/tools/conv_traffic.py run 32 > /tmp/ctw.log 2>&1
python3 $REPO/tools/conv_traffic.py join /tmp/ctf /tmp/ctw 32 > $OUT/conv_traffic.txt 2>&1
cat $OUT/conv_traffic.txt
