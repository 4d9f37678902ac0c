#!/bin/bash
# bash tools/x3p_traffic.sh [batch]   (through gpurun, from the repo root) -> gpurun_out/x3p_traffic.json, x3_kernel_stats.csv
B=${1:-16}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/xf /tmp/xw /tmp/xs
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/xf -o f -- python3 $REPO/tools/precision_profile.py fp16x3 $B 2 > /tmp/xf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/xw -o w -- python3 $REPO/tools/precision_profile.py fp16x3 $B 2 > /tmp/xw.log 2>&1
python3 $REPO/tools/x3p_traffic.py /tmp/xf /tmp/xw $B $OUT/x3p_traffic.json
# the same mode through bench.py under --kernel-trace --stats (a record a reader can recompute the fraction from)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xs -o xs -- python3 $REPO/bench.py --precision fp16x3 --batch $B --steps 5 --warmup 2 --no-cpu-baseline --stack3d 0 --engine2d 0 --latency 0 --fine-boundaries 0 --fp32-mode 0 > $OUT/x3_bench_profiled.json 2> /tmp/xs.log
cp $(find /tmp/xs -name "*kernel_stats.csv" | head -1) $OUT/x3_kernel_stats.csv
tail -1 $OUT/x3_bench_profiled.json | cut -c1-1500
head -12 $OUT/x3_kernel_stats.csv | cut -c1-220
