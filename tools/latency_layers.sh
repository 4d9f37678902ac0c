#!/bin/bash
# Per-layer timings of ONE batch-1 forward (1024^2): bash tools/latency_layers.sh   (through gpurun)
REPO=$(pwd); OUT=$REPO/gpurun_out/latency; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ll
EMP_LAYER_LOG=/tmp/ll_layers.log rocprofv3 --kernel-trace --output-format csv -d /tmp/ll -o ll -- python3 $REPO/tools/layer_roofline.py run 1 1024 > /tmp/ll.log 2>&1
python3 $REPO/tools/layer_roofline.py join /tmp/ll /tmp/ll_layers.log $OUT/layers_b1.csv > /dev/null || tail -5 /tmp/ll.log
python3 - <<PY
import csv
f=open('$OUT/layers_b1.csv'); f.readline()
rows=list(csv.DictReader(f))
tot=0
for r in rows:
    if r['kind']=='total': continue
    tot+=float(r['us'])
    print('%-40s M=%7s %5s->%5s k%s  %7.1f us  %6s TF'%(r['layer'][:40], r['M'], r['Cin'], r['Cout'], r['k'], float(r['us']), r['TFLOPs']))
print('sum of MFMA launches', tot)
PY
