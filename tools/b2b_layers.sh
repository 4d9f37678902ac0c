#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/b2b; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/lr_$v
  EMP_FUSE_B2B=$v EMP_LAYER_LOG=/tmp/layers_$v.log rocprofv3 --kernel-trace --output-format csv -d /tmp/lr_$v -o lr -- python3 $REPO/tools/layer_roofline.py run 32 1024 > /tmp/lr_$v.log 2>&1
  python3 $REPO/tools/layer_roofline.py join /tmp/lr_$v /tmp/layers_$v.log $OUT/$v.csv > /dev/null || tail -5 /tmp/lr_$v.log
done
python3 - <<PY
import csv
t={}
for n in '01':
    f=open('$OUT/%s.csv'%n); f.readline()
    t[n]={r['layer']:float(r['us']) for r in csv.DictReader(f)}
for k in t['0']:
    if 'layer1' in k or 'layer2.0' in k or k=='': print('%-36s %8.1f %8s'%(k,t['0'][k], ('%.1f'%t['1'][k]) if k in t['1'] else '-'))
PY
