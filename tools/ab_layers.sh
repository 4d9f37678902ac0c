#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/b2b; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/lr_$v
  env ${AB_VAR:-EMP_FUSE_B2B}=$v EMP_LAYER_LOG=/tmp/layers_$v.log rocprofv3 --kernel-trace --output-format csv -d /tmp/lr_$v -o lr -- python3 $REPO/tools/layer_roofline.py run 32 1024 > /tmp/lr_$v.log 2>&1
  python3 $REPO/tools/layer_roofline.py join /tmp/lr_$v /tmp/layers_$v.log $OUT/$v.csv > /dev/null || tail -5 /tmp/lr_$v.log
done
python3 - <<PY
import csv
t={}
for n in '01':
    f=open('$OUT/%s.csv'%n); f.readline()
    t[n]={r['layer']:float(r['us']) for r in csv.DictReader(f)}
keys=list(t['0'])
for k in t['1']:
    if k not in t['0']: keys.append(k)
for k in keys:
    a=t['0'].get(k); b=t['1'].get(k)
    if a is None or b is None or abs(a-b) > 0.04*max(a,b) or k=='':
        print('%-44s %8s %8s'%(k, '-' if a is None else '%.1f'%a, '-' if b is None else '%.1f'%b))
PY
