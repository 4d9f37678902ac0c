#!/bin/bash
# Per-layer timings of one forward under different conv tile dispatch rules (env knobs of conv_igemm.hip), merged into
# one table: bash tools/rule_sweep.sh   (through gpurun, from the repo root)
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/rule_sweep
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1
  rm -rf /tmp/lr_$name
  EMP_LAYER_LOG=/tmp/layers_$name.log rocprofv3 --kernel-trace --output-format csv -d /tmp/lr_$name -o lr -- python3 $REPO/tools/layer_roofline.py run 32 1024 > /tmp/lr_$name.log 2>&1
  python3 $REPO/tools/layer_roofline.py join /tmp/lr_$name /tmp/layers_$name.log $OUT/$name.csv > /dev/null || tail -5 /tmp/lr_$name.log
}
run default
EMP_CONV_256_MINK=128 EMP_CONV_256_RESMUL=1 run all256
EMP_CONV_H256=1 run h256
EMP_CONV_NO256=1 EMP_CONV_H256=1 run h256only
python3 - <<PY
import csv
names=['default','all256','h256','h256only']
tabs={}
for n in names:
    f=open('$OUT/%s.csv'%n); f.readline()
    tabs[n]={r['layer']:r for r in csv.DictReader(f)}
print('%-40s %8s %5s %5s | '%('layer','M','Cin','Cout')+' '.join('%9s'%n for n in names))
tot={n:0.0 for n in names}; best=0.0
for lay,r in tabs['default'].items():
    us=[float(tabs[n][lay]['us']) for n in names]
    if lay: best+=min(us)
    for n,u in zip(names,us):
        if lay: tot[n]+=u
    print('%-40s %8s %5s %5s | '%(lay[:40],r['M'],r['Cin'],r['Cout'])+' '.join('%9.1f'%u for u in us))
print('sum', tot, 'best-per-layer', best)
PY
