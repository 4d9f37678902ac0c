#!/bin/bash
# Regenerates the committed round profiles on the GPU box (run through gpurun from the repo root):
#   bench line, rocprofv3 kernel stats of the same command, per-step breakdown, per-layer roofline, HBM traffic (PMC).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/profiles
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --stack3d 0 --engine2d 0 --latency 0 --fine-boundaries 0 --fp32-mode 0 > /tmp/ks.log 2>&1
cp /tmp/ks/ks_kernel_stats.csv $OUT/bench_kernel_stats.csv
python3 $REPO/tools/step_breakdown.py /tmp/ks $OUT/step_breakdown.csv > /dev/null
EMP_LAYER_LOG=/tmp/layers.log rocprofv3 --kernel-trace --output-format csv -d /tmp/lr -o lr -- python3 $REPO/tools/layer_roofline.py run 32 1024 > /tmp/lr.log 2>&1
python3 $REPO/tools/layer_roofline.py join /tmp/lr /tmp/layers.log $OUT/layer_roofline.csv > /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o pf -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --stack3d 0 --engine2d 0 --latency 0 --fine-boundaries 0 --fp32-mode 0 > /tmp/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o pw -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --stack3d 0 --engine2d 0 --latency 0 --fine-boundaries 0 --fp32-mode 0 > /tmp/pw.log 2>&1
python3 $REPO/tools/hbm_traffic.py /tmp/pf /tmp/pw $OUT/hbm_traffic.json > $OUT/hbm_traffic.txt 2>&1
# the bench line joins the per-launch HBM bytes from profiles/<round>_hbm_traffic.json: refresh it first (box-local copy)
python3 - <<PY
import json
import sys; sys.path.insert(0, '$REPO')
import bench
d=json.load(open('$OUT/hbm_traffic.json')); d['commit']='${EMP_COMMIT:-unknown}'; d['source_hash']=bench.kernel_source_hash()
json.dump(d, open('$OUT/hbm_traffic.json','w'), indent=1)
PY
cp $OUT/hbm_traffic.json $REPO/profiles/${EMP_ROUND:-r05}_hbm_traffic.json
python3 $REPO/bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench.json
ls -la $OUT
tail -3 $OUT/hbm_traffic.txt
python3 -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['ms_per_step_uninstrumented'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
# BiFPN network (configs[0] / [4]): forward rate and per-layer table
python3 $REPO/tools/bench_bifpn.py > $OUT/bifpn.txt 2>&1
(cd $REPO && bash tools/refresh_bifpn_profile.sh) >> $OUT/bifpn.txt 2>&1
tail -3 $OUT/bifpn.txt
