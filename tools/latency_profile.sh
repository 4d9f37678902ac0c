#!/bin/bash
# Per-kernel breakdown of the batch-1 call (one 1024^2 tile): bash tools/latency_profile.sh   (through gpurun)
REPO=$(pwd); OUT=$REPO/gpurun_out/latency; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lat
EMP_LAYER_LOG=/tmp/lat_layers.log rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lat -o lat -- python3 $REPO/tools/latency.py > $OUT/latency.txt 2>&1
python3 - <<PY
import csv, glob
f=glob.glob('/tmp/lat/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:16]: print('%-90s calls %5s avg %8.1f us share %5.1f%%'%(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
tail -3 $OUT/latency.txt
