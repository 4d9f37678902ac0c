cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/xt
rocprofv3 --kernel-trace --output-format csv -d /tmp/xt -o xt -- python3 $GRAFT_REPO_ROOT/tools/precision_profile.py ${PREC:-fp16x3} ${1:-16} 2 ${2:-pdl} > /tmp/xt.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/xt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step: find the last stem kernel
idx = [i for i, r in enumerate(rows) if 'stem' in r['Kernel_Name']]
start = idx[-1]
tot = 0
for r in rows[start:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    name = r['Kernel_Name'].replace('emp::(anonymous namespace)::', '').replace('void ', '')[:60]
    print(f"{d:9.1f} us  grid {r.get('Grid_Size','?'):>9s} {name}")
print('sum us', tot)
PY
