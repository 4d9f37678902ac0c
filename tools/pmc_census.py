"""Joins the passes of tools/pmc_census.sh: one line per kernel (per launch averages over the captured launches)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hbm_traffic import short  # noqa: E402

per = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
dur = defaultdict(list)
for d in sys.argv[1:]:
    fc = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    ft = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
    if not fc or not ft:
        print('no data in', d)
        continue
    name = {}
    for r in csv.DictReader(open(ft[0])):
        name[r['Dispatch_Id']] = short(r['Kernel_Name'])
        if d == sys.argv[1]:
            dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    seen = set()
    for r in csv.DictReader(open(fc[0])):
        k = name.get(r['Dispatch_Id'])
        if k is None:
            continue
        per[k][r['Counter_Name']] += float(r['Counter_Value'])
        if (r['Dispatch_Id'], r['Counter_Name']) not in seen:
            seen.add((r['Dispatch_Id'], r['Counter_Name']))
            cnt[k][r['Counter_Name']] += 1
rows = []
for k, c in per.items():
    def avg(n):
        return c[n] / cnt[k][n] if cnt[k].get(n) else float('nan')
    us = sum(dur[k]) / max(len(dur[k]), 1)
    cyc = avg('GRBM_GUI_ACTIVE') / 8            # shader-engine cycles of the launch (sum over the 8 XCDs)
    cu = 256 * cyc
    fetch = 2 * avg('FETCH_SIZE') * 1024        # KiB units, doubled (MI355X_MICROARCH.md, HBM / rocprofv3)
    write = avg('WRITE_SIZE') * 1024
    rd, wr = avg('TCP_TCC_READ_REQ_sum'), avg('TCP_TCC_WRITE_REQ_sum')
    rows.append((us * len(dur[k]), k, len(dur[k]), us, cyc / us / 1e3 if us else 0, avg('TD_TD_BUSY_sum') / cu, avg('TA_TA_BUSY_sum') / cu,
                 avg('SQ_LDS_IDX_ACTIVE') / cu, avg('SQ_LDS_BANK_CONFLICT') / max(avg('SQ_LDS_IDX_ACTIVE'), 1),
                 avg('SQ_VALU_MFMA_BUSY_CYCLES') / (4 * cu), rd / 1e6, wr / 1e6, fetch / 1e9, write / 1e9))
rows.sort(reverse=True)
print('%-34s %4s %8s %5s %5s %5s %5s %5s %5s %8s %8s %7s %7s' % ('kernel', 'n', 'us', 'GHz', 'TD', 'TA', 'LDS', 'confl', 'MFMA', 'rdreq M', 'wrreq M', 'fetchGB', 'writeGB'))
for _, k, n, us, ghz, td, ta, lds, cf, mf, rd, wr, fe, wrb in rows[:28]:
    print('%-34s %4d %8.1f %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f %8.1f %8.1f %7.3f %7.3f' % (k[:34], n, us, ghz, td, ta, lds, cf, mf, rd, wr, fe, wrb))
