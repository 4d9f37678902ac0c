"""HBM bytes per launch of conv16x3p_kernel (the fp16x3 mode's dominant kernel) from two rocprofv3 PMC passes over the mode's step
(tools/x3p_traffic.sh): FETCH_SIZE x 2 (gfx950 request-size correction, MI355X_MICROARCH.md section HBM; KiB units) + WRITE_SIZE,
summed over the kernel's dispatches and divided by their number -> profiles/r06_x3p_traffic.json (quoted by bench.py while
csrc/conv16x3p.hip is the profiled source).  python tools/x3p_traffic.py <fetch_dir> <write_dir> <batch> <out.json>"""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(d, counter):
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if 'conv16x3p_kernel' in r['Kernel_Name'] and r['Counter_Name'] == counter:
            per[r['Dispatch_Id']] = per.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
    return per


def algorithmic_bytes(B):
    """the plane region of PanopticDeepLabPR / resnet50 at 1024^2 (pdl_net.hip run32): per launch input map + packed weights +
    output (+ residual), 4 B per element (hl32 or fp32), one pass each"""
    M = B * 64 * 64
    L = []      # (M_in, Cin, K, Cout, res)

    def conv(m_in, cin, k, cout, res=False, m_out=None):
        L.append((m_in * cin * 4 + cout * k * k * cin * 4 + (m_out or M) * cout * 4 * (2 if res else 1)))
    conv(B * 128 * 128, 256, 3, 256)                       # layer3.0.conv2 (stride 2)
    for _ in range(5):
        conv(M, 1024, 1, 256); conv(M, 256, 3, 256); conv(M, 256, 1, 1024, True)
    conv(M, 1024, 1, 512); conv(M, 512, 3, 512); conv(M, 1024, 1, 2048); conv(M, 512, 1, 2048, True)      # layer4.0 (+ shortcut)
    for _ in range(2):
        conv(M, 2048, 1, 512); conv(M, 512, 3, 512); conv(M, 512, 1, 2048, True)
    conv(M, 2048, 1, 512); [conv(M, 2048, 3, 512) for _ in range(3)]                                       # merged ASPP branches
    conv(M, 1024, 1, 256); conv(M, 1024, 1, 256)                                                           # projections
    return sum(L), len(L)


def main():
    fd, wd, B, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fe, wr = per_dispatch(fd, 'FETCH_SIZE'), per_dispatch(wd, 'WRITE_SIZE')
    n = len(fe)
    assert n and n == len(wr), (n, len(wr))
    fetch = sum(fe.values()) * 1024 * 2
    write = sum(wr.values()) * 1024
    alg, per_step = algorithmic_bytes(B)
    steps = n / per_step
    src = open(os.path.join(ROOT, 'empanada-napari_amd', 'csrc', 'conv16x3p.hip'), 'rb').read()
    res = {'kernel': 'conv16x3p_kernel', 'batch': B, 'dispatches': n, 'launches_per_step': per_step, 'steps_in_trace': steps,
           'fetch_bytes_corrected': fetch, 'write_bytes': write, 'hbm_bytes_per_launch': (fetch + write) / n,
           'algorithmic_bytes_per_launch': alg / per_step, 'ratio': (fetch + write) / n / (alg / per_step),
           'source_sha16': hashlib.sha256(src).hexdigest()[:16],
           'note': 'FETCH_SIZE counts the L2 memory-side requests, Infinity-Cache hits included (re-reads of the 3x3 taps and of weights '
                   'shared by all workgroups appear here although most never reach HBM)'}
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
