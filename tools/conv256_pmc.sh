#!/bin/bash
# PMC passes on the ASPP 3x3 launch of conv_igemm256_kernel (profiles/r02_conv256_pmc.txt): bash tools/conv256_pmc.sh  (gpurun)
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf /tmp/pmc_$name; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- python3 $REPO/tools/one_conv.py 64 64 2048 256 3 4 67 32 > /tmp/pmc_$name.log 2>&1; }
run clk GRBM_GUI_ACTIVE
run mfma SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16
python3 - <<PY
import csv, glob, collections
for name in ('clk','mfma'):
    f=glob.glob('/tmp/pmc_%s/**/*counter_collection.csv'%name, recursive=True)
    if not f: print(name,'no counter file'); continue
    rows=[r for r in csv.DictReader(open(f[0])) if 'conv_igemm256' in r['Kernel_Name']]
    by=collections.OrderedDict()
    for r in rows:
        k=(r['Dispatch_Id'], r['Counter_Name'])
        by[k]=by.get(k,0)+float(r['Counter_Value'])
    t=glob.glob('/tmp/pmc_%s/**/*kernel_trace.csv'%name, recursive=True)[0]
    dur={r['Dispatch_Id']:(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(t)) if 'conv_igemm256' in r['Kernel_Name']}
    for (d,c),v in by.items():
        extra=''
        if c=='GRBM_GUI_ACTIVE': extra=' -> effective clock %.2f GHz'%(v/8/dur[d]/1e3)
        if c=='SQ_VALU_MFMA_BUSY_CYCLES': extra=' -> %.3f M busy cycles per SIMD'%(v/1024/1e6)
        print('%-5s dispatch %-4s %-32s %16.0f  duration %8.1f us%s'%(name,d,c,v,dur[d],extra))
PY
