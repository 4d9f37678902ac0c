REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf /tmp/pmc_$name; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- python3 $REPO/tools/one_conv16x3.py 64 64 2048 256 3 4 8 > /tmp/pmc_$name.log 2>&1; }
run clk GRBM_GUI_ACTIVE
run mfma SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16
run inst SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
python3 - <<PY
import csv, glob, collections
for name in ('clk','mfma','inst','wait','lds'):
    f=glob.glob('/tmp/pmc_%s/**/*counter_collection.csv'%name, recursive=True)
    if not f: print(name,'no counter file'); print(open('/tmp/pmc_%s.log'%name).read()[-500:]); continue
    rows=[r for r in csv.DictReader(open(f[0])) if 'conv16x3' in r['Kernel_Name']]
    by=collections.OrderedDict()
    for r in rows:
        k=(r['Dispatch_Id'], r['Counter_Name'])
        by[k]=by.get(k,0)+float(r['Counter_Value'])
    t=glob.glob('/tmp/pmc_%s/**/*kernel_trace.csv'%name, recursive=True)[0]
    dur={r['Dispatch_Id']:(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(t)) if 'conv16x3' in r['Kernel_Name']}
    for (d,c),v in by.items():
        print('%-5s dispatch %-4s %-32s %16.0f  duration %8.1f us'%(name,d,c,v,dur[d]))
PY
