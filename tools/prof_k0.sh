set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/prof_k0
mkdir -p $O
cd $R
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq1 -- python3 tools/jump_cost.py 10000 1000 0,1 > $O/out1.txt 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2 -- python3 tools/jump_cost.py 10000 1000 0,1 > $O/out2.txt 2>&1
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r2/prof_k0'
for d in ('sq1','sq2'):
    f=glob.glob(f'{O}/{d}/**/*_counter_collection.csv', recursive=True)[0]
    rows=[r for r in csv.DictReader(open(f)) if 'logl_kernel' in r['Kernel_Name'] and 'true>' not in r['Kernel_Name']]
    # dispatches in order: first 23 launches are k=0, next 23 k=1
    by=collections.defaultdict(lambda: collections.defaultdict(list))
    ids=sorted(set(int(r['Dispatch_Id']) for r in rows))
    half=ids[len(ids)//2]
    for r in rows:
        key='k=0' if int(r['Dispatch_Id'])<half else 'k=1'
        by[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for key in by:
        print(d,key,{c:round(sum(v)/len(v)/2500,1) for c,v in by[key].items()},' (per wave; SQ cycle counters in quad-cycles)')
PY
