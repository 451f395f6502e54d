# SQ counters of tg_step_i8 (one pass per counter group):  bash tools/pmc_step.sh S B   (run on the GPU box)
set -o pipefail
S=${1:-25}; B=${2:-4096}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_step; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -o pmc -- python3 $R/tools/prof_one.py --op step --S $S --B $B --iters 5 > $OUT/p1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p2 -o pmc -- python3 $R/tools/prof_one.py --op step --S $S --B $B --iters 5 > $OUT/p2.log 2>&1 || exit 1
python3 - <<'PY'
import csv,glob,collections,os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_step'
for p in ('p1','p2'):
    f=glob.glob(out+f'/{p}/**/*counter_collection.csv',recursive=True)[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'step_kernel' in r['Kernel_Name'] or 'packed_kernel' in r['Kernel_Name']: agg[(r['Kernel_Name'].split('(')[0][-40:],r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print(p,k[0],k[1],sum(v)/len(v))
PY
rm -rf $OUT
