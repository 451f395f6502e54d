set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_genfused; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -o pmc -- python3 $R/tools/prof_one.py --op gen --S 25 --B 4096 --R 64 --iters 5 > $OUT/p1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p2 -o pmc -- python3 $R/tools/prof_one.py --op gen --S 25 --B 4096 --R 64 --iters 5 > $OUT/p2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/p3 -o pmc -- python3 $R/tools/prof_one.py --op gen --S 25 --B 4096 --R 64 --iters 5 > $OUT/p3.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $R/tools/prof_one.py --op gen --S 25 --B 4096 --R 64 --iters 20 > $OUT/kt.log 2>&1 || exit 1
python3 - <<'PY'
import csv,glob,collections,os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_genfused'
for p in ('p1','p2','p3'):
    f=glob.glob(out+f'/{p}/**/*counter_collection.csv',recursive=True)[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'gen_fused' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(p,k,sum(v)/len(v))
f=glob.glob(out+'/kt/**/*kernel_stats.csv',recursive=True)[0]
print(open(f).read()[:1500])
PY
