# PMC passes over the HBM-regime single step:  bash tools/pmc_hbm.sh   (on the GPU box)
# S=25, 139 264 games (2 GiB) at 7 / 5 / 3 workgroups per CU (unused dynamic LDS, A/B library) and S=16, 524 288 games.
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_hbm; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export TG_LIB_VARIANT=ab
P1="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
P2="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum"
P3="TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"
P4="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P5="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"
P6="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
P7="GRBM_GUI_ACTIVE TCC_EA0_WRREQ_DRAM_sum TCC_NORMAL_WRITEBACK_sum TCC_BUSY_sum"
run() {  # tag S B pad
  tag=$1; S=$2; B=$3; pad=$4
  if [ -n "$pad" ]; then export TG_S25_LDS_PAD=$pad; else unset TG_S25_LDS_PAD; fi
  i=0
  for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6" "$P7"; do
    i=$((i+1))
    rocprofv3 --pmc $P --output-format csv -d $OUT/$tag/p$i -o pmc -- python3 $R/tools/prof_one.py --op step --S $S --B $B --R 8 --iters 4 > $OUT/$tag.p$i.log 2>&1 || { echo "pass $i of $tag failed"; tail -3 $OUT/$tag.p$i.log; }
    echo "$tag pass $i done"
  done
}
run s25_wg7 25 139264 0
run s25_wg5 25 139264 20000
run s25_wg3 25 139264 36000
run s16 16 524288 ""
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_hbm'
res = collections.OrderedDict()
for tag in ('s25_wg7', 's25_wg5', 's25_wg3', 's16'):
    agg = collections.defaultdict(list)
    for f in glob.glob(out + f'/{tag}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'step_kernel' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    res[tag] = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
names = sorted({k for v in res.values() for k in v})
with open(out + '/../r04_pmc_hbm.txt', 'w') as fo:
    fo.write('%-44s' % 'counter (mean per launch)' + ''.join('%16s' % t for t in res) + '\n')
    for n in names:
        fo.write('%-44s' % n + ''.join('%16.4g' % res[t].get(n, float('nan')) for t in res) + '\n')
print(open(out + '/../r04_pmc_hbm.txt').read())
PY
rm -rf $OUT
