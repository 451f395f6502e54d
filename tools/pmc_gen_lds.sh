# Which phase of the fused generator owns its LDS bank conflicts?  PMC pass per ablation mask (A/B library).  On the GPU box.
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_gen; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export TG_LIB_VARIANT=ab
for m in 0 2 4 8 24 28; do
  export TG_GF_ABLATE=$m
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/m$m -o pmc -- python3 $R/tools/prof_one.py --op gen --S 25 --B 4096 --R 64 --iters 4 > $OUT/m$m.log 2>&1 || { echo "mask $m failed"; tail -3 $OUT/m$m.log; }
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_gen'
print("mask (1 no Philox, 2 no draw LDS writes, 4 no tiles, 8 no target store, 16 no token store)")
for m in (0, 2, 4, 8, 24, 28):
    agg = collections.defaultdict(list)
    for f in glob.glob(out + f'/m{m}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'gen_fused' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(m, {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())})
PY
rm -rf $OUT
