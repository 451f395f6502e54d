#!/bin/bash
# Profiling recipe for one round (run on the GPU box via gpurun from the repo root):
#   bash profiles/run_profiles.sh r04 && python3 profiles/summarize.py r04 && mkdir -p gpurun_out/profiles_r04 && \
#     cp profiles/r04_* profiles/traffic_r04.json gpurun_out/profiles_r04/ && rm -rf gpurun_out/prof_r04
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/ (tens of MB of traces: more than gpurun copies back);
# `python profiles/summarize.py <tag>` condenses it ON THE BOX into the small files committed under profiles/.
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-also"
PART=${2:-all}   # "a": steps 0-3 (bench runs, kernel traces, HBM traffic counters); "b": the rest; "c": step 10 only; default all
if [ "$PART" = "all" ] || [ "$PART" = "a" ]; then
# 0. unprofiled: the driver's own command (what BENCH_rNN.json records) and the default command, full JSON lines
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err || exit 1
cp $R/bench_also.json $OUT/bench_also_driver_cmd.json
python3 $R/bench.py $LEAN > $OUT/bench_default_lean.json 2> $OUT/bench_default_lean.err || exit 1
# 1. kernel trace + stats of the driver's command and of the default command (hipGraph replay), headline workload only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_driver -o bench_driver -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 $LEAN > $OUT/bench_driver.json 2> $OUT/bench_driver.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_graph -o bench_graph -- python3 $R/bench.py $LEAN > $OUT/bench_graph.json 2> $OUT/bench_graph.err || exit 1
# 1b. the same for BASELINE config 3 (S=16, B=8192)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_graph_S16 -o bench_graph_S16 -- python3 $R/bench.py --dim 16 --steps 512 --warmup 64 $LEAN > $OUT/bench_graph_S16.json 2> $OUT/bench_graph_S16.err || exit 1
# 2. the same in eager mode (one ctypes launch per step)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_eager -o bench_eager -- python3 $R/bench.py --mode eager --steps 504 --warmup 56 --samples 3 $LEAN > $OUT/bench_eager.json 2> $OUT/bench_eager.err || exit 1
# 3. HBM traffic counters, one pass each (FETCH_SIZE and WRITE_SIZE do not fit one pass), eager, few steps
for wl in "4 65536" "4 131072" "4 1048576" "4 4194304" "16 8192" "16 131072" "25 4096" "25 32768" "4 33554432" "16 524288" "25 139264"; do
  set -- $wl
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_S$1_B$2_$ctr -o pmc -- python3 $R/bench.py --mode eager --steps 28 --warmup 14 --samples 2 --dim $1 --batch $2 $LEAN > $OUT/pmc_S$1_B$2_$ctr.json 2> $OUT/pmc_S$1_B$2_$ctr.err || exit 1
  done
done
fi
if [ "$PART" = "a" ]; then echo profiles part a done; exit 0; fi
if [ "$PART" = "all" ] || [ "$PART" = "b" ]; then
# 4. the generator of BASELINE config 5, plain and in a random basis
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/generator -o generator -- python3 $R/tools/prof_basis.py > $OUT/generator.log 2> $OUT/generator.err || exit 1
# 5. the matrix-core kernels under PMC counters (what bounds them: DESIGN.md section 3)
for op in gen genf many; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/mfma_${op}_p1 -o pmc -- python3 $R/tools/prof_one.py --op $op --S 25 --B 4096 --R 64 --iters 5 > $OUT/mfma_${op}_p1.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/mfma_${op}_p2 -o pmc -- python3 $R/tools/prof_one.py --op $op --S 25 --B 4096 --R 64 --iters 5 > $OUT/mfma_${op}_p2.log 2>&1 || exit 1
done
# 5b. the generator in a random basis (bench.py's valu_issue_frac of that line)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/mfma_genb_p1 -o pmc -- python3 $R/tools/prof_one.py --op gen --basis --S 25 --B 4096 --R 64 --iters 5 > $OUT/mfma_genb_p1.log 2>&1 || exit 1
# 6. the byte-streaming entries either side of the step by graph replay (the chip probes of rounds 2-3 -- launch floor, read
#    bandwidth, issue rates, shader clock, write-stream patterns -- are in the git history; their outputs stay under profiles/r0[23]_*)
python3 $R/tools/aux_time.py > $OUT/aux_ops.txt 2> $OUT/aux_ops.err || exit 1
# 9. round 3: how the rate of a VALU-bound kernel (the generator) and of the launch- / memory-bound steps moves while the
#    GPU's clocks settle after an idle gap; game strides of 16- against 128-byte multiples; the generator's phases
python3 $R/tools/gen_series.py > $OUT/generator_series.txt 2> $OUT/generator_series.err || exit 1
python3 $R/tools/step_series.py > $OUT/step_series.txt 2> $OUT/step_series.err || exit 1
fi
# 10. round 4: the resident stepper at S=4 by batch size, with / without ready words and progress; N1 / N2 fused entries
python3 $R/tools/stream_share_probe.py > $OUT/stream_share.txt 2> $OUT/stream_share.err || exit 1
python3 $R/tools/stream_share_probe.py small > $OUT/stream_small.txt 2> $OUT/stream_small.err || exit 1
python3 $R/tools/n2_probe.py > $OUT/fused_n1_n2.jsonl 2> $OUT/fused_n1_n2.err || exit 1
echo profiles done
