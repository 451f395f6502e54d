#!/bin/bash
# Profiling recipe for one round (run on the GPU box via gpurun from the repo root):
#   bash profiles/run_profiles.sh r03
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/ ; `python profiles/summarize.py <tag>` then condenses
# it into the small files committed under profiles/.
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-also"
PART=${2:-all}   # "a": steps 0-3 (bench runs, kernel traces, HBM traffic counters); "b": the rest; default both
if [ "$PART" != "b" ]; then
# 0. unprofiled: the driver's own command (what BENCH_rNN.json records) and the default command, full JSON lines
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err || exit 1
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
# 4. the generator of BASELINE config 5, plain and in a random basis
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/generator -o generator -- python3 $R/tools/prof_basis.py > $OUT/generator.log 2> $OUT/generator.err || exit 1
# 5. the matrix-core kernels under PMC counters (what bounds them: DESIGN.md section 3)
for op in gen genf many; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/mfma_${op}_p1 -o pmc -- python3 $R/tools/prof_one.py --op $op --S 25 --B 4096 --R 64 --iters 5 > $OUT/mfma_${op}_p1.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/mfma_${op}_p2 -o pmc -- python3 $R/tools/prof_one.py --op $op --S 25 --B 4096 --R 64 --iters 5 > $OUT/mfma_${op}_p2.log 2>&1 || exit 1
done
# 5b. the generator in a random basis (bench.py's valu_issue_frac of that line)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/mfma_genb_p1 -o pmc -- python3 $R/tools/prof_one.py --op gen --basis --S 25 --B 4096 --R 64 --iters 5 > $OUT/mfma_genb_p1.log 2>&1 || exit 1
# 6. the launch floor at BASELINE config 2: empty kernel / copy / step variants as hipGraphs of 2000 chained launches
hipcc -O3 -std=c++17 --offload-arch=gfx950 $R/tools/microbench_step.hip -L$R/mat_mul_amd/lib -ltensorgame -o $OUT/microbench_step > $OUT/microbench_build.log 2>&1 || exit 1
LD_LIBRARY_PATH=$R/mat_mul_amd/lib:$LD_LIBRARY_PATH $OUT/microbench_step 65536 2000 > $OUT/launch_floor.txt 2> $OUT/launch_floor.err || exit 1
rm -f $OUT/microbench_step
# 6b. the same question at BASELINE config 4's per-GPU share (131 072 games), 14 token buffers as bench.py cycles them
hipcc -O3 -std=c++17 --offload-arch=gfx950 $R/tools/s4_share_probe.hip -L$R/mat_mul_amd/lib -ltensorgame -o $OUT/s4_share_probe > $OUT/s4_share_probe_build.log 2>&1 || exit 1
LD_LIBRARY_PATH=$R/mat_mul_amd/lib:$LD_LIBRARY_PATH $OUT/s4_share_probe 131072 2000 14 > $OUT/share_floor.txt 2> $OUT/share_floor.err || exit 1
rm -f $OUT/s4_share_probe
# 7. chip probes behind DESIGN.md's bounds: read-only and copy bandwidth, issue rates, shader clock
for t in read_bw_probe issue_rate_probe shader_clock_probe; do
  hipcc -O3 --offload-arch=gfx950 $R/tools/$t.hip -o $OUT/$t > $OUT/${t}_build.log 2>&1 || exit 1
  $OUT/$t > $OUT/$t.txt 2> $OUT/$t.err || exit 1
  rm -f $OUT/$t
done
# 8. write-stream patterns behind tg_expand_i8 (plain against non-temporal stores), and the byte-streaming entries either
#    side of the step by graph replay
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result $R/tools/expand_probe.hip -o $OUT/expand_probe > $OUT/expand_probe_build.log 2>&1 || exit 1
( echo "---- 1048576 parents x 8 children ----"; $OUT/expand_probe 1048576; echo "---- 65536 parents x 8 children ----"; $OUT/expand_probe 65536 ) > $OUT/expand_probe.txt 2> $OUT/expand_probe.err || exit 1
rm -f $OUT/expand_probe
python3 $R/tools/aux_time.py > $OUT/aux_ops.txt 2> $OUT/aux_ops.err || exit 1
# 9. round 3: how the rate of a VALU-bound kernel (the generator) and of the launch- / memory-bound steps moves while the
#    GPU's clocks settle after an idle gap; game strides of 16- against 128-byte multiples; the generator's phases
python3 $R/tools/gen_series.py > $OUT/generator_series.txt 2> $OUT/generator_series.err || exit 1
python3 $R/tools/step_series.py > $OUT/step_series.txt 2> $OUT/step_series.err || exit 1
python3 $R/tools/stride_ab.py > $OUT/stride_ab.txt 2> $OUT/stride_ab.err || exit 1
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result $R/tools/expand25_probe.hip -o $OUT/expand25_probe > $OUT/expand25_probe_build.log 2>&1 || exit 1
$OUT/expand25_probe > $OUT/expand25_probe.txt 2> $OUT/expand25_probe.err || exit 1
rm -f $OUT/expand25_probe
echo profiles done
