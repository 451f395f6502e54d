#!/bin/bash
# Alternating sweep direction (sweep_index) against one direction, with and without non-temporal state loads, over the
# cache regimes: python tools/step_sizes_bench.py per variant (A/B library).  Output: gpurun_out/sweep_ab.txt
set -u
out=gpurun_out/sweep_ab.txt
mkdir -p gpurun_out
: > $out
export TG_LIB_VARIANT=ab
S4="4 262144 256 4 1048576 112 4 2097152 64 4 4194304 64 4 8388608 32"
S16="16 8192 512 16 16384 256 16 32768 128 16 65536 64 16 131072 64 16 262144 32"
S25="25 2048 256 25 4096 208 25 8192 128 25 16384 64 25 32768 64 25 65536 32"
run() { echo "== $1" >> $out; shift; env "$@" timeout -k 10 300 python tools/step_sizes_bench.py $SIZES >> $out 2>&1 || echo "FAILED" >> $out; }
SIZES="$S4";  run "S4 sweep" X=1; run "S4 one direction" TG_NO_SWEEP=1; run "S4 sweep, plain loads" TG_S4_NO_NT_LOADS=1; run "S4 one direction, plain loads" TG_NO_SWEEP=1 TG_S4_NO_NT_LOADS=1
SIZES="$S16"; run "S16 sweep" X=1; run "S16 one direction" TG_NO_SWEEP=1; run "S16 sweep, plain loads" TG_S16_NO_NT_LOADS=1; run "S16 one direction, plain loads" TG_NO_SWEEP=1 TG_S16_NO_NT_LOADS=1
SIZES="$S25"; run "S25 sweep" X=1; run "S25 one direction" TG_NO_SWEEP=1; run "S25 sweep, plain loads" TG_S25_NO_NT_LOADS=1; run "S25 one direction, plain loads" TG_NO_SWEEP=1 TG_S25_NO_NT_LOADS=1
cat $out | cut -c1-150
