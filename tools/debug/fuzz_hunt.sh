#!/bin/bash
# Hunt for the intermittent 'hops' mismatch of the dynamic-order repair (round 6): the "dynamic, kfused" line of tools/gpu_fuzz_round6.sh,
# P processes side by side, R rounds, mismatches dumped (FUZZ_DUMP).   usage: bash tools/debug/fuzz_hunt.sh [rounds [procs [extra env]]]
R=${1:-4}; P=${2:-3}
S="HSPF_SINGLE_MAX_N=0 HSPF_LV_MAX_ROOTS=0 HSPF_XCD_MAX_ROOTS=0 FUZZ_ZERO=1 FUZZ_DUMP=1 ${3:-HSPF_VARIANT=32768}"
mkdir -p gpurun_out; : > gpurun_out/fuzz_hunt.txt
for r in $(seq 1 $R); do
  for p in $(seq 1 $P); do
    (env $S timeout 120 python tools/gpu_fuzz.py $((487000 + 1000 * ((r * P + p) % 7))) 120 2>&1 | grep -v amdgpu.ids | grep -v "fuzz_layout\|fuzz_routes" | cut -c1-1200 >> gpurun_out/fuzz_hunt.txt) &
  done
  wait
done
cat gpurun_out/fuzz_hunt.txt
