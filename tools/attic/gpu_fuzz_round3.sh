#!/bin/bash
# The round's randomised differential campaign (tools/gpu_fuzz.py) over five engine configurations side by side:
# product defaults, sweep engine only, everything on the wide-mask path (leaves derived in the emit), the same on
# arbitrarily patched graphs, and LANs of 150-900 routers.   usage: bash tools/gpu_fuzz_round3.sh [graphs per configuration]
set -u
N=${1:-500}
OUT=gpurun_out/fuzz_r03.txt; mkdir -p gpurun_out; : > $OUT
(echo "default:   $(python tools/gpu_fuzz.py 20000 $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "sweeps:    $(HSPF_SINGLE_MAX_N=0 HSPF_LV_MAX_ROOTS=0 python tools/gpu_fuzz.py 40000 $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "widemask:  $(HSPF_SINGLE_MAX_N=0 HSPF_LV_MAX_ROOTS=0 HSPF_VARIANT=1 python tools/gpu_fuzz.py 60000 $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "arbitrary: $(FUZZ_ARBITRARY=1 HSPF_SINGLE_MAX_N=0 HSPF_LV_MAX_ROOTS=0 HSPF_VARIANT=1 python tools/gpu_fuzz.py 80000 $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "kfused:    $(HSPF_SINGLE_MAX_N=0 HSPF_LV_MAX_ROOTS=0 HSPF_VARIANT=32768 python tools/gpu_fuzz.py 90000 $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "wide LANs: $(FUZZ_WIDE=$((N / 8)) python tools/gpu_fuzz.py 1000 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
wait
cat $OUT
