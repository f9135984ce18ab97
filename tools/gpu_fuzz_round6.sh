#!/bin/bash
# Round 6's slice of the randomised differential campaign (tools/gpu_fuzz.py): the configurations of round 5 in short, plus
# the DYNAMIC lines — every graph with zero-cost router links and costs 0..4 (FUZZ_ZERO), a third of the runs with
# HSPF_RUN_POP_RANK — through the product engine, the sweep engine, k_fused, the wide-mask path, lane = vertex and k_xcd:
# hops / masks / pop ranks of roots with a dynamic pop order come from k_repair (holo_amd/csrc/spf_repair.hip.h), the
# summary line says how many roots that was and how many still went to the sequential kernel (u32 saturation only).
# usage: bash tools/gpu_fuzz_round6.sh [graphs per configuration [seed offset]]   — keep a gpurun call of this under ~100 s.
set -u
N=${1:-150}
O=${2:-0}
OUT=gpurun_out/fuzz_r06.txt; mkdir -p gpurun_out; : > $OUT
S="HSPF_SINGLE_MAX_N=0 HSPF_LV_MAX_ROOTS=0 HSPF_XCD_MAX_ROOTS=0"
(echo "default:            $(python tools/gpu_fuzz.py $((420000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "dynamic, default:   $(FUZZ_ZERO=1 python tools/gpu_fuzz.py $((440000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
wait
(echo "dynamic, sweeps:    $(env $S FUZZ_ZERO=1 python tools/gpu_fuzz.py $((460000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "dynamic, kfused:    $(env $S HSPF_VARIANT=32768 FUZZ_ZERO=1 python tools/gpu_fuzz.py $((480000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
wait
(echo "dynamic, widemask:  $(env $S HSPF_VARIANT=1 FUZZ_ZERO=1 python tools/gpu_fuzz.py $((500000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "dynamic, twophase:  $(env $S HSPF_VARIANT=64 FUZZ_ZERO=1 python tools/gpu_fuzz.py $((520000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
wait
(echo "dynamic, lanevertex: $(env HSPF_SINGLE_MAX_N=0 HSPF_LV_MAX_ROOTS=64 HSPF_LV_MIN_N=0 HSPF_XCD_MAX_ROOTS=0 FUZZ_ZERO=1 python tools/gpu_fuzz.py $((540000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "dynamic, xcd:       $(env HSPF_SINGLE_MAX_N=0 HSPF_LV_MAX_ROOTS=0 HSPF_XCD_ALWAYS=1 FUZZ_MAX_ROOTS=8 FUZZ_ZERO=1 python tools/gpu_fuzz.py $((560000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
wait
(echo "sweeps:             $(env $S python tools/gpu_fuzz.py $((580000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "mid, product:       $(FUZZ_MID=$((N / 10)) python tools/gpu_fuzz.py $((600000 + O)) 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
wait
cat $OUT
