#!/bin/bash
# The randomised differential campaign (tools/gpu_fuzz.py) of round 5: every run is checked twice, through hspf_run (four
# arrays) and through hspf_run_packed (ABI 7), under every way the lean sweep's launches can decide — product
# thresholds, dense stretch from the first sweep on, stretches that stop after their second pass, multi-pass launches on
# graphs so small that all passes run side by side — next to the older paths (k_fused everywhere, fused path off, LANs of
# 150-900 routers).        usage: bash tools/gpu_fuzz_round5.sh [graphs per configuration [seed offset]]
# Two processes at a time, and a gpurun call of this kept WELL under two minutes (300 graphs per configuration since the
# lanevertex and mid-size lines joined: ~70 s): round 4 lost two boxes under longer calls, round 5 one more (600 graphs,
# box gone after 120 s: profiles/r05_notes.md) — whatever it is on the box's side, it is a function of the call's length.
set -u
N=${1:-400}
O=${2:-0}          # added to every configuration's first seed: another slice of the seed space per call
OUT=gpurun_out/fuzz_r05.txt; mkdir -p gpurun_out; : > $OUT
S="HSPF_SINGLE_MAX_N=0 HSPF_LV_MAX_ROOTS=0 HSPF_XCD_MAX_ROOTS=0"
(echo "default:        $(python tools/gpu_fuzz.py $((120000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "sweeps:         $(env $S python tools/gpu_fuzz.py $((140000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
wait
(echo "dense at once:  $(env $S HSPF_DENSE_PCT=0 HSPF_LEAN_HEAD=1 python tools/gpu_fuzz.py $((160000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "early stop:     $(env $S HSPF_DENSE_STAY_PCT=95 python tools/gpu_fuzz.py $((180000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
wait
(echo "side by side:   $(env $S HSPF_DENSE_MIN_WGS=1 HSPF_DENSE_PASSES=4 HSPF_DENSE_PCT=5 python tools/gpu_fuzz.py $((200000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "never dense:    $(env $S HSPF_DENSE_PCT=100000 python tools/gpu_fuzz.py $((220000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
wait
(echo "kfused:         $(env $S HSPF_VARIANT=32768 python tools/gpu_fuzz.py $((240000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "widemask:       $(env $S HSPF_VARIANT=1 python tools/gpu_fuzz.py $((260000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
wait
(echo "xcd:            $(env HSPF_SINGLE_MAX_N=0 HSPF_LV_MAX_ROOTS=0 HSPF_XCD_ALWAYS=1 FUZZ_MAX_ROOTS=8 python tools/gpu_fuzz.py $((280000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "wide LANs:      $(FUZZ_WIDE=$((N / 8)) python tools/gpu_fuzz.py 3000 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "lanevertex:     $(env HSPF_SINGLE_MAX_N=0 HSPF_LV_MAX_ROOTS=64 HSPF_LV_MIN_N=0 HSPF_XCD_MAX_ROOTS=0 python tools/gpu_fuzz.py $((340000 + O)) $N 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
wait
(echo "mid, product:   $(FUZZ_MID=$((N / 10)) python tools/gpu_fuzz.py $((300000 + O)) 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
(echo "mid, k_xcd:     $(HSPF_XCD_ALWAYS=1 FUZZ_MID=$((N / 10)) python tools/gpu_fuzz.py $((320000 + O)) 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $OUT) &
wait
cat $OUT
