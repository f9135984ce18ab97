#!/bin/bash
# A/B of two builds of the library on ONE box: bench (no CPU baseline) default, experimental, default, experimental.
# usage (gpurun): bash tools/debug/ab_lib.sh holo_amd/libholo_spf_hip_exp.so <tag>
set -u
EXP=$1; TAG=${2:-ab}
R=$(pwd); OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cp holo_amd/libholo_spf_hip.so /tmp/lib_default.so
pick() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[1].split("/")[-1], "in flight", d["value"], "one at a time", d["pipeline"]["one_at_a_time"]["runs_per_s"], "passes", d["phases_ms_per_step"].get("dense_passes"),
              "rows_x_N", d["phases_ms_per_step"]["rows_x_N"], "avg launch us", d["roofline"]["avg_launch_us"], "launches", d["roofline"]["launches_per_step"], "verified", d["verified_roots"])
PY
}
for rep in 1 2; do
  cp /tmp/lib_default.so holo_amd/libholo_spf_hip.so
  python bench.py --no-cpu-baseline > $OUT/default_$rep.json 2> $OUT/default_$rep.err; pick $OUT/default_$rep.json
  cp $EXP holo_amd/libholo_spf_hip.so
  python bench.py --no-cpu-baseline > $OUT/exp_$rep.json 2> $OUT/exp_$rep.err; pick $OUT/exp_$rep.json
done
cp $EXP holo_amd/libholo_spf_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "isis_100k or random_lsdb_normal or lean" 2>&1 | tail -3
cp /tmp/lib_default.so holo_amd/libholo_spf_hip.so
