#!/bin/bash
# A/B/... of several builds of the library on ONE box: bench (no CPU baseline), default first, twice round.
# usage (gpurun): bash tools/debug/abc_lib.sh <tag> lib1.so [lib2.so ...]
set -u
TAG=$1; shift
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
  for L in "$@"; do
    b=$(basename $L .so)
    cp $L holo_amd/libholo_spf_hip.so
    python bench.py --no-cpu-baseline > $OUT/${b}_$rep.json 2> $OUT/${b}_$rep.err; pick $OUT/${b}_$rep.json
  done
done
cp /tmp/lib_default.so holo_amd/libholo_spf_hip.so
