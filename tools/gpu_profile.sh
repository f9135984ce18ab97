#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel-trace stats + PMC passes -> gpurun_out/prof
# usage: bash tools/gpu_profile.sh <tag>      (summaries are then copied to profiles/<tag>_* by hand)
#        STAGE=1 bash tools/gpu_profile.sh <tag>   bench + kernel trace only        (two shorter gpurun calls instead of one:
#        STAGE=2 bash tools/gpu_profile.sh <tag>   PMC passes + traffic summary only  gpurun_out/ of both merges back)
set -u
export TMPDIR=/tmp
STAGE=${STAGE:-0}
R=$(pwd); OUT=$R/gpurun_out/prof; rm -rf $OUT; mkdir -p $OUT
if [ "$STAGE" != "2" ]; then
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err < /dev/null
tail -c 2500 $OUT/bench.json
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o spf -- python $R/bench.py --steps 10 --warmup 2 --min-timed-ms 0 --no-cpu-baseline > $OUT/trace.log 2>&1 < /dev/null
cp $OUT/trace/spf_kernel_stats.csv $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -size +2M -delete
cd $R
fi
if [ "$STAGE" = "1" ]; then exit 0; fi
cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
            "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$(echo "$pass" | md5sum | cut -c1-6)
  timeout -k 5 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$i -o q -- python $R/bench.py --steps 3 --warmup 1 --min-timed-ms 0 --no-cpu-baseline > $OUT/pmc_$i.log 2>&1 < /dev/null
done
# calibration of FETCH_SIZE / WRITE_SIZE on known byte counts in the access shapes of the engine's kernels (4 B per lane:
# 256-byte rows streamed and gathered; 16 B per lane; Infinity-Cache hits) — tools/ubench/fetch_calib.hip
if hipcc --offload-arch=gfx950 -O3 $R/tools/ubench/fetch_calib.hip -o /tmp/fetch_calib > $OUT/calib_build.log 2>&1; then
  /tmp/fetch_calib > $OUT/calib_run.txt 2>&1
  for pass in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_calib_$pass -o q -- /tmp/fetch_calib > $OUT/pmc_calib_$pass.log 2>&1 < /dev/null
  done
fi
cd $R
python - <<'PY'
import csv, glob, collections, json
# ---- calibration: bytes really moved / bytes the counter reports, per access shape
calib = {}
moved = {"calib_read4": 1 << 32, "calib_read16": 1 << 32, "calib_gather256": 1 << 32, "calib_write4": 1 << 32, "calib_read4_mall": 64 << 20}
for f in sorted(glob.glob("gpurun_out/prof/pmc_calib_*/*counter_collection.csv")):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k in moved: per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        for c, x in v.items():
            x = x[1:] if len(x) > 1 else x                      # (first launch of each: cold TLBs, first touch)
            kib = sum(x) / len(x)
            calib.setdefault(k, {})[c] = {"counter_kib": round(kib, 1), "bytes_moved": moved[k],
                                          "bytes_per_counted_byte": (round(moved[k] / (kib * 1024), 4) if kib else None)}
fetch_corr = (calib.get("calib_read4", {}).get("FETCH_SIZE", {}) or {}).get("bytes_per_counted_byte") or 2.0
gather_corr = (calib.get("calib_gather256", {}).get("FETCH_SIZE", {}) or {}).get("bytes_per_counted_byte")
write_corr = (calib.get("calib_write4", {}).get("WRITE_SIZE", {}) or {}).get("bytes_per_counted_byte") or 1.0
out = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/prof/pmc_*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "hspf" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        for c, x in v.items():
            x = sorted(x); top = x[len(x)//2:]
            out[k][c] = {"launches": len(x), "mean_all": sum(x)/len(x), "mean_top_half": sum(top)/len(top), "max": x[-1]}
json.dump(out, open("gpurun_out/prof/pmc_summary.json", "w"), indent=1)
# Fabric-side traffic per launch of the dominant kernel.  FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction
# (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-B requests as 64 B -> x2 for 16-B/lane streams.  The
# engine's kernels read 4 B per lane (one 256-byte row per wave instruction): the factor used here is the one MEASURED
# just above on that shape (calib_read4; calib_gather256 = the same rows at scattered places, recorded next to it), 2.0
# only if the calibration did not run; k_emit_fused (reads the packed state once, writes every result once) cross-checks.
traffic = {}
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        key = k.replace("hspf::", "").split("<")[0]
        nl = v["FETCH_SIZE"]["launches"]
        t = traffic.setdefault(key, {"kernel": [], "fetch_kib_per_launch": 0.0, "write_kib_per_launch": 0.0, "launches_sampled": 0, "fetch_correction": fetch_corr, "write_correction": write_corr})
        # several instantiations of one kernel (k_fused_lean's launch modes): the mean over ALL their launches
        t["fetch_kib_per_launch"] = (t["fetch_kib_per_launch"] * t["launches_sampled"] + v["FETCH_SIZE"]["mean_all"] * nl) / (t["launches_sampled"] + nl)
        t["write_kib_per_launch"] = (t["write_kib_per_launch"] * t["launches_sampled"] + v["WRITE_SIZE"]["mean_all"] * v["WRITE_SIZE"]["launches"]) / (t["launches_sampled"] + nl)
        t["launches_sampled"] += nl
        t["kernel"].append(k)
for t in traffic.values():
    t["hbm_bytes_per_launch"] = int((fetch_corr * t["fetch_kib_per_launch"] + write_corr * t["write_kib_per_launch"]) * 1024)
    t["kernel"] = ", ".join(t["kernel"])
# ONE STEP (= one 64-root run of the headline workload), all its kernels: every launch of the run's kernels in the PMC
# passes, summed, divided by the number of runs (k_init_fused is launched exactly once per run)
import os
step_kernels = ("k_fused_lean", "k_emit_fused", "k_init_fill", "k_init_fused", "k_clear_lane_flag")
runs = traffic.get("k_init_fused", {}).get("launches_sampled", 0)
if runs:
    tot = sum(traffic[k]["hbm_bytes_per_launch"] * traffic[k]["launches_sampled"] for k in step_kernels if k in traffic)
    traffic["per_step"] = {"hbm_bytes": int(tot / runs), "runs_sampled": runs,
                           "kernels": {k: {"launches_per_step": round(traffic[k]["launches_sampled"] / runs, 2),
                                           "hbm_bytes_per_step": int(traffic[k]["hbm_bytes_per_launch"] * traffic[k]["launches_sampled"] / runs)}
                                       for k in step_kernels if k in traffic},
                           "note": "fetch_correction x FETCH_SIZE + write_correction x WRITE_SIZE (KiB counters, separate --pmc passes); the corrections are measured in the same session on known byte counts in the kernels' access shape (calibration block); whether reads served by the Infinity Cache are counted is what calib_read4_mall shows (64 MB read again and again): if they are, this is fabric-side traffic, an upper bound of the HBM bytes"}
traffic["calibration"] = {"kernels": calib, "fetch_correction_used": fetch_corr, "fetch_correction_gather256": gather_corr, "write_correction_used": write_corr,
                          "source": "tools/ubench/fetch_calib.hip under rocprofv3 --pmc, same box, same session" if calib else "calibration did not run: the guide's x2"}
traffic["git_rev"] = os.environ.get("GIT_REV", "unknown")
json.dump(traffic, open("gpurun_out/prof/traffic.json", "w"), indent=1)
print(json.dumps({k: traffic[k] for k in ("per_step", "git_rev", "calibration", "k_fused_lean", "k_emit_fused") if k in traffic}, indent=1))
PY
rm -rf $OUT/pmc_*/
