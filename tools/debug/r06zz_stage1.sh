set -u
export TMPDIR=/tmp
GIT_REV=5b37f56 STAGE=1 bash tools/gpu_profile.sh r06zz 2>&1 | tail -5
cp gpurun_out/prof/bench.json gpurun_out/r06zz_bench.json; cp gpurun_out/prof/bench.err gpurun_out/r06zz_bench.err; cp gpurun_out/prof/kernel_stats.csv gpurun_out/r06zz_kernel_stats.csv
for i in 1 2 3; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('repeat', d['value'], d['pipeline']['one_at_a_time']['runs_per_s'], d['roofline']['frac'], d['roofline']['kernel_frac'])
"; done > gpurun_out/r06zz_bench_repeats.txt
cat gpurun_out/r06zz_bench_repeats.txt
