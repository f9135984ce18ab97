set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r06zz_gpu_tests.txt
cat gpurun_out/r06zz_gpu_tests.txt
GIT_REV=5b37f56 STAGE=2 bash tools/gpu_profile.sh r06zz 2>&1 | tail -40
cp gpurun_out/prof/traffic.json gpurun_out/r06zz_traffic.json; cp gpurun_out/prof/pmc_summary.json gpurun_out/r06zz_pmc_summary.json; cp gpurun_out/prof/calib_run.txt gpurun_out/r06zz_fetch_calib_run.txt
