set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_graph_build.py tests/test_gpu_keyed_upload.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r06zb_build_tests.txt
HSPF_PATCH_TIMING=1 python tools/gpu_patch_probe.py 14 > gpurun_out/r06zb_patch_probe.txt 2>&1
R=$(pwd); cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06zb_patch_trace -o p -- python $R/tools/gpu_patch_probe.py 14 > $R/gpurun_out/r06zb_trace.log 2>&1
cd $R
cat gpurun_out/r06zb_build_tests.txt; grep -v "hspf patch\|hspf build" gpurun_out/r06zb_patch_probe.txt; grep "kb_pa_shift\|kb_scatter\|kb_pa_rows" gpurun_out/r06zb_patch_trace/p_kernel_stats.csv | cut -d, -f2-8 
