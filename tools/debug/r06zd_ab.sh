set -u
export TMPDIR=/tmp
mkdir -p gpurun_out; : > gpurun_out/r06zd_ab.txt
for cfg in "HSPF_RP_XCD=0" "HSPF_RP_XCD=1" "HSPF_RP_XCD=1 HSPF_RP_GX=64" "HSPF_RP_XCD=1 HSPF_RP_GX=128" "HSPF_RP_XCD=0 HSPF_RP_GX=256"; do
  echo "== $cfg" >> gpurun_out/r06zd_ab.txt
  env $cfg timeout 300 python tools/gpu_dynamic_probe.py --quick 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l)
    if d['zero_share']: print(' ', d['graph'],d['zero_share'],d['roots'],'ms',d['ms_per_run_call'],'repair',d['ms_repair'],'sweeps',d['repair_sweeps'])
" >> gpurun_out/r06zd_ab.txt
done
cat gpurun_out/r06zd_ab.txt
echo "tests xcd=1: $(timeout 500 python -m pytest tests/test_gpu_dynamic.py -m gpu -x -q 2>&1 | tail -1)"
