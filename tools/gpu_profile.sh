#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel trace + PMC passes -> gpurun_out/
set -u
export TMPDIR=/tmp
R=$(pwd)
OUT=$R/gpurun_out/prof
mkdir -p $OUT
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o spf -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o spf -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o spf -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o spf -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_l2.log 2>&1
cd $R
find $OUT -name "*.csv" | head -20
ls -la $OUT/trace/* | head
# keep only small files (kernel trace csv can be large): stats + per-kernel aggregated
find $OUT -name "*kernel_trace.csv" -size +8M -delete
