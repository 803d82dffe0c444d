#!/bin/bash
# round-6 baseline on one box: the driver's command, then a kernel trace of the headline loop (one iteration launch by launch) and of one forward
set -o pipefail
R=$PWD; OUT=$R/gpurun_out/${1:-r6a}; mkdir -p $OUT; export TMPDIR=/tmp
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err || echo bench failed
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/it -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline --no-slices --no-secondary > $OUT/it_bench.json 2> $OUT/it.err || echo it trace failed
rocprofv3 --kernel-trace --output-format csv -d $OUT/lt -- python3 $R/tools/prof_net.py 1 6 > $OUT/lt.log 2>&1 || echo lt trace failed
cd $R
python3 tools/iter_times.py $OUT/it > $OUT/iter_times.txt 2>&1
python3 tools/iter_times.py $OUT/it 3 > $OUT/iter_times_early.txt 2>&1
python3 tools/layer_times.py $OUT/lt > $OUT/layer_times.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
tail -c 600 $OUT/bench_driver.json
