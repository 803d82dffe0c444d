#!/bin/bash
set -o pipefail
R=$PWD; OUT=$R/gpurun_out/${1:-r6c}; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_net.py -x -q -m gpu > $OUT/pytest_net.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_net.log
tail -5 $OUT/pytest_net.log
bash tools/ab_knob.sh conv_deepxcd 50 0 1 > $OUT/ab_conv_deepxcd.txt 2>&1; cat $OUT/ab_conv_deepxcd.txt
bash tools/ab_knob.sh ew_norm_mult 50 1 2 4 > $OUT/ab_ew_norm_mult.txt 2>&1; cat $OUT/ab_ew_norm_mult.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-slices --no-secondary --no-cpu-baseline > $OUT/bench_cold.json 2> $OUT/bench_cold.err || echo bench failed
python3 -c "
import json; d=json.load(open('$OUT/bench_cold.json')); print(d['value'], json.dumps(d['cold_start']['setup_ms']), d['cold_start']['time_to_first_slice_s'])"
