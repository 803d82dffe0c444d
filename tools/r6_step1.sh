#!/bin/bash
set -o pipefail
R=$PWD; OUT=$R/gpurun_out/${1:-r6b}; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_admm.py tests/test_gpu_mex.py -x -q -m gpu -k "health or eight_workers or two_workers or single_slice_commands" > $OUT/pytest_new.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_new.log
tail -5 $OUT/pytest_new.log
bash tools/pmc_xupdate.sh ${1:-r6b} > $OUT/pmc_xupdate.log 2>&1 || echo pmc failed
python3 tools/lsqr_persist_stamps.py > $OUT/lsqr_persist_stamps.txt 2>&1 || echo stamps failed
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err || echo bench failed
tail -c 300 $OUT/bench_driver.json
