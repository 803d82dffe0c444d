#!/bin/bash
set -o pipefail
R=$PWD; OUT=$R/gpurun_out/${1:-r6d}; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout -k 10 600 python3 tools/provoke_straggler.py > $OUT/provoke_straggler.txt 2> $OUT/provoke.err || echo provoke failed
cat $OUT/provoke_straggler.txt | cut -c1-1500
