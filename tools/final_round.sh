#!/bin/bash
# Round-6 end-of-round evidence on ONE box (through gpurun from the repository root):  bash tools/final_round.sh <tag>
#   1. the GPU test suite   2. the driver's command   3. the default run (100 iterations)   4. tools/profile_round.sh (kernel statistics of the driver's
#   command, PMC traffic passes at 1 / 15 / 30 slices, matrix-core busy)   5. PMC traffic of the LSQR iteration kernels
set -o pipefail
TAG=${1:-r06}
R=$PWD; OUT=$R/gpurun_out/final_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver.err || echo "driver bench failed"
python3 bench.py --no-slices --no-secondary --no-cold-start > $OUT/bench_default_steps100.json 2> $OUT/bench_default.err || echo "default bench failed"
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1 || echo "smoke failed"
bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1 || echo "profile round failed"
bash tools/pmc_xupdate.sh $TAG > $OUT/pmc_xupdate.log 2>&1 || echo "pmc xupdate failed"
cp profiles/xupdate_traffic.json $OUT/xupdate_traffic_configs_only.json 2>/dev/null
for f in $OUT/bench_driver_command.json $OUT/bench_default_steps100.json; do python3 -c "
import json
d=json.load(open('$f')); r=d.get('roofline') or {}; s=d.get('slices') or {}
print('$f'.split('/')[-1], '| it/s', d.get('value'), '| ms/step', d.get('ms_per_step'), '| frac', r.get('frac'), '| slices/s', s.get('value'), '| epi', (d.get('epi_batch15') or {}).get('value'), '| cut0', (d.get('cut0') or {}).get('value'))
"; done
cat $OUT/smoke.txt | tail -1
