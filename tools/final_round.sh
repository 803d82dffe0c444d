#!/bin/bash
# End-of-round measurements on ONE box (run through gpurun from the repository root):  bash tools/final_round.sh <tag>
# bench lines (default, driver-like 20 steps, slices per GPU, 120-slice fixed total), then the kernel statistics under rocprofv3.
TAG=${1:-r03}
O=gpurun_out/final_$TAG
mkdir -p $O
python bench.py > $O/bench_default_steps100.json 2> $O/bench_default.err || echo "default bench failed"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_steps20.json 2>> $O/bench_default.err || echo "20-step bench failed"
python bench.py --workload slices --no-cpu-baseline > $O/bench_slices_default.json 2> $O/bench_slices.err || echo "slices bench failed"
python bench.py --workload slices --total-slices 120 --no-cpu-baseline > $O/bench_slices_total120.json 2>> $O/bench_slices.err || echo "120-slice bench failed"
python tools/bench_dict.py > $O/bench_dict.json 2> $O/bench_dict.err || echo "dict bench failed"
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/admm -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/$O/bench_admm_under_rocprof.json 2> $R/$O/admm.err || echo "admm trace failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/slices -- python3 $R/bench.py --workload slices --steps 20 --warmup 2 --no-cpu-baseline > $R/$O/bench_slices_under_rocprof.json 2> $R/$O/slices.err || echo "slices trace failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/dict -- python3 $R/tools/bench_dict.py > $R/$O/bench_dict_under_rocprof.json 2> $R/$O/dict.err || echo "dict trace failed"
cd $R
find $O -name "*kernel_trace.csv" -size +10M -delete
for f in $O/bench_*.json; do echo "== $f"; python -c "
import json,sys
d=json.load(open('$f'))
r=d.get('roofline') or {}
print(d.get('metric','')[:60], '| value', d.get('value'), d.get('unit'), '| ms/step', d.get('ms_per_step'), '| frac', r.get('frac'), '| launch us', r.get('avg_launch_us'), '| scaling', d.get('scaling'))
" 2>/dev/null; done
