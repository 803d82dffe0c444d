#!/bin/bash
# Final validation of a round on the GPU box (through gpurun, from the repository root):  bash tools/final_round.sh <tag>
# full GPU test-suite, the default bench line (with CPU baseline and parity), the slices bench line, the dictionary bench.
TAG=${1:-r02_i}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout -k 10 400 python bench.py > $OUT/bench_default_steps100.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default_steps100.json | head -c 300; echo
timeout -k 10 400 python bench.py --workload slices > $OUT/bench_slices_default.json 2> $OUT/bench_slices.err
timeout -k 10 200 python tools/bench_dict.py > $OUT/bench_dict.json 2> $OUT/bench_dict.err
python - <<PY
import json
for f in ("bench_default_steps100","bench_slices_default","bench_dict"):
    try:
        d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d.get("value"), d.get("unit"), (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("avg_launch_us"), d.get("parity"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
