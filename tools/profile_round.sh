#!/bin/bash
# One profiling pass on the GPU box (run through gpurun from the repository root):  bash tools/profile_round.sh <tag>
# kernel-trace statistics of the default bench and of the slices workload, then the two PMC passes for tools/pmc_traffic.py.
# rocprofv3 is given the program itself after `--` (python3 ...), never a wrapper; counters and traces are separate runs.
set -o pipefail
TAG=${1:-r02}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/admm -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_admm_under_rocprof.json 2> $OUT/admm.err || echo "admm trace failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/slices -- python3 $R/bench.py --workload slices --steps 20 --warmup 2 --no-cpu-baseline > $OUT/bench_slices_under_rocprof.json 2> $OUT/slices.err || echo "slices trace failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/prof_net.py 1 3 > $OUT/pmc_fetch.log 2>&1 || echo "pmc fetch failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/tools/prof_net.py 1 3 > $OUT/pmc_write.log 2>&1 || echo "pmc write failed"
cd $R
find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls -R $OUT | head -40
