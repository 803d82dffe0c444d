#!/bin/bash
# One profiling pass on the GPU box (run through gpurun from the repository root):  bash tools/profile_round.sh <tag>
# kernel-trace statistics of the DRIVER's command (bench.py --steps 20 --warmup 5: ADMM it/s + the slices object + CPU baseline), then the PMC
# passes for tools/pmc_traffic.py (one slice and a 15-slice batch) and the matrix-core / clock pass of tools/pmc_conv.sh.
# rocprofv3 is given the program itself after `--` (python3 ...), never a wrapper; counters and traces are separate runs.
set -o pipefail
TAG=${1:-r06}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/driver -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command_under_rocprof.json 2> $OUT/driver.err || echo "driver-command trace failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/prof_net.py 1 3 > $OUT/pmc_fetch.log 2>&1 || echo "pmc fetch failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/tools/prof_net.py 1 3 > $OUT/pmc_write.log 2>&1 || echo "pmc write failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch15 -- python3 $R/tools/prof_net.py 15 2 > $OUT/pmc_fetch15.log 2>&1 || echo "pmc fetch (batch) failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write15 -- python3 $R/tools/prof_net.py 15 2 > $OUT/pmc_write15.log 2>&1 || echo "pmc write (batch) failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch30 -- python3 $R/tools/prof_net.py 30 2 > $OUT/pmc_fetch30.log 2>&1 || echo "pmc fetch (batch 30) failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write30 -- python3 $R/tools/prof_net.py 30 2 > $OUT/pmc_write30.log 2>&1 || echo "pmc write (batch 30) failed"
cd $R
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_conv_traffic.txt || echo "pmc_traffic failed"
python3 tools/pmc_traffic.py --batch 15 $OUT/pmc_fetch15 $OUT/pmc_write15 $OUT/pmc_conv_traffic_batch15.txt || echo "pmc_traffic (batch) failed"
python3 tools/pmc_traffic.py --batch 30 $OUT/pmc_fetch30 $OUT/pmc_write30 $OUT/pmc_conv_traffic_batch30.txt || echo "pmc_traffic (batch 30) failed"
cp profiles/conv_traffic.json profiles/conv_traffic_batch15.json profiles/conv_traffic_batch30.json $OUT/ 2>/dev/null
bash tools/pmc_conv.sh 1 gpurun_out/prof_$TAG/pmc_conv_mfma_busy_single.txt < /dev/null > /dev/null 2>&1 || echo "pmc_conv 1 failed"
bash tools/pmc_conv.sh 15 gpurun_out/prof_$TAG/pmc_conv_mfma_busy_batch15.txt < /dev/null > /dev/null 2>&1 || echo "pmc_conv 15 failed"
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +5M -delete
ls -R $OUT | head -60
