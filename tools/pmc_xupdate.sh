#!/bin/bash
# PMC traffic of the LSQR iteration kernels (bench.py: xupdate.roofline.traffic), one configuration per pass pair:  bash tools/pmc_xupdate.sh <tag>
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (they do not fit one), the program itself after `--`, no trace domains beside --kernel-trace.
set -o pipefail
TAG=${1:-r06}
R=$PWD
OUT=$R/gpurun_out/pmcx_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for cfg in 0 2 3 4; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/c${cfg}_$ctr -- python3 $R/tools/xupdate_times.py $cfg > $OUT/c${cfg}_$ctr.log 2>&1 || echo "pass $cfg $ctr failed"
  done
done
cd $R
python3 tools/pmc_xupdate.py $OUT $OUT/pmc_xupdate_traffic.txt || echo "pmc_xupdate.py failed"
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +5M -delete
