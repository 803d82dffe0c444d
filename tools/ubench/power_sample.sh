#!/bin/bash
# Sample socket power and shader clock (rocm-smi) while the batched network forward runs.  Usage: power_sample.sh <B> <reps> <out>
python3 tools/prof_net.py $1 $2 > /dev/null 2>&1 &
BP=$!
for i in $(seq 1 40); do
  echo "t=$i $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Package Power|sclk' | sed 's/.*: //' | tr '\n' ' ')" >> $3
  sleep 1
  kill -0 $BP 2>/dev/null || break
done
wait $BP
