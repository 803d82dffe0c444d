mkdir -p gpurun_out/r05u
for i in 1 2 3 4 5 6; do
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r05u/run$i.json 2> gpurun_out/r05u/run$i.err || echo "run $i failed"
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r05u/run$i.json") if l.startswith("{")][0]); s=d["slices"]; print("run", $i, d["value"], s["value"], s["seconds"], s["launch_seconds_rank0"])
PY
done
