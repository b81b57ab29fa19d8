#!/bin/bash
mkdir -p gpurun_out/r5d
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r5d/gputests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r5d/gputests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5d/bench.json 2> gpurun_out/r5d/bench.err
echo "bench rc=$?"; python - <<'P'
import json
l = [x for x in open('gpurun_out/r5d/bench.json') if x.startswith('{')]
d = json.loads(l[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('roofline', {}).get('step'))
P
