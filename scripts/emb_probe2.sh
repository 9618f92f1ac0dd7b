#!/bin/bash
# development: does the embedded VGICP line depend on what main() ran before it?
show() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
e=d['extra']['vgicp']; print('$2', round(e['ms_per_step'],4), round(e['roofline']['target_prep_ms'],4))"; }
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/e1.json 2>/dev/null; show gpurun_out/e1.json "steps 2:"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --map-points 200000 > gpurun_out/e2.json 2>/dev/null; show gpurun_out/e2.json "map 200k:"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --scans 2 > gpurun_out/e3.json 2>/dev/null; show gpurun_out/e3.json "scans 2:"
