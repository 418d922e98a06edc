#!/bin/bash
# scratch driver for one gpurun call (tests + bench); not part of the product
TAG=${1:-x}
python -m pytest tests/ -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${TAG}_tests.log
cat gpurun_out/${TAG}_tests.log
python bench.py --no-cpu-baseline --steps 300 > gpurun_out/${TAG}_bench300.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench300.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['timed_calls_ms_per_step'])
c=d.get('converged') or {}
print({k:c.get(k) for k in ('ms_per_step','value','psnr_test_mean','train_wall_s','test_views_wall_s')})
PY
