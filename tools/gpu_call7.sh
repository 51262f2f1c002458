#!/bin/bash
bash tools/gpu_call.sh $1 > /dev/null 2>&1; tail -3 gpurun_out/$1/pytest.log
for cfg in cfg3 cfg5 cfg2 cfg3s; do python - <<EOF
import json
d=json.loads(open("gpurun_out/$1/bench_$cfg.json").read().strip().splitlines()[-1]); print("$cfg", d["value"], d["ms_per_step"], d["config"]["stages_ms"])
EOF
done
for cfg in cfg3 cfg3s cfg5; do
  python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --equal-runs 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$cfg equal-runs', d['value'], d['ms_per_step'], d['config']['stages_ms'])"
done
