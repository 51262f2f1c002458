#!/bin/bash
out=gpurun_out/${1:-q}; mkdir -p $out
timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_zz_reference_pin.py -m gpu -q -x -k "not full_size or cfg5" > $out/pytest.log 2>&1; tail -2 $out/pytest.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config']['stages_ms'])"; }
B="--steps 30 --warmup 3 --settle 1 --dist-blocks 0 --sustained-seconds 0 --no-cpu-baseline"
for cfg in cfg5 cfg3 cfg5s; do python bench.py --config $cfg $B 2>/dev/null | line "$cfg"; done
