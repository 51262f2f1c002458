#!/bin/bash
# quick check of a build: a slice of the GPU suite + default / --equal-runs lines of cfg3 and cfg3s
out=gpurun_out/${1:-q}; mkdir -p $out
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_abi.py -m gpu -q -x > $out/pytest.log 2>&1; tail -2 $out/pytest.log
for cfg in cfg3 cfg3s; do for fl in "" "--equal-runs"; do
  python bench.py --config $cfg --steps 30 --warmup 3 --settle 1 --dist-blocks 0 --sustained-seconds 0 --no-cpu-baseline $fl 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$cfg $fl', d['value'], d['ms_per_step'], d['config']['stages_ms'])"
done; done
