#!/bin/bash
# One gpurun call of the round: GPU test suite + bench lines of the four configurations.  Everything lands in gpurun_out/<tag>/.
tag=${1:-c1}; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout ${TEST_TIMEOUT:-600} python -m pytest tests -m gpu -q --tb=short -x ${PYTEST_ARGS:-} > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
  tail -15 $out/pytest.log
fi
for cfg in ${CFGS:-cfg3 cfg5 cfg2 cfg3s}; do
  timeout 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > $out/bench_$cfg.json 2> $out/bench_$cfg.err
  python - <<EOF
import json
try:
    d=json.loads(open("$out/bench_$cfg.json").read().strip().splitlines()[-1])
    print("$cfg", d["value"], d["unit"], d["ms_per_step"], d["config"].get("stages_ms"), "frac", d["roofline"]["frac"])
except Exception as e:
    print("$cfg FAILED", e); print(open("$out/bench_$cfg.err").read()[-1500:])
EOF
done
