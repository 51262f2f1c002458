timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for a in 0 1; do MI_RAST_ABLATE=$a timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate', $a, d['value'], d['config']['stages_ms'])"; done
MI_RAST_ABLATE=32 timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "debug" | tail -1
