python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for a in 0 1; do MI_RAST_ABLATE=$a python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate', $a, d['value'], d['config']['stages_ms'])"; done
