timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['stages_ms'])"
timeout 200 python bench.py --config cfg5 --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['stages_ms'])"
