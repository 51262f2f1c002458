import sys, json
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line); print(round(d['value'],1), d['ms_per_step'], d['config']['stages_ms'])
