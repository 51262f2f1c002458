for rep in 1 2; do
for v in on off; do
  if [ $v = off ]; then export MI_RAST_NO_PREZERO=1; else unset MI_RAST_NO_PREZERO; fi
  python bench.py --no-cpu-baseline --steps 40 --warmup 5 --settle 1 $1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('prezero $v', d['value'], d['ms_per_step'], d['timing']['ms_per_step_median'], d['config']['stages_ms'])"
done
done
