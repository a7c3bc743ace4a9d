# A/B helper: end-to-end bench.py under different y3_set_tuning knobs ("tuning:streams" pairs as arguments)
for spec in "$@"; do
  t="${spec%%:*}"; st="${spec##*:}"
  echo "== tuning=[$t] streams=$st"
  timeout 200 python bench.py --no-cpu-baseline --steps 30 --warmup 8 --streams $st ${t:+--tuning $t} 2>&1 | grep -v amdgpu | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'], d['roofline'].get('all_kernels_ms_per_step'))
    elif l: print(l[:200])
"
done
