# Same-box end-to-end A/B of y3_set_tuning knob sets: tools/ab_e2e.sh <reps> <tuning A> <tuning B> ...   ("-" = defaults)
reps=$1; shift
for rep in $(seq $reps); do
  for t in "$@"; do
    if [ "$t" = "-" ]; then targ=""; else targ="--tuning $t"; fi
    timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 80 --warmup 10 $targ 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-24s value %8.1f  ms/step %.4f  dominant %7.1f TF  resident %s' % ('$t', d['value'], d['ms_per_step'], d['roofline']['achieved'], d.get('resident')))
"
  done
done
