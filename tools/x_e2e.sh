#!/bin/bash
# end-to-end A/B: tools/x_e2e.sh "tuning1" "tuning2" ...  ("" = defaults); two interleaved rounds
B="python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10"
show() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], {k:(v['ms'],v['launches']) for k,v in d['kernels'].items() if 'halo' in k or 'patch' in k})
"; }
for rep in 1 2; do for t in "$@"; do echo "== tuning=[$t]"; $B ${t:+--tuning $t} 2>/dev/null | show; done; done
