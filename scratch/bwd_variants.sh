#!/bin/bash
# scratch (round 6): the backward of the few-rows big shapes under each plan switch; prints bwd_pair / splitk / step us per shape
S=${1:-256x8192,384x8192,512x8192}
for o in "" no_8pb=1 no_big_bwd=1 unfused_bwd=1 dq_cap_few=8 dq_cap_few=32; do
  python bench_sweep.py ${o:+--opt $o} --shapes $S 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    if 'B' in r: print('${o:-default}'.ljust(16), r['B'], r['Nc'], 'bwd', r['bwd_pair']['us'], 'step', r['step']['us'])
"
done
