#!/bin/bash
# scratch (round 6): option p16_staged (numerators of the 256 x 256 one-pass forward through the LDS patch) -- fwd / step us, arms alternating
S=${1:-2048x8192,4096x8192,8192x8192,1024x65536,2048x65536,8192x65536,1024x32768,4096x16384}
for o in 0 1 0 1; do
  python bench_sweep.py --opt p16_staged=$o --shapes $S 2>/dev/null | python -c "
import sys, json
out = []
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    if 'B' in r: out.append('%dx%d %.1f/%.1f' % (r['B'], r['Nc'], r['fwd_bf16']['us'], r['step']['us']))
print('staged=$o', ' | '.join(out))
"
done
