#!/bin/bash
# scratch (round 6): K slices of a dQ tile in the LDS-DMA backward pair (option pair128_slices; 0 = the rule) -- bwd us per shape
S=${1:-512x8192,768x8192,1024x8192,1536x8192,2048x8192,1024x4096,2048x4096,1024x16384,2048x16384,4096x4096,512x16384}
for o in 0 2 3 4 6 8 12 16; do
  python bench_sweep.py --opt pair128_slices=$o --shapes $S 2>/dev/null | python -c "
import sys, json
out = []
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    if 'B' in r: out.append('%dx%d %.1f' % (r['B'], r['Nc'], r['bwd_pair']['us']))
print('slices=$o', ' | '.join(out))
"
done
