#!/bin/bash
# scratch (round 6): the backward pair on the LDS-DMA 128 x 128 tile (option pair128: 0 off, 1 rule, 2 everywhere) -- bwd / step us per shape
S=${1:-512x2048,1024x2048,2048x2048,512x4096,1024x4096,1536x4096,2048x4096,3072x4096,256x4096,256x8192,384x8192,512x8192,768x8192,1024x8192,2048x8192,256x16384,1024x16384}
for r in 1 2; do for o in pair128=0 pair128=2; do
  python bench_sweep.py --opt $o --shapes $S 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    if 'B' in r: print('$o', r['B'], r['Nc'], 'bwd', r['bwd_pair']['us'], 'step', r['step']['us'])
"
done; done
