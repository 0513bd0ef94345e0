#!/bin/bash
S=${1:-256x8192,384x8192,448x8192,512x8192,256x16384,384x16384,448x16384,256x32768,384x32768,256x65536,512x16384,128x16384,192x8192}
for o in pair128=0 pair128=2 pair128=2,pair128_cap=32 pair128=2,pair128_cap=8; do
  python bench_sweep.py --opt $o --shapes $S 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    if 'B' in r: print('$o', r['B'], r['Nc'], 'bwd', r['bwd_pair']['us'], 'step', r['step']['us'])
"
done
