#!/bin/bash
S=${1:-256x8192,384x8192,512x8192,768x8192,1024x8192,2048x8192,512x16384,768x16384,1024x16384,256x32768,512x32768,256x65536,512x65536,1024x4096,1536x4096,2048x4096,3072x4096,192x8192,160x4096}
for o in pair128=0 pair128=2; do
  python bench_sweep.py --opt $o --shapes $S 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    if 'B' in r: print('$o', r['B'], r['Nc'], 'bwd', r['bwd_pair']['us'], 'step', r['step']['us'])
"
done
