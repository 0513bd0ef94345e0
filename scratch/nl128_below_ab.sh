#!/bin/bash
# scratch (round 6): the one-pass forward on the 128 x 128 tile against the 256 x 256 one at shapes with >= 256 tiles of 256 x 256 (run when the option was nl128_below = a bound on the 256 x 256 tile count; today: nl128_max_tiles)
S=${1:-2048x8192,1024x16384,4096x4096,2048x16384,4096x8192,1024x32768,512x32768,512x65536,8192x8192}
for o in nl128_below=256 nl128_below=100000 nl128_below=256 nl128_below=100000; do
  python bench_sweep.py --opt $o --shapes $S 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    if 'B' in r: print('$o', r['B'], r['Nc'], r['fused_forward'], 'fwd', r['fwd_bf16']['us'], 'step', r['step']['us'])
"
done
