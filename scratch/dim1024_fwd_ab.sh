for o in nl128_max_tiles=512 nl128_max_tiles=511 p16_staged=0 p16_staged=2; do
  python bench_sweep.py --dim 1024 --opt $o --shapes 1024x8192,2048x4096,512x16384,4096x8192,8192x8192 2>/dev/null | python -c "
import sys, json
out = []
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    if 'B' in r: out.append('%dx%d %s %.1f/%.1f' % (r['B'], r['Nc'], r['fused_forward'][-11:], r['fwd_bf16']['us'], r['step']['us']))
print('d=1024', '$o', ' | '.join(out))
"
done
