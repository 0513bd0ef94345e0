for D in ${DIMS:-1024 512}; do
for o in "pair128=0" "pair128=1" "pair128=2"; do
  python bench_sweep.py --dim $D --opt $o --shapes ${SHAPES:-512x8192,768x8192,1024x8192,1536x8192,2048x8192,2048x16384,4096x8192,1024x16384,1024x32768,3072x8192} 2>/dev/null | python -c "
import sys, json
out = []
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    if 'B' in r: out.append('%dx%d %.1f/%.1f/%.1f' % (r['B'], r['Nc'], r['fwd_bf16']['us'], r['bwd_pair']['us'], r['step']['us']))
print('d=$D', '$o', ' | '.join(out))
"
done; done
