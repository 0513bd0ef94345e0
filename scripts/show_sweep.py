import sys, json
src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
for l in src:
    if not l.startswith('{'):
        continue
    d = json.loads(l)
    ff = {'logits stored': 'S', 'one pass, 256 x 256 tile': 'P256', 'one pass, 128 x 128 tile': 'P128'}.get(d.get('fused_forward'), '')
    out = 'B=%5d Nc=%6d %-9s %-4s |' % (d['B'], d['Nc'], d.get('forward_plan', ''), ff)
    for k, nm in (('prep', 'prep'), ('sim_stats_f32', 'simf32'), ('sim_stats_bf16', 'sim'), ('softmax_finish', 'fin'), ('dscores', 'dsc'), ('fwd_bf16', 'fwd'), ('bwd_pair', 'bwd'),
                  ('step', 'STEP')):
        if k in d:
            out += ' %s %7.1f |' % (nm, d[k]['us'])
    if 'hbm_floor_us' in d:
        out += ' floor %6.1f  step/floor %.2f' % (d['hbm_floor_us'], d['hbm_floor_us'] / d['step_us'])
    print(out)
