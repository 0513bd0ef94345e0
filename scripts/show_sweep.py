import sys, json
src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
for l in src:
    if not l.startswith('{'):
        continue
    d = json.loads(l)
    out = 'B=%5d Nc=%6d %s |' % (d['B'], d['Nc'], d.get('forward_plan', ''))
    for k in ['sim_stats_f32', 'prep', 'sim_stats_bf16', 'softmax_finish', 'dscores', 'bwd_pair']:
        if k in d:
            out += ' %s %8.1fus %6.1fTF %5.0fGB/s |' % (k.replace('sim_stats_', 'sim').replace('softmax_finish', 'fin').replace('bwd_pair', 'bwd'), d[k]['us'], d[k]['TFLOPs'], d[k]['GBps'])
    print(out)
