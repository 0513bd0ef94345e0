import sys, json
for l in open(sys.argv[1] if len(sys.argv) > 1 else '/root/repo/gpurun_out/sweep.jsonl'):
    d = json.loads(l)
    out = 'B=%5d Nc=%6d |' % (d['B'], d['Nc'])
    for k in ['sim_stats_f32', 'prep', 'sim_stats_bf16', 'softmax_finish', 'bwd_pair']:
        if k in d:
            out += ' %s %8.1fus %6.1fTF %5.0fGB/s |' % (k.replace('sim_stats_', 'sim').replace('softmax_finish', 'gfin').replace('bwd_pair', 'bwd'), d[k]['us'], d[k]['TFLOPs'], d[k]['GBps'])
    print(out)
