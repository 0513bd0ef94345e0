import sys, torch
sys.path.insert(0, "/root/repo")
from dpr_scale_amd.hotpath import HipKernels
B, K, d, stage = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dev = torch.device("cuda", 0)
kn = HipKernels()
g = torch.Generator().manual_seed(0)
n = B * K
q = torch.randn(B, d, generator=g).to(dev); c = torch.randn(n, d, generator=g).to(dev)
y = (torch.arange(B) * K).to(dev); m = torch.zeros(n, dtype=torch.uint8, device=dev)
Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev); Cb = torch.empty((n, d), dtype=torch.bfloat16, device=dev)
kn.prep(q, Qb, c, Cb); torch.cuda.synchronize(); print("prep ok", flush=True)
if stage == "sim":
    S = kn.sim(Qb, Cb, m, 1.0); torch.cuda.synchronize(); print("sim ok", flush=True)
if stage == "fwd":
    r = kn.inbatch_fwd(Qb, Cb, y, 0, m, 1.0, 1.0 / B); torch.cuda.synchronize(); print("fwd ok", flush=True)
if stage == "fwd32":
    r = kn.inbatch_fwd_f32(q, c, Qb, Cb, y, 0, m, 1.0, 1.0 / B); torch.cuda.synchronize(); print("fwd32 ok", flush=True)
if stage == "bwd":
    G = (torch.randn(B, n, generator=g) * 0.01).to(torch.bfloat16).to(dev)
    one = torch.ones(1, device=dev)
    r = kn.inbatch_bwd(G, Qb, Cb, 1.0, one); torch.cuda.synchronize(); print("bwd ok", flush=True)
if stage == "step":
    r = kn.inbatch_step_f32(q, c, Qb, Cb, y, 0, m, 1.0, 1.0 / B); torch.cuda.synchronize(); print("step ok", flush=True)
