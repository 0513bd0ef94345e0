import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from dpr_scale_amd.hotpath import default_kernels
kn = default_kernels()
dev = torch.device("cuda", 0)
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
for n in (8192 * 768, 262144 * 768, 2097152 * 768 // 4):
    x = torch.randn(n, device=dev)
    y = torch.empty(n, dtype=torch.bfloat16, device=dev)
    us = t(lambda: kn.cast_bf16(x, y))
    ust = t(lambda: y.copy_(x))
    assert torch.equal(y, x.to(torch.bfloat16))
    print(f"n={n}: dprhot_cast_bf16 {us:.1f} us = {6*n/us*1e-3:.0f} GB/s; torch copy_ {ust:.1f} us = {6*n/ust*1e-3:.0f} GB/s")
