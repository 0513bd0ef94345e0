"""Memory safety of the library's launches, by guard bands: every device buffer the host side hands to libdprhot -- outputs and the
workspace, sized EXACTLY as the ABI says (dprhot_workspace_bytes & co.) -- sits between two 4 KiB bands of a byte pattern; after the
calls the bands must be intact.  Shapes on both sides of every plan boundary (fused small step, few-rows plan with and without the
dScores launch, the 256 x 256 GEMM plans, the no-logits plan), ragged sizes, hidden sizes that need padding, the packed multi-rank
step in both wire formats, the streaming top-k in its three forms and the retrieval loop.  (An out-of-bounds candidate write in an
option that has since been removed was found by review, not by a test: ADVICE r4.)"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

GUARD = 4096
PATTERN = 0xA5


class _Guards:
    """torch.empty for HIP tensors, replaced: the tensor is the interior of a larger byte buffer filled with PATTERN."""

    def __init__(self):
        self.orig = torch.empty
        self.regions = []

    def empty(self, *size, dtype=None, device=None, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        dev = torch.device(device) if device is not None else None
        size = tuple(int(x) for x in size)
        if dev is None or dev.type != "cuda" or kw:
            return self.orig(size, dtype=dtype, device=device, **kw)
        dt = dtype if dtype is not None else torch.get_default_dtype()
        n = int(math.prod(size)) * self.orig((), dtype=dt).element_size()
        if n == 0:
            return self.orig(size, dtype=dtype, device=device)
        raw = torch.full((n + 2 * GUARD,), PATTERN, dtype=torch.uint8, device=dev)
        self.regions.append((raw, n))
        return raw[GUARD:GUARD + n].view(dt).view(size)

    def check(self, what):
        torch.cuda.synchronize()
        assert self.regions, "nothing was allocated through the guarded allocator"
        for raw, n in self.regions:
            head, tail = raw[:GUARD], raw[GUARD + n:]
            assert bool((head == PATTERN).all()), f"{what}: bytes BEFORE a {n}-byte buffer were overwritten"
            assert bool((tail == PATTERN).all()), f"{what}: bytes BEHIND a {n}-byte buffer were overwritten"
        count = len(self.regions)
        self.regions = []
        return count


@pytest.fixture()
def guarded(monkeypatch):
    from dpr_scale_amd.hotpath import HipKernels

    g = _Guards()

    class GuardedKernels(HipKernels):
        def _workspace(self, device, nbytes):  # exactly the size the ABI asks for, a fresh guarded buffer per call
            return g.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)

    kn = GuardedKernels()
    monkeypatch.setattr(torch, "empty", g.empty)
    yield kn, g
    monkeypatch.undo()


def _inputs(B, K, d, dev, seed, ragged=True):
    gen = torch.Generator().manual_seed(seed)
    q = (torch.randn(B, d, generator=gen) * d ** -0.25).to(dev).requires_grad_(True)
    c = (torch.randn(B * K, d, generator=gen) * d ** -0.25).to(dev).requires_grad_(True)
    y = (torch.arange(B) * K).to(dev)
    m = torch.zeros(B * K, dtype=torch.bool)
    if ragged and K > 1:
        m[K - 1::K] = torch.rand(B, generator=gen) < 0.4  # dummy contexts at the end of some queries' lists
    return q, c, y, m.to(dev)


# (B, K, d): one shape per plan and per boundary the host side knows
OPERATOR_SHAPES = [
    (32, 8, 768),     # fused small step (cfg2)
    (8, 2, 64),       # tiny
    (33, 3, 100),     # nothing a multiple of anything: padded rows, columns and hidden size
    (64, 2, 1024),    # cfg5 per rank
    (128, 8, 768),    # few-rows plan, 1024 columns
    (128, 64, 768),   # few-rows plan without the dScores launch (8192 columns)
    (96, 24, 512),    # few-rows plan, three row blocks, 2304 columns
    (512, 8, 256),    # 256 x 256 GEMM plans
    (1000, 5, 768),   # ... ragged
    (4096, 2, 768),   # no-logits plan (statistics GEMM, dScores GEMM)
]


@pytest.mark.parametrize("B,K,d", OPERATOR_SHAPES)
def test_operator_forward_backward_stays_inside_its_buffers(B, K, d, guarded):
    from dpr_scale_amd.hotpath import inbatch_contrastive_loss

    kn, g = guarded
    dev = torch.device("cuda", 0)
    q, c, y, m = _inputs(B, K, d, dev, seed=B + K + d)
    for T in (1.0, 0.05):
        loss = inbatch_contrastive_loss(q, c, y, m, T, False, kn)
        loss.backward()
        assert torch.isfinite(loss) and torch.isfinite(q.grad).all() and torch.isfinite(c.grad).all()
    assert g.check(f"operator {B}x{K}x{d}") >= 6


@pytest.mark.parametrize("W,B,K,d", [(8, 128, 8, 768), (8, 32, 8, 768), (2, 64, 2, 1024), (4, 128, 16, 256), (3, 96, 5, 640)])
@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_packed_multi_rank_step_stays_inside_its_buffers(W, B, K, d, wire, guarded):
    kn, g = guarded
    dev = torch.device("cuda", 0)
    n_ctx = B * K
    rows_c = kn._lib.packed_rows(n_ctx, d)
    q, c, y, m = _inputs(B, K, d, dev, seed=W * 1000 + B)
    send = torch.empty((rows_c, d), dtype=torch.bfloat16, device=dev)
    kn.pack_ctx(c.detach(), m.view(torch.uint8), send)
    Cb = torch.empty((W * rows_c, d), dtype=torch.bfloat16, device=dev)
    for r in range(W):
        Cb[r * rows_c:(r + 1) * rows_c].copy_(send)
    Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
    Nq = W * B
    dc_dtype = torch.float32 if wire == "fp32" else torch.bfloat16
    go = torch.ones(1, device=dev)
    for r in (0, W - 1):
        out = kn.inbatch_step_packed_f32(q.detach(), Cb, Qb, W, r, n_ctx, y, 1.0, 1.0 / Nq, want_G="auto")
        assert torch.isfinite(out[4]).all()
        for defer in (False, True):
            try:
                res = kn.train_step_packed_f32(q.detach(), Cb, Qb, W, r, n_ctx, y, 1.0, 1.0 / Nq, 1.0 / Nq, go, dc_dtype=dc_dtype,
                                               defer_dq=defer, want_G="auto")
            except RuntimeError as e:  # a plan without a bf16 dC epilogue says so (the operator then casts): not a memory question
                assert wire == "bf16" and "dc_kind" in str(e), e
                continue
            dQ = res[4]
            if isinstance(dQ, tuple):
                kn.rescale_grads(dQ, res[5], go, go)
                dQ = dQ[0]
            assert torch.isfinite(dQ).all()
    assert g.check(f"packed step W{W} {B}x{K}x{d} {wire}") >= 8


@pytest.mark.parametrize("rows,cols,k", [(3, 70, 16), (9, 4097, 256), (5, 20000, 1000), (2, 70001, 4096), (3, 50000, 5000), (1, 90000, 30000)])
def test_streaming_topk_stays_inside_its_buffers(rows, cols, k, guarded):
    kn, g = guarded
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(rows * cols)
    S = torch.randint(0, 50, (rows, cols), generator=gen).float().to(dev)  # heavy ties
    order = torch.sort(S, dim=1, descending=True, stable=True).indices[:, :k]
    v = torch.empty((rows, k), dtype=torch.float32, device=dev)
    i = torch.empty((rows, k), dtype=torch.int64, device=dev)
    cut = cols // 3
    if k <= 4096:
        v1, i1 = kn.topk(S, k)
        assert torch.equal(i1, order)
        kn.topk_update(S, cut, 0, v, i, True)
        kn.topk_update(S[:, cut:], cols - cut, cut, v, i, False)
    else:
        ws = kn.topk_wide_workspace(rows, k, S)
        kn.topk_update_wide(S, cut, 0, v, i, True, ws)
        kn.topk_update_wide(S[:, cut:], cols - cut, cut, v, i, False, ws)
    assert torch.equal(i, order)
    assert g.check(f"top-k {rows}x{cols} k={k}") >= 2


@pytest.mark.parametrize("nq,n,d,k,chunk", [(7, 3000, 64, 100, 1024), (300, 20000, 128, 1000, 4096), (64, 40000, 256, 5000, 16384)])
def test_retrieval_loop_stays_inside_its_buffers(nq, n, d, k, chunk, guarded):
    from dpr_scale_amd.hotpath import CorpusSearch

    kn, g = guarded
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(nq + n)
    q = torch.randn(nq, d, generator=gen).to(dev)
    c = torch.randn(n, d, generator=gen).to(dev)
    s = CorpusSearch(q, k, chunk=chunk, kernels=kn)
    s.add(c[: n // 2], 0)
    s.add(c[n // 2:], n // 2)
    v, i = s.result()
    assert i.shape == (nq, k) and int(i.min()) >= 0 and int(i.max()) < n
    assert g.check(f"search {nq}x{n}x{d} k={k}") >= 3


@pytest.mark.parametrize("B,Nc,d", [(64, 1000, 768), (1024, 8192, 256), (2048, 16384, 768)])
def test_validation_scoring_stays_inside_its_buffers(B, Nc, d, guarded):
    from dpr_scale_amd.hotpath import rank_and_loss, rank_of_gold, sim_score

    kn, g = guarded
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(B + Nc)
    q = torch.randn(B, d, generator=gen).to(dev)
    c = torch.randn(Nc, d, generator=gen).to(dev)
    y = torch.randint(0, Nc, (B,), generator=gen).to(dev)
    m = (torch.rand(Nc, generator=gen) < 0.1).to(dev)
    m[y] = False
    ranks, loss = rank_and_loss(q, c, y, m, 1.0, kn)
    S = sim_score(q, c, m, 1.0, kn)
    assert torch.equal(rank_of_gold(S, y, kn), ranks)
    assert torch.isfinite(loss)
    assert g.check(f"validation {B}x{Nc}x{d}") >= 3


def test_the_checker_itself_sees_one_stray_byte_on_either_side(guarded):
    _, g = guarded
    dev = torch.device("cuda", 0)
    for where in ("before", "behind"):
        t = torch.empty((8, 3), dtype=torch.float32, device=dev)  # (the patched torch.empty)
        assert t.shape == (8, 3) and t.data_ptr() % 256 == 0
        raw, n = g.regions[-1]
        assert n == 96
        raw[GUARD - 1 if where == "before" else GUARD + n] = 0
        with pytest.raises(AssertionError, match=where.upper()):
            g.check("control")
        g.regions = []


@pytest.mark.parametrize("B,Nc,d", [(64, 1000, 768), (16, 64, 30522), (128, 1032, 30528), (300, 4100, 128)])
def test_differentiable_scoring_stays_inside_its_buffers(B, Nc, d, guarded):
    """sim_score with gradients (citadel_task.py:249-262: loss(sim_score(q, c, mask), y)), vocabulary-wide vectors included (the
    fp32 LDS-DMA sim launch of csrc/wide.h; 30522 is zero-padded to a multiple of 8 first)."""
    from dpr_scale_amd.hotpath import cross_entropy_mean, sim_score

    kn, g = guarded
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(B + Nc + d)
    q = (torch.randn(B, d, generator=gen) * d ** -0.25).to(dev).requires_grad_(True)
    c = (torch.randn(Nc, d, generator=gen) * d ** -0.25).to(dev).requires_grad_(True)
    y = torch.randint(0, Nc, (B,), generator=gen).to(dev)
    m = (torch.rand(Nc, generator=gen) < 0.1).to(dev)
    m[y] = False
    S = sim_score(q, c, m, 1.0, kn)
    loss = cross_entropy_mean(S.detach(), y, kn)
    S.backward(torch.ones_like(S) / S.numel())
    assert torch.isfinite(loss) and torch.isfinite(q.grad).all() and torch.isfinite(c.grad).all()
    assert g.check(f"sim_score {B}x{Nc}x{d}") >= 3


@pytest.mark.parametrize("B,K,d", [(32, 8, 768), (64, 3, 100), (7, 5, 256)])
def test_windowed_and_pairwise_scoring_stay_inside_their_buffers(B, K, d, guarded):
    """in_batch_negatives = False (dpr_task.py:198-207) and the per-query score block the router loss uses."""
    from dpr_scale_amd.hotpath import pairwise_score, windowed_contrastive_loss

    kn, g = guarded
    dev = torch.device("cuda", 0)
    q, c, y, m = _inputs(B, K, d, dev, seed=3 * B + K)
    loss = windowed_contrastive_loss(q, c, y, m, 1.0, kn)
    loss.backward()
    assert torch.isfinite(loss) and torch.isfinite(q.grad).all() and torch.isfinite(c.grad).all()
    if d % 8 == 0:
        P = pairwise_score(q, c, m, kn)
        P.masked_fill(~torch.isfinite(P), 0.0).sum().backward()
        assert P.shape == (B, K)
    assert g.check(f"windowed / pairwise {B}x{K}x{d}") >= 3


@pytest.mark.parametrize("n,W", [(8 * 1000, 2), (110_000_000 // 8 * 8, 8), (8 * 12345, 3)])
@pytest.mark.parametrize("wire", [torch.bfloat16, torch.float16, torch.float32])
def test_gradient_bucket_legs_stay_inside_their_buffers(n, W, wire, guarded):
    """The three local legs of the DDP hook (comm_hooks._HipLegs: pack with pre-division, fp32 sum of the W shards, widen)."""
    from dpr_scale_amd.comm_hooks import _HipLegs

    _, g = guarded
    dev = torch.device("cuda", 0)
    legs = _HipLegs()
    shard = (n + W - 1) // W
    shard = (shard + 7) // 8 * 8
    buf = torch.randn(n, device=dev)
    send = torch.empty(W * shard, dtype=wire, device=dev)
    legs.pack(buf, 1.0 / W, send)
    red = torch.empty(shard, dtype=wire, device=dev)
    legs.sum_shards(send, W, red)  # (this rank's view of an exchange in which every peer sent the same bytes)
    red32 = torch.empty(shard, dtype=torch.float32, device=dev)
    legs.sum_shards(send, W, red32)
    full = torch.empty(W * shard, dtype=wire, device=dev)
    full.copy_(send)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    legs.unpack(full, out)
    tol = 0.0 if wire == torch.float32 else (2 ** -8 if wire == torch.bfloat16 else 2 ** -10)
    assert torch.allclose(out, buf / W, rtol=tol, atol=1e-6)
    assert g.check(f"gradient legs n={n} W={W} {wire}") == 5
