"""The few-rows step WITHOUT its dScores launch (csrc/skinny.h, round 4: sk_sim_kernel leaves every tile's own softmax, sk_bwdf_kernel
derives the row logsumexp and one factor per (row, tile) itself) against (a) an fp64 restatement of dpr_task.py:197-212 and its
autograd backward on the same bf16-representable inputs and (b) the four-launch plan of the same library (G materialised).
Through the C ABI (dprhot_inbatch_step_f32 / _packed_f32 / dprhot_train_step_packed_f32 with G == NULL)."""
import functools

import numpy as np
import pytest
import torch

from oracle import inbatch_oracle as O  # the checker (numpy restatement of dpr_task.py:153-214, pinned to the reference's fixtures)

pytestmark = pytest.mark.gpu

GRAD_BAR = 1e-2   # of max |grad| (tests/test_gpu_parity.py); the fused path is expected well inside it
LOSS_BAR = 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


@pytest.fixture(scope="module", params=[(0, 0), (1, 0), (1, 1)], ids=["two-kinds-4waves", "two-kinds-8waves", "one-kind"])
def kn(request):
    from dpr_scale_amd import _lib
    from dpr_scale_amd.hotpath import HipKernels

    # the plan is chosen by default only where it measured no slower (B x Nc >= 2^19); the tests run it wherever it exists, in
    # every form of the backward launch: dQ units + dC units with four or eight waves (sk_bwdf_kernel, option sk_w8), and the one
    # kind of unit that multiplies a P tile both ways (sk_bwdp_kernel, option sk_pair)
    defaults = {k: _lib.get_option(k) for k in ("sk_fused", "sk_w8", "sk_pair")}
    _lib.set_option("sk_fused", 2)
    _lib.set_option("sk_w8", request.param[0])
    _lib.set_option("sk_pair", request.param[1])
    yield HipKernels()
    for k, v in defaults.items():
        _lib.set_option(k, v)


def _world(W, B, K, d, dev, seed, peaky=False, dup=False, mask_frac=0.05):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    sc = 1.0 if peaky else d ** -0.25
    qs = [(torch.randn(B, d, generator=gen) * sc).to(torch.bfloat16).float().to(dev) for _ in range(W)]
    cs = [(torch.randn(B * K, d, generator=gen) * sc).to(torch.bfloat16).float().to(dev) for _ in range(W)]
    y = torch.arange(B) * K
    if dup:  # several rows share one gold column (the chain of sk_dc_unit_f), one of them at the rank's last context
        y[1] = y[0]
        y[5] = y[0]
        y[B - 1] = B * K - 1
        y[B - 2] = B * K - 1
    ms = []
    for _ in range(W):
        m = torch.rand(B * K, generator=gen) < mask_frac
        m[y] = False
        ms.append(m)
    return qs, cs, y, ms


def _reference(q, C, colmask, yg, T, Nq):
    """fp64: loss numerator, row logsumexp, dQ, dC of this rank's rows (grad_output 1, mean over Nq rows)."""
    C = C.double().clone()
    C[colmask.bool()] = 0.0  # masked columns (and the packed layout's header rows, whose bytes are not numbers) carry no gradient
    S = (q.double() @ C.t()) / T
    S = S.masked_fill(colmask.bool()[None, :], float("-inf"))
    lse = torch.logsumexp(S, dim=1)
    rows = torch.arange(q.shape[0], device=q.device)
    loss_sum = (lse - S[rows, yg]).sum()
    G = torch.exp(S - lse[:, None])
    G[rows, yg] -= 1.0
    G /= (Nq * T)
    return loss_sum, lse, G @ C, G.t() @ q.double()


def _err(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def _packed(kn, qs, cs, ms, dev):
    n_ctx, d = cs[0].shape
    rows_c = kn.packed_rows(n_ctx, d)
    sends = []
    for c, m in zip(cs, ms):
        send = torch.empty((rows_c, d), dtype=torch.bfloat16, device=dev)
        kn.pack_ctx(c, m.to(torch.uint8).to(dev), send)
        sends.append(send)
    Cb = torch.cat(sends, 0).contiguous()
    colmask = torch.empty(len(cs) * rows_c, dtype=torch.uint8, device=dev)
    kn.unpack_mask(Cb, len(cs), n_ctx, colmask)
    return rows_c, Cb, colmask


PACKED_CASES = [
    # W, B, K, d, T, peaky, dup
    (8, 128, 8, 768, 1.0, False, False),    # BASELINE cfg3 per rank: remapped tiles, 64 statistics tiles
    (8, 128, 8, 768, 0.05, True, True),     # near-one-hot rows: factors underflow, gold tiles far below the row maximum; shared golds
    (8, 128, 9, 256, 1.0, False, True),     # 9 tiles per rank: 72 statistics tiles (the NG = 16 instantiation)
    (8, 96, 11, 512, 0.5, False, False),    # n_ctx = 1056: no whole number of tiles per rank -> plain column tiles, ragged last tile
    (8, 64, 16, 512, 1.0, False, True),     # two row blocks
    (4, 128, 16, 128, 1.0, False, False),   # d = 128: one dC column tile, two dQ column tiles
    (8, 32, 32, 768, 1.0, False, True),     # one row block of 32: a single LDS-DMA piece per wave and image in the eight-wave dC units
    (2, 128, 16, 1024, 0.25, True, False),  # d = 1024: sixteen dQ column tiles, the widest finishing row
    (2, 128, 32, 1024, 1.0, False, True),   # ... at 8208 columns: the plan's 8 slices would span 9 statistics tiles, this form cuts its own 10
]


@pytest.mark.parametrize("W,B,K,d,T,peaky,dup", PACKED_CASES)
def test_packed_step_without_dscores_launch(W, B, K, d, T, peaky, dup, kn, dev):
    qs, cs, y, ms = _world(W, B, K, d, dev, seed=W * 100 + K, peaky=peaky, dup=dup)
    n_ctx = B * K
    rows_c, Cb, colmask = _packed(kn, qs, cs, ms, dev)
    Nc = W * rows_c
    assert not kn._lib.step_wants_g(B, Nc, d), "shape is expected to take the fused-dScores plan"
    yd = y.to(dev)
    Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
    inv_T, Nq = 1.0 / T, W * B
    for r in (0, W - 1, W // 2):
        rl, lse, ls, G, dq, dcp = kn.inbatch_step_packed_f32(qs[r], Cb, Qb, W, r, n_ctx, yd, inv_T, inv_T / Nq, want_G=False)
        assert G is None
        rl0, lse0, ls0, G0, dq0, dcp0 = kn.inbatch_step_packed_f32(qs[r], Cb, Qb, W, r, n_ctx, yd, inv_T, inv_T / Nq, want_G=True)
        ref_ls, ref_lse, ref_dq, ref_dc = _reference(qs[r], Cb.float(), colmask, yd + r * rows_c, T, Nq)
        # loss, logsumexp
        assert abs(ls.item() - ref_ls.item()) <= LOSS_BAR * max(1.0, abs(ref_ls.item()))
        assert abs(ls.item() - rl.double().sum().item()) <= 1e-5 * max(1.0, abs(ls.item()))
        assert torch.allclose(lse.double(), ref_lse, rtol=1e-5, atol=1e-4)
        assert torch.allclose(rl, rl0, rtol=1e-4, atol=1e-4)
        # gradients: against fp64, and no worse than the plan that rounds G to bf16
        e_dq, e_dq0 = _err(dq, ref_dq), _err(dq0, ref_dq)
        st = dcp.clone()
        st0 = dcp0.clone()
        for k in range(W):
            assert abs(st[k * rows_c + n_ctx, 0].item() - ls.item()) <= 1e-6 * max(1.0, abs(ls.item()))  # the loss stamp
            st[k * rows_c + n_ctx, 0] = 0.0
            st0[k * rows_c + n_ctx, 0] = 0.0
        e_dc, e_dc0 = _err(st, ref_dc), _err(st0, ref_dc)
        print(f"W{W} B{B} K{K} d{d} T{T} r{r}: dQ err fused {e_dq:.2e} / with-G {e_dq0:.2e}; dC err fused {e_dc:.2e} / with-G {e_dc0:.2e}")
        assert e_dq <= GRAD_BAR and e_dc <= GRAD_BAR
        # (dQ is far better than the plan that rounds G to bf16 -- the gold terms never pass through bf16; dC carries TWO bf16 roundings
        #  per product, the tile softmax and f x q, where that plan has one: up to ~2x its error on near-one-hot rows, e.g. 2.4e-3 against
        #  1.3e-3 at d = 1024, T = 0.25)
        assert e_dq <= max(1.5 * e_dq0, 2e-3) and e_dc <= max(2.0 * e_dc0, 3e-3)
        assert torch.all(st.view(W, rows_c, d)[:, n_ctx:] == 0)  # header rows: exactly zero gradient


@functools.lru_cache(maxsize=4)
def _oracle_world(W, B, K, d, dist, ragged, T, seed):
    """The ORACLE's inputs and global step for a BASELINE-shaped world (computed once per case, shared by the three kernel forms)."""
    parts = [O.synth_embeddings(seed + r, B, K, d, dist, ragged) for r in range(W)]
    n_ctx = B * K
    Q = np.concatenate([p[0] for p in parts])
    C = np.concatenate([p[1] for p in parts])
    y = O.gathered_labels(np.stack([p[2] for p in parts]), n_ctx)
    m = np.concatenate([p[3] for p in parts])
    ref = O.training_step_global(Q, C, y, m, T)
    return parts, {k: ref[k] for k in ("loss", "lse", "dQ", "dC")}


# BASELINE configs[2] (cfg3: W8 B128 K8 d768) in the distributions / temperatures the reference-generated fixtures use, and
# configs[4] at a per-rank batch wide enough for this plan (cfg5's d = 1024 vectors, B 128 K 8)
ORACLE_CASES = [
    (8, 128, 8, 768, "U", True, 1.0, 3100),
    (8, 128, 8, 768, "U", True, 0.05, 3200),
    (8, 128, 8, 768, "P", False, 1.0, 3300),
    (4, 128, 8, 1024, "U", True, 1.0, 3400),
]


@pytest.mark.parametrize("W,B,K,d,dist,ragged,T,seed", ORACLE_CASES)
def test_default_plan_against_the_oracle_at_baseline_shapes(W, B, K, d, dist, ragged, T, seed, kn, dev):
    """Every rank's packed step WITHOUT a dScores launch (G == NULL: what production runs at these shapes) on the oracle's own
    synthetic inputs (O.synth_embeddings) against O.training_step_global -- the restatement of dpr_task.py:163-212
    that tests/test_oracle_golden.py pins to the reference-generated fixtures: loss, every row's logsumexp, every rank's q.grad,
    and c.grad of all W * B * K contexts after the (emulated) reduce-scatter."""
    parts, ref = _oracle_world(W, B, K, d, dist, ragged, T, seed)
    n_ctx, Nq = B * K, W * B
    rows_c = kn.packed_rows(n_ctx, d)
    sends = []
    for r in range(W):
        send = torch.empty((rows_c, d), dtype=torch.bfloat16, device=dev)
        kn.pack_ctx(torch.from_numpy(parts[r][1]).to(dev), torch.from_numpy(parts[r][3].astype(np.uint8)).to(dev), send)
        sends.append(send)
    Cb = torch.cat(sends, 0).contiguous()
    if kn._lib.step_wants_g(B, W * rows_c, d):
        pytest.skip("this shape's plan keeps the dScores launch")
    Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
    inv_T = 1.0 / T
    dC = torch.zeros((W * rows_c, d), dtype=torch.float64, device=dev)
    loss, e_dq = 0.0, 0.0
    for r in range(W):
        q = torch.from_numpy(parts[r][0]).to(dev)
        y = torch.from_numpy(parts[r][2]).to(dev)
        rl, lse, ls, G, dq, dcp = kn.inbatch_step_packed_f32(q, Cb, Qb, W, r, n_ctx, y, inv_T, inv_T / Nq, want_G=False)
        assert G is None
        loss += ls.item()
        dC += dcp.double()
        ref_lse = ref["lse"][r * B:(r + 1) * B]
        assert np.abs(lse.cpu().numpy() - ref_lse).max() <= 1e-3 * max(1.0, np.abs(ref_lse).max())
        ref_dq = ref["dQ"][r * B:(r + 1) * B]
        e = np.abs(dq.cpu().numpy() - ref_dq).max() / np.abs(ref["dQ"]).max()
        e_dq = max(e_dq, e)
    assert abs(loss / Nq - ref["loss"]) <= LOSS_BAR * max(1.0, abs(ref["loss"]))
    got = dC.cpu().numpy().reshape(W, rows_c, d)
    for k in range(W):  # the piggy-backed loss numerator at [n_ctx][0] of every chunk, then nothing but zeros in the header rows
        assert abs(got[k, n_ctx, 0] - loss) <= 1e-5 * max(1.0, abs(loss))  # (summed over the ranks: the global numerator)
        got[k, n_ctx, 0] = 0.0
    assert np.all(got[:, n_ctx:] == 0.0)
    e_dc = np.abs(got[:, :n_ctx].reshape(W * n_ctx, d) - ref["dC"]).max() / np.abs(ref["dC"]).max()
    print(f"[plan-error] oracle W{W} B{B} K{K} d{d} {dist} T{T}: dQ {e_dq:.2e}  dC {e_dc:.2e} (of max |grad|, default plan)")
    assert e_dq <= GRAD_BAR and e_dc <= GRAD_BAR


@pytest.mark.parametrize("B,Nc,d,T", [(128, 8192, 768, 1.0), (128, 4096, 768, 0.05), (96, 8200, 256, 1.0), (128, 16384, 128, 1.0), (128, 12288, 768, 1.0)])
def test_single_rank_step_without_dscores_launch(B, Nc, d, T, kn, dev):
    """dprhot_inbatch_step_f32 with G == NULL: explicit mask vector, plain column tiles (Nc = 8200: a ragged last tile of 8 columns;
    16384: the plan's widest shape -- 128 statistics tiles)."""
    gen = torch.Generator(device="cpu").manual_seed(Nc + B)
    q = (torch.randn(B, d, generator=gen) * d ** -0.25).to(torch.bfloat16).float().to(dev)
    c = (torch.randn(Nc, d, generator=gen) * d ** -0.25).to(torch.bfloat16).float().to(dev)
    y = (torch.randperm(Nc, generator=gen)[:B]).sort().values
    y[3] = y[2]
    m = torch.rand(Nc, generator=gen) < 0.05
    m[y] = False
    m8 = m.to(torch.uint8).to(dev)
    yd = y.to(dev)
    if kn._lib.step_wants_g(B, Nc, d):
        pytest.skip("this shape's plan keeps the dScores launch")
    Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
    Cb = torch.empty((Nc, d), dtype=torch.bfloat16, device=dev)
    inv_T = 1.0 / T
    rl, lse, ls, G, dq, dc = kn.inbatch_step_f32(q, c, Qb, Cb, yd, 0, m8, inv_T, inv_T / B, want_G=False)
    assert G is None
    _, _, ls0, _, dq0, dc0 = kn.inbatch_step_f32(q, c, Qb, Cb, yd, 0, m8, inv_T, inv_T / B, want_G=True)
    ref_ls, ref_lse, ref_dq, ref_dc = _reference(q, c, m8, yd, T, B)
    assert abs(ls.item() - ref_ls.item()) <= LOSS_BAR * max(1.0, abs(ref_ls.item()))
    assert torch.allclose(lse.double(), ref_lse, rtol=1e-5, atol=1e-4)
    e_dq, e_dq0, e_dc, e_dc0 = _err(dq, ref_dq), _err(dq0, ref_dq), _err(dc, ref_dc), _err(dc0, ref_dc)
    print(f"B{B} Nc{Nc} d{d} T{T}: dQ err fused {e_dq:.2e} / with-G {e_dq0:.2e}; dC err fused {e_dc:.2e} / with-G {e_dc0:.2e}")
    assert e_dq <= GRAD_BAR and e_dc <= GRAD_BAR
    assert e_dq <= max(1.5 * e_dq0, 2e-3) and e_dc <= max(1.5 * e_dc0, 2e-3)


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_train_step_without_dscores_launch_both_wires_and_deferred_dq(wire, kn, dev):
    """What the operator runs under DDP (dprhot_train_step_packed_f32, G == NULL): loss as the mean, a device grad_output, dC as fp32
    or as the bf16 wire format, dQ left as slabs for dprhot_rescale_grads."""
    W, B, K, d, T = 8, 128, 8, 768, 1.0
    qs, cs, y, ms = _world(W, B, K, d, dev, seed=77)
    n_ctx = B * K
    rows_c, Cb, colmask = _packed(kn, qs, cs, ms, dev)
    yd = y.to(dev)
    Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
    Nq, go = W * B, 8.0
    dt = torch.float32 if wire == "fp32" else torch.bfloat16
    d_scale = torch.full((1,), go, device=dev)
    r = 3
    rl, lse, lo, G, dq, dcp = kn.train_step_packed_f32(qs[r], Cb, Qb, W, r, n_ctx, yd, 1.0, 1.0 / Nq, 1.0 / Nq, d_scale, dt, want_G=False)
    assert G is None and dcp.dtype == dt
    ref_ls, ref_lse, ref_dq, ref_dc = _reference(qs[r], Cb.float(), colmask, yd + r * rows_c, T, Nq)
    assert abs(lo[0].item() * Nq - ref_ls.item()) <= LOSS_BAR * max(1.0, abs(ref_ls.item()))
    assert _err(dq / go, ref_dq) <= GRAD_BAR
    st = dcp.float() / go
    if wire == "fp32":
        for k in range(W):
            st[k * rows_c + n_ctx, 0] = 0.0
    assert _err(st, ref_dc) <= GRAD_BAR
    assert torch.all(st.view(W, rows_c, d)[:, n_ctx:] == 0)
    _, _, _, _, dq_def, dcp2 = kn.train_step_packed_f32(qs[r], Cb, Qb, W, r, n_ctx, yd, 1.0, 1.0 / Nq, 1.0 / Nq, d_scale, dt, defer_dq=True,
                                                        want_G=False)
    # (the step finishes dQ itself at these shapes -- its slabs are slice-normalised: dprhot_train_dq_slabs = 0, nothing is deferred)
    assert kn._lib.train_dq_slabs(B, W * rows_c, d) == 0 and not isinstance(dq_def, tuple)
    go2 = torch.full((1,), 2.0, device=dev)
    out2 = kn.rescale_grads(dq_def, dcp2, go2, d_scale)
    assert out2.tolist() == [2.0, 2.0]
    assert _err(dq_def / 2.0, dq / go) <= 1e-5


def test_operator_takes_the_no_dscores_plan_and_survives_a_retained_graph(dev):
    """InBatchContrastive at a fused-plan shape: same loss / gradients as an fp64 reference; a second backward through a retained
    graph recomputes the dScores it never stored."""
    from dpr_scale_amd.hotpath import inbatch_contrastive_loss

    B, K, d = 128, 64, 768
    gen = torch.Generator(device="cpu").manual_seed(5)
    q = (torch.randn(B, d, generator=gen) * d ** -0.25).to(torch.bfloat16).float().to(dev).requires_grad_(True)
    c = (torch.randn(B * K, d, generator=gen) * d ** -0.25).to(torch.bfloat16).float().to(dev).requires_grad_(True)
    y = (torch.arange(B) * K).to(dev)
    m = torch.zeros(B * K, dtype=torch.bool, device=dev)
    loss = inbatch_contrastive_loss(q, c, y, m, 1.0, group=False)
    loss.backward(retain_graph=True)
    dq1, dc1 = q.grad.clone(), c.grad.clone()
    ref_ls, _, ref_dq, ref_dc = _reference(q.detach(), c.detach(), m.to(torch.uint8), y, 1.0, B)
    assert abs(loss.item() - ref_ls.item() / B) <= LOSS_BAR
    assert _err(dq1, ref_dq) <= GRAD_BAR and _err(dc1, ref_dc) <= GRAD_BAR
    q.grad = None
    c.grad = None
    loss.backward()
    assert _err(q.grad, dq1) <= 5e-3 and _err(c.grad, dc1) <= 5e-3


def test_dc_units_scaling_fragments_in_registers_is_bit_identical(dev):
    """Option sk_dc_regscale (round 5; measured no faster, off by default): f x q formed per fragment element in registers instead of by
    a pass over the LDS image -- the same products, the same rounding: bit-identical gradients."""
    from dpr_scale_amd import _lib
    from dpr_scale_amd.hotpath import HipKernels

    kn = HipKernels()
    for W, B, K, d, T in ((8, 128, 8, 768, 1.0), (8, 64, 16, 512, 0.25), (2, 128, 16, 1024, 0.25)):
        qs, cs, y, ms = _world(W, B, K, d, dev, seed=W * 11 + K, dup=True)
        n_ctx = B * K
        rows_c, Cb, colmask = _packed(kn, qs, cs, ms, dev)
        if _lib.step_wants_g(B, W * rows_c, d):
            continue
        yd = y.to(dev)
        Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
        outs = []
        for rs in (0, 1):
            _lib.set_option("sk_dc_regscale", rs)
            try:
                rl, lse, ls, G, dq, dcp = kn.inbatch_step_packed_f32(qs[0], Cb, Qb, W, 0, n_ctx, yd, 1.0 / T, 1.0 / (T * W * B), want_G=False)
                outs.append((dq.clone(), dcp.clone()))
            finally:
                _lib.set_option("sk_dc_regscale", 0)
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("mode", [1, 2])
def test_finishing_role_inside_the_backward_launch(mode, dev):
    """Option sk_tail (round 5; measured slower, off by default, kept with this test): the dQ slabs folded by the last workgroups of
    sk_bwdf_kernel behind a count of the dQ units -- mode 1 write-through slab stores + sc1 loads, mode 2 ordinary stores + release /
    acquire fences -- must give exactly what the finishing launch gives (same slabs, same order of additions)."""
    from dpr_scale_amd import _lib
    from dpr_scale_amd.hotpath import HipKernels

    kn = HipKernels()
    for W, B, K, d, T in ((8, 128, 8, 768, 1.0), (8, 96, 11, 512, 0.5), (2, 128, 16, 1024, 0.25)):
        qs, cs, y, ms = _world(W, B, K, d, dev, seed=W * 7 + K)
        n_ctx = B * K
        rows_c, Cb, colmask = _packed(kn, qs, cs, ms, dev)
        if _lib.step_wants_g(B, W * rows_c, d):
            continue
        yd = y.to(dev)
        Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
        outs = []
        for tail in (0, mode, mode, 0):
            _lib.set_option("sk_tail", tail)
            try:
                rl, lse, ls, G, dq, dcp = kn.inbatch_step_packed_f32(qs[1], Cb, Qb, W, 1, n_ctx, yd, 1.0 / T, 1.0 / (T * W * B), want_G=False)
                outs.append((dq.clone(), dcp.clone(), ls.item()))
            finally:
                _lib.set_option("sk_tail", 0)
        for o in outs[1:]:
            assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) and o[2] == outs[0][2]
        assert torch.isfinite(outs[1][0]).all()


@pytest.mark.parametrize("w8", [1, 0])
@pytest.mark.parametrize("W,B,K,d,T,peaky,dup", PACKED_CASES)
def test_dq_units_adding_into_dq_without_a_finishing_launch(W, B, K, d, T, peaky, dup, w8, dev):
    """Option sk_dq_atomic (round 6): the dQ units derive the rows' logsumexp at their END, scale their tile to the row softmax, add the
    gold term of the rows whose gold column lies in their slice and ADD the tile into dQ (global_atomic_add_f32; dQ zero-filled by the
    sim launch) -- no slabs, no sk_dq_finish_kernel.  Against fp64 (the bar of the slab form) and against the slab form itself: the same
    terms in another order of addition, so equal to fp32 rounding (<= 1e-5 of max |dQ|), dC / loss / logsumexp BIT-identical (untouched);
    a second run may differ in the last bit (run-dependent order), never beyond rounding; poisoned dQ buffers (NaN) must not leak."""
    from dpr_scale_amd import _lib
    from dpr_scale_amd.hotpath import HipKernels

    kn = HipKernels()
    defaults = {k: _lib.get_option(k) for k in ("sk_fused", "sk_w8", "sk_pair", "sk_dq_atomic")}
    try:
        _lib.set_option("sk_fused", 2)
        _lib.set_option("sk_w8", w8)
        _lib.set_option("sk_pair", 0)
        qs, cs, y, ms = _world(W, B, K, d, dev, seed=W * 100 + K, peaky=peaky, dup=dup)
        n_ctx = B * K
        rows_c, Cb, colmask = _packed(kn, qs, cs, ms, dev)
        Nc = W * rows_c
        if _lib.step_wants_g(B, Nc, d):
            pytest.skip("this shape's plan keeps the dScores launch")
        yd = y.to(dev)
        Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
        inv_T, Nq = 1.0 / T, W * B
        for r in (0, W - 1):
            _lib.set_option("sk_dq_atomic", 0)
            rl0, lse0, ls0, _, dq0, dcp0 = kn.inbatch_step_packed_f32(qs[r], Cb, Qb, W, r, n_ctx, yd, inv_T, inv_T / Nq, want_G=False)
            dq0, dcp0 = dq0.clone(), dcp0.clone()
            _lib.set_option("sk_dq_atomic", 1)
            outs = []
            for rep in range(3):
                rl, lse, ls, G, dq, dcp = kn.inbatch_step_packed_f32(qs[r], Cb, Qb, W, r, n_ctx, yd, inv_T, inv_T / Nq, want_G=False)
                assert G is None
                outs.append(dq.clone())
                assert torch.equal(dcp, dcp0) and torch.equal(lse, lse0) and ls.item() == ls0.item()
            ref_ls, ref_lse, ref_dq, ref_dc = _reference(qs[r], Cb.float(), colmask, yd + r * rows_c, T, Nq)
            e_dq, e_dq0 = _err(outs[0], ref_dq), _err(dq0, ref_dq)
            print(f"W{W} B{B} K{K} d{d} T{T} r{r} w8={w8}: dQ err atomic {e_dq:.2e} / slabs {e_dq0:.2e}; atomic vs slabs {_err(outs[0], dq0):.2e}")
            assert torch.isfinite(outs[0]).all() and e_dq <= GRAD_BAR and e_dq <= max(1.5 * e_dq0, 1e-4)
            assert _err(outs[0], dq0) <= 1e-5
            for o in outs[1:]:
                assert _err(o, outs[0]) <= 2e-6
    finally:
        for k, v in defaults.items():
            _lib.set_option(k, v)
