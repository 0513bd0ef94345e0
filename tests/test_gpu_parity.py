"""Parity tests proper: the HIP path (through the C ABI) against the reference-generated fixtures, the numpy
oracle, and size-independent properties at BASELINE.json's full sizes.  All need a real MI355X.

Tolerances (north_star): loss / logits <= 1e-3 relative to the fp32 reference on identical bf16-representable
inputs (measured ~1e-6: only the accumulation order differs); rank / top-k indices bit-exact under the frozen
tie rule; gradients <= 1e-2 of max|grad| because dScores travels as bf16 (2^-9 per element), as it does in the
reference under AMP.
"""
import math

import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from oracle import inbatch_oracle as O

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-3
LOGIT_RTOL = 1e-3
GRAD_RTOL = 1e-2


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def kn():
    from dpr_scale_amd.hotpath import default_kernels

    return default_kernels()


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def rank_inputs(meta):
    W, B, K, d = meta["W"], meta["B"], meta["K"], meta["d"]
    return [O.synth_embeddings(meta["seed"] + r, B, K, d, meta["dist"], meta["ragged"]) for r in range(W)]


def t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def bf16(x, dev):
    return t(x, dev).to(torch.bfloat16)


def test_native_library_is_the_in_tree_one(kn):
    import os

    from dpr_scale_amd import _lib

    assert os.path.samefile(os.path.dirname(_lib.LIB_PATH), os.path.dirname(_lib.__file__))
    maps = open("/proc/self/maps").read()
    assert "libdprhot.so" in maps


@pytest.mark.parametrize("name", golden_names("cfg1") + golden_names("cfg2"))
def test_autograd_op_against_reference_fixture(name, dev):
    """The whole operator (cast -> sim -> softmax CE -> backward) vs the reference's training_step outputs."""
    from dpr_scale_amd.hotpath import inbatch_contrastive_loss, rank_of_gold, sim_score

    meta, g = load_golden(name)
    q, c, y, m = rank_inputs(meta)[0]
    tq = t(q, dev).requires_grad_(True)
    tc = t(c, dev).requires_grad_(True)
    loss = inbatch_contrastive_loss(tq, tc, t(y, dev), t(m, dev), meta["T"])
    loss.backward()
    assert abs(loss.item() - g["loss"]) <= LOSS_RTOL * max(1.0, abs(g["loss"]))
    assert rel(tq.grad.cpu().numpy(), g["dQ"]) <= GRAD_RTOL
    assert rel(tc.grad.cpu().numpy(), g["dC"]) <= GRAD_RTOL
    S = sim_score(tq.detach(), tc.detach(), t(m, dev)).cpu().numpy()
    fin = np.isfinite(g["S"])
    assert np.array_equal(fin, np.isfinite(S))
    assert np.all(S[~fin] == -np.inf)
    assert rel(S[fin], g["S"][fin]) <= LOGIT_RTOL
    ranks = rank_of_gold(t(S, dev), t(y, dev)).cpu().numpy()
    assert np.array_equal(ranks, g["ranks"])


def test_grad_output_scalar_is_applied_on_device(dev):
    from dpr_scale_amd.hotpath import inbatch_contrastive_loss

    meta, g = load_golden("cfg2_U_T1")
    q, c, y, m = rank_inputs(meta)[0]
    tq = t(q, dev).requires_grad_(True)
    tc = t(c, dev).requires_grad_(True)
    loss = inbatch_contrastive_loss(tq, tc, t(y, dev), t(m, dev), meta["T"])
    (loss * 1024.0).backward()  # AMP-style loss scale
    assert rel(tq.grad.cpu().numpy() / 1024.0, g["dQ"]) <= GRAD_RTOL
    assert rel(tc.grad.cpu().numpy() / 1024.0, g["dC"]) <= GRAD_RTOL


def _emulate_ranks(meta, kn, dev, want_logits_rank=None):
    """Run every rank's local-rows step on this one GPU; return per-rank pieces and the summed dC."""
    W, B, K, d = meta["W"], meta["B"], meta["K"], meta["d"]
    parts = rank_inputs(meta)
    Cb = bf16(np.concatenate([p[1] for p in parts]), dev)
    mask = t(np.concatenate([p[3] for p in parts]).astype(np.uint8), dev)
    inv_T = 1.0 / meta["T"]
    out = []
    dC = torch.zeros((W * B * K, d), dtype=torch.float32, device=dev)
    one = torch.ones(1, dtype=torch.float32, device=dev)
    for r in range(W):
        Qb = bf16(parts[r][0], dev)
        y = t(parts[r][2], dev)
        row_loss, lse, loss_sum, G, S = kn.inbatch_fwd(Qb, Cb, y, r * B * K, mask, inv_T, inv_T / (W * B),
                                                       want_logits=(r == want_logits_rank))
        dq, dcp = kn.inbatch_bwd(G, Qb, Cb, 1.0, one)
        dC += dcp
        out.append(dict(loss_sum=loss_sum.item(), lse=lse.cpu().numpy(), dq=dq.cpu().numpy(),
                        S=None if S is None else S.cpu().numpy(), y=parts[r][2] + r * B * K))
    return out, dC.cpu().numpy()


@pytest.mark.parametrize("name", ["w2_ddp", "w4_ddp", "cfg4_ddp"])
def test_local_rows_vs_reference_ddp_branch(name, kn, dev):
    """Per-rank loss / q.grad / c.grad of the reference's DDP branch (real gloo ranks) from the HIP kernels."""
    meta, g = load_golden(name)
    W, B, K = meta["W"], meta["B"], meta["K"]
    outs, dC = _emulate_ranks(meta, kn, dev)
    loss = sum(o["loss_sum"] for o in outs) / (W * B)
    for r in range(W):
        assert abs(loss - g["loss_per_rank"][r]) <= LOSS_RTOL * max(1.0, abs(loss))
        assert rel(outs[r]["dq"], g["dq_per_rank"][r]) <= GRAD_RTOL
        assert rel(dC[r * B * K:(r + 1) * B * K], g["dc_per_rank"][r]) <= GRAD_RTOL


@pytest.mark.parametrize("name", ["w2_ddp", "w4_ddp", "cfg4_ddp"])
def test_packed_single_gather_layout_vs_reference_ddp(name, kn, dev):
    """The multi-rank wire format (context rows + mask bytes in ONE buffer per rank, dprhot_pack_ctx /
    dprhot_unpack_mask) emulated on one GPU: every rank's send buffer is built by the kernel, the buffers are
    concatenated as an all-gather would, and each rank's step must give the reference DDP-branch results."""
    meta, g = load_golden(name)
    W, B, K, d = meta["W"], meta["B"], meta["K"], meta["d"]
    n_ctx = B * K
    parts = rank_inputs(meta)
    rows_c = kn.packed_rows(n_ctx, d)
    assert rows_c % 8 == 0 and rows_c >= n_ctx + 1
    sends = []
    for r in range(W):
        send = torch.empty((rows_c, d), dtype=torch.bfloat16, device=dev)
        kn.pack_ctx(t(parts[r][1], dev), t(parts[r][3].astype(np.uint8), dev), send)
        sends.append(send)
    Cb = torch.cat(sends, 0).contiguous()  # == all_gather_into_tensor
    colmask = torch.empty(W * rows_c, dtype=torch.uint8, device=dev)
    kn.unpack_mask(Cb, W, n_ctx, colmask)
    cm = colmask.cpu().numpy().reshape(W, rows_c)
    for r in range(W):
        assert np.array_equal(cm[r, :n_ctx], parts[r][3].astype(np.uint8)) and np.all(cm[r, n_ctx:] == 1)
        assert np.array_equal(Cb[r * rows_c:r * rows_c + n_ctx].float().cpu().numpy(), parts[r][1])
    inv_T = 1.0 / meta["T"]
    one = torch.ones(1, dtype=torch.float32, device=dev)
    dC = torch.zeros((W * rows_c, d), dtype=torch.float32, device=dev)
    Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
    loss_sum, dqs = 0.0, []
    for r in range(W):
        _, _, ls, G, _ = kn.inbatch_fwd_f32(t(parts[r][0], dev), None, Qb, Cb, t(parts[r][2], dev), r * rows_c, colmask,
                                            inv_T, inv_T / (W * B))
        dq, dcp = kn.inbatch_bwd(G, Qb, Cb, 1.0, one)
        dC += dcp
        loss_sum += ls.item()
        dqs.append(dq.cpu().numpy())
    dC = dC.cpu().numpy()
    for r in range(W):
        assert abs(loss_sum / (W * B) - g["loss_per_rank"][r]) <= LOSS_RTOL * max(1.0, abs(g["loss_per_rank"][r]))
        assert rel(dqs[r], g["dq_per_rank"][r]) <= GRAD_RTOL
        assert rel(dC[r * rows_c:r * rows_c + n_ctx], g["dc_per_rank"][r]) <= GRAD_RTOL


@pytest.mark.parametrize("name", ["w2_ddp", "w4_ddp", "cfg4_ddp"])
def test_packed_step_mask_from_buffer_and_loss_in_reduce_scatter(name, kn, dev):
    """dprhot_inbatch_step_packed_f32 for every rank on one GPU: no unpack launch (the sim kernel reads the mask bytes from
    the gathered buffer), and the reduce-scatter of dC_part -- emulated by summing the ranks' buffers -- must deliver
    the reference's c.grad in the first n_ctx rows of every rank chunk AND the global loss numerator at [n_ctx][0]."""
    meta, g = load_golden(name)
    W, B, K, d = meta["W"], meta["B"], meta["K"], meta["d"]
    n_ctx = B * K
    parts = rank_inputs(meta)
    rows_c = kn.packed_rows(n_ctx, d)
    sends = []
    for r in range(W):
        send = torch.empty((rows_c, d), dtype=torch.bfloat16, device=dev)
        kn.pack_ctx(t(parts[r][1], dev), t(parts[r][3].astype(np.uint8), dev), send)
        sends.append(send)
    Cb = torch.cat(sends, 0).contiguous()  # == all_gather_into_tensor
    inv_T = 1.0 / meta["T"]
    Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
    dC = torch.zeros((W * rows_c, d), dtype=torch.float64, device=dev)
    local, dqs = [], []
    for r in range(W):
        _, _, ls, _, dq, dcp = kn.inbatch_step_packed_f32(t(parts[r][0], dev), Cb, Qb, W, r, n_ctx, t(parts[r][2], dev), inv_T,
                                                           inv_T / (W * B))
        dC += dcp.double()  # what the reduce-scatter sums
        local.append(ls.item())
        dqs.append(dq.cpu().numpy())
    dC = dC.cpu().numpy().reshape(W, rows_c, d)
    for r in range(W):
        chunk = dC[r]
        assert rel(dqs[r], g["dq_per_rank"][r]) <= GRAD_RTOL
        assert rel(chunk[:n_ctx], g["dc_per_rank"][r]) <= GRAD_RTOL
        loss = chunk[n_ctx, 0] / (W * B)  # the piggy-backed sum of the loss numerators
        assert abs(chunk[n_ctx, 0] - sum(local)) <= 1e-6 * max(1.0, abs(sum(local)))
        assert abs(loss - g["loss_per_rank"][r]) <= LOSS_RTOL * max(1.0, abs(g["loss_per_rank"][r]))
        dead = chunk[n_ctx:].copy()
        dead[0, 0] = 0.0
        assert np.all(dead == 0.0)  # the other mask-row gradients are exactly zero (masked columns have G = 0)


@pytest.mark.parametrize("name", golden_names("cfg3") + golden_names("cfg5"))
def test_full_size_configs_against_reference_summaries(name, kn, dev):
    """cfg3 (W8 B128 K8 d768: 1024 x 8192) and cfg5 (W8 B64 K2 d1024) at full size."""
    meta, g = load_golden(name)
    W, B, K = meta["W"], meta["B"], meta["K"]
    own = meta["own_rank"]
    outs, dC = _emulate_ranks(meta, kn, dev, want_logits_rank=own)
    loss = sum(o["loss_sum"] for o in outs) / (W * B)
    assert abs(loss - g["loss"]) <= LOSS_RTOL * max(1.0, abs(g["loss"]))
    assert rel(np.concatenate([o["lse"] for o in outs]), g["lse"]) <= LOGIT_RTOL
    assert rel(outs[own]["dq"], g["dq_own"]) <= GRAD_RTOL
    S = outs[own]["S"] * meta["T"]
    si, sj = g["sample_i"], g["sample_j"]
    sel = (si >= own * B) & (si < (own + 1) * B)
    mine, ref = S[si[sel] - own * B, sj[sel]], g["sample_S"][sel]
    fin = np.isfinite(ref)
    assert np.array_equal(fin, np.isfinite(mine)) and rel(mine[fin], ref[fin]) <= LOGIT_RTOL
    from dpr_scale_amd.hotpath import rank_of_gold

    ranks = rank_of_gold(t(outs[own]["S"], dev), t(outs[own]["y"], dev)).cpu().numpy()
    assert np.array_equal(ranks, g["ranks"][own * B:(own + 1) * B])
    assert rel(dC[own * B * K:own * B * K + 64], g["dc_own_head"]) <= GRAD_RTOL


def _packed_world(meta, kn, dev):
    """Every rank's all-gather send buffer built by dprhot_pack_ctx, concatenated as the all-gather would."""
    W, B, K, d = meta["W"], meta["B"], meta["K"], meta["d"]
    n_ctx = B * K
    parts = rank_inputs(meta)
    rows_c = kn.packed_rows(n_ctx, d)
    sends = []
    for r in range(W):
        send = torch.empty((rows_c, d), dtype=torch.bfloat16, device=dev)
        kn.pack_ctx(t(parts[r][1], dev), t(parts[r][3].astype(np.uint8), dev), send)
        sends.append(send)
    return parts, rows_c, torch.cat(sends, 0).contiguous()


@pytest.mark.parametrize("plan", ["with-dscores", "default-plan"])
@pytest.mark.parametrize("name", golden_names("cfg3") + golden_names("cfg5"))
def test_production_packed_step_every_rank_full_size(name, plan, kn, dev):
    """plan: "with-dscores" passes a G buffer (the plan that materialises the dScores, four launches at cfg3 per rank);
    "default-plan" passes G == NULL wherever dprhot_step_wants_g says the shape's plan needs none -- what the autograd operator
    (want_G = "auto") and bench.py run since round 4: three launches at cfg3 per rank, sk_sim -> sk_bwdf -> sk_dq_finish.
    The entry point DDP training actually runs -- dprhot_inbatch_step_packed_f32: fp32 q straight into the sim kernel
    (dprhot_sim_stats_f32<AF, !BF>, long-row statistics plan at B = 128 / packed Nc = 8 * 1032 = 8256 for cfg3; the
    short-row plan with d = 1024 for cfg5), mask bytes read from the gathered buffer, loss numerator riding in dC_part --
    for EVERY rank of cfg3 (W8 B128 K8 d768) and cfg5 (W8 B64 K2 d1024), against the reference's own global step
    (dpr_task.py:163-212 run through oracle/ref_shim.py): loss, row logsumexp of all 1024 / 512 rows, q.grad of the
    fixture's rank, c.grad head / row sums / column sums of its chunk after the (emulated) reduce-scatter."""
    meta, g = load_golden(name)
    W, B, K, d, T = meta["W"], meta["B"], meta["K"], meta["d"], meta["T"]
    own, n_ctx = meta["own_rank"], B * K
    parts, rows_c, Cb = _packed_world(meta, kn, dev)
    want_G = plan == "with-dscores"
    if not want_G and kn._wants_g(B, W * rows_c, d):
        pytest.skip("this shape's default plan materialises the dScores: covered by the with-dscores leg")
    inv_T = 1.0 / T
    Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
    dC = torch.zeros((W * rows_c, d), dtype=torch.float64, device=dev)
    local, lses, dq_own, rl_all = [], [], None, []
    for r in range(W):
        rl, lse, ls, G, dq, dcp = kn.inbatch_step_packed_f32(t(parts[r][0], dev), Cb, Qb, W, r, n_ctx, t(parts[r][2], dev),
                                                              inv_T, inv_T / (W * B), want_G=want_G)
        assert np.array_equal(Qb.float().cpu().numpy(), parts[r][0])  # bf16 copy-out of the fp32-operand sim kernel
        dC += dcp.double()  # what the reduce-scatter sums
        local.append(ls.item())
        lses.append(lse.cpu().numpy())
        rl_all.append(rl.cpu().numpy())
        assert abs(ls.item() - rl.double().sum().item()) <= 1e-5 * max(1.0, abs(ls.item()))
        assert (G is not None) == want_G
        if want_G:
            assert G.float().sum(dim=1).abs().max().item() <= 2e-2 * inv_T / (W * B)  # softmax - onehot: rows sum to zero
        if r == own:
            dq_own = dq.cpu().numpy()
    loss = sum(local) / (W * B)
    assert abs(loss - g["loss"]) <= LOSS_RTOL * max(1.0, abs(g["loss"]))
    assert rel(np.concatenate(lses), g["lse"]) <= LOGIT_RTOL
    assert rel(dq_own, g["dq_own"]) <= GRAD_RTOL
    chunk = dC.cpu().numpy().reshape(W, rows_c, d)[own]
    print(f"[plan-error] {name} {plan}: loss rel {abs(loss - g['loss']) / max(1.0, abs(g['loss'])):.2e}  dq_own {rel(dq_own, g['dq_own']):.2e}  "
          f"dc_own_head {rel(chunk[:64], g['dc_own_head']):.2e}  dc_own_rowsum {rel(chunk[:n_ctx].sum(1), g['dc_own_rowsum']):.2e}  "
          f"dc_own_colsum {rel(chunk[:n_ctx].sum(0), g['dc_own_colsum']):.2e}")
    assert rel(chunk[:64], g["dc_own_head"]) <= GRAD_RTOL
    assert rel(chunk[:n_ctx].sum(1), g["dc_own_rowsum"]) <= GRAD_RTOL
    assert rel(chunk[:n_ctx].sum(0), g["dc_own_colsum"]) <= GRAD_RTOL
    # the global column sum is ~0 (rows of softmax - onehot sum to zero): absolute bar at the scale of one rank's chunk
    tot = dC.cpu().numpy().reshape(W, rows_c, d)[:, :n_ctx].sum((0, 1))
    assert np.abs(tot - g["dC_colsum"]).max() <= GRAD_RTOL * W * np.abs(g["dc_own_colsum"]).max()
    assert abs(chunk[n_ctx, 0] - sum(local)) <= 1e-6 * max(1.0, abs(sum(local)))  # piggy-backed loss numerator
    dead = chunk[n_ctx:].copy()
    dead[0, 0] = 0.0
    assert np.all(dead == 0.0)
    # logits and ranks of the fixture's rank from the same fp32-operand sim instance (mask through dprhot_unpack_mask)
    colmask = torch.empty(W * rows_c, dtype=torch.uint8, device=dev)
    kn.unpack_mask(Cb, W, n_ctx, colmask)
    _, lse2, _, _, S = kn.inbatch_fwd_f32(t(parts[own][0], dev), None, Qb, Cb, t(parts[own][2], dev), own * rows_c, colmask,
                                         inv_T, inv_T / (W * B), want_logits=True)
    assert rel(lse2.cpu().numpy(), lses[own]) <= 1e-6
    S = S.cpu().numpy().reshape(B, W, rows_c)[:, :, :n_ctx].reshape(B, W * n_ctx) * T  # drop the mask rows' columns
    si, sj = g["sample_i"], g["sample_j"]
    sel = (si >= own * B) & (si < (own + 1) * B)
    mine, ref = S[si[sel] - own * B, sj[sel]], g["sample_S"][sel]
    fin = np.isfinite(ref)
    assert np.array_equal(fin, np.isfinite(mine)) and rel(mine[fin], ref[fin]) <= LOGIT_RTOL
    from dpr_scale_amd.hotpath import rank_of_gold

    ranks = rank_of_gold(t(np.ascontiguousarray(S), dev), t(parts[own][2] + own * n_ctx, dev)).cpu().numpy()
    assert np.array_equal(ranks, g["ranks"][own * B:(own + 1) * B])


@pytest.mark.parametrize("B,K,d,T,ragged", [(64, 8, 768, 1.0, True), (48, 4, 256, 0.5, True), (40, 16, 128, 0.05, False), (64, 2, 1024, 1.0, False),
                                            (33, 8, 64, 1.0, True), (64, 12, 768, 1.0, True)])
def test_two_row_blocks_in_the_fused_softmax_backward_kernel(B, K, d, T, ragged, kn, dev):
    """32 < B <= 64 (the per-GPU batch of the DRAGON / NQ recipes) with up to 256 contexts: the step is still TWO launches -- the
    fused softmax + backward kernel takes two turns of 32 rows with the C tile resident and dC accumulated in registers
    (step_small.h, NRB = 2); wider shapes of the list take the three-launch plan.  The whole operator against the fp64 oracle,
    AMP-style grad_output included, and the fused kernel against the three-launch plan."""
    from dpr_scale_amd import _lib, hotpath

    q, c, y, m = O.synth_embeddings(5000 + B + K, B, K, d, "U", ragged)
    ref = O.training_step_global(q, c, y, m, T)
    tq, tc = t(q, dev).requires_grad_(True), t(c, dev).requires_grad_(True)
    loss = hotpath.inbatch_contrastive_loss(tq, tc, t(y, dev), t(m, dev), T)
    (loss * 4.0).backward()
    assert abs(loss.item() - ref["loss"]) <= LOSS_RTOL * max(1.0, abs(ref["loss"]))
    assert rel(tq.grad.cpu().numpy() / 4.0, ref["dQ"]) <= GRAD_RTOL and rel(tc.grad.cpu().numpy() / 4.0, ref["dC"]) <= GRAD_RTOL
    # the same step with the fused kernel switched off (three launches): logsumexp, G and gradients agree
    Nc = B * K
    if Nc % 8 == 0:
        outs = []
        for off in (0, 1):
            _lib.set_option("no_small_step", off)
            try:
                Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
                Cb = torch.empty((Nc, d), dtype=torch.bfloat16, device=dev)
                rl, lse, ls, G, dQ, dC = kn.inbatch_step_f32(t(q, dev), t(c, dev), Qb, Cb, t(y, dev), 0, t(m.astype(np.uint8), dev), 1.0 / T,
                                                             1.0 / (T * B))
                outs.append((lse.cpu().numpy(), G.float().cpu().numpy(), dQ.cpu().numpy(), dC.cpu().numpy(), ls.item(), rl.cpu().numpy()))
            finally:
                _lib.set_option("no_small_step", 0)
        a, b = outs
        assert rel(a[0], ref["lse"]) <= LOGIT_RTOL and rel(a[0], b[0]) <= 1e-5
        assert rel(a[1], b[1]) <= 1e-2 and rel(a[2], b[2]) <= 2e-3 and rel(a[3], b[3]) <= 2e-3
        assert abs(a[4] - b[4]) <= 1e-5 * max(1.0, abs(b[4])) and abs(a[4] - a[5].sum()) <= 1e-4 * max(1.0, abs(a[4]))


def test_two_row_blocks_in_the_packed_multi_rank_step(kn, dev):
    """cfg5-like ranks over a small world (W 2 x B 64 x K 2 x d 1024 -> 272 gathered columns): the packed step on the two-block fused
    kernel, loss stamp included, against the oracle of the global step."""
    meta = dict(W=2, B=64, K=2, d=1024, seed=9100, dist="U", ragged=True)
    W, B, K, d = 2, 64, 2, 1024
    parts, rows_c, Cb = _packed_world(meta, kn, dev)
    n_ctx = B * K
    Q = np.concatenate([p[0] for p in parts])
    C = np.concatenate([p[1] for p in parts])
    y = np.concatenate([p[2] + r * n_ctx for r, p in enumerate(parts)])
    m = np.concatenate([p[3] for p in parts])
    ref = O.training_step_global(Q, C, y, m, 1.0)
    Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
    dC = torch.zeros((W * rows_c, d), dtype=torch.float64, device=dev)
    local = []
    for r in range(W):
        rl, lse, ls, G, dq, dcp = kn.inbatch_step_packed_f32(t(parts[r][0], dev), Cb, Qb, W, r, n_ctx, t(parts[r][2], dev), 1.0, 1.0 / (W * B))
        local.append(ls.item())
        dC += dcp.double()
        assert rel(dq.cpu().numpy(), ref["dQ"][r * B:(r + 1) * B]) <= GRAD_RTOL
    assert abs(sum(local) / (W * B) - ref["loss"]) <= LOSS_RTOL * max(1.0, abs(ref["loss"]))
    chunks = dC.cpu().numpy().reshape(W, rows_c, d)
    assert rel(chunks[:, :n_ctx].reshape(W * n_ctx, d), ref["dC"]) <= GRAD_RTOL
    for r in range(W):
        assert abs(chunks[r, n_ctx, 0] - sum(local)) <= 1e-5 * max(1.0, abs(sum(local)))  # the piggy-backed loss numerator


@pytest.mark.parametrize("W,B,K,d", [(8, 32, 8, 768), (4, 64, 8, 512)])
def test_packed_step_narrow_sim_units_against_the_oracle(W, B, K, d, kn, dev):
    """The 64-column sim unit of the skinny plan (chosen when the 128-column units would leave most CUs idle: cfg2 gathered over 8
    ranks = B 32 x Nc 2112, sk_plan) had only HIP-against-HIP checks: here every rank's packed step against the numpy oracle of the
    reference's global step (dpr_task.py:163-212) -- loss, this rank's q.grad, and c.grad of every rank after the emulated
    reduce-scatter."""
    meta = dict(W=W, B=B, K=K, d=d, seed=8800 + W, dist="U", ragged=True)
    parts, rows_c, Cb = _packed_world(meta, kn, dev)
    n_ctx, T = B * K, 0.5
    Q = np.concatenate([p[0] for p in parts])
    C = np.concatenate([p[1] for p in parts])
    y = np.concatenate([p[2] + r * n_ctx for r, p in enumerate(parts)])
    m = np.concatenate([p[3] for p in parts])
    ref = O.training_step_global(Q, C, y, m, T)
    Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
    dC = torch.zeros((W * rows_c, d), dtype=torch.float64, device=dev)
    loss = 0.0
    for r in range(W):
        rl, lse, ls, G, dq, dcp = kn.inbatch_step_packed_f32(t(parts[r][0], dev), Cb, Qb, W, r, n_ctx, t(parts[r][2], dev), 1.0 / T,
                                                              1.0 / (T * W * B))
        loss += ls.item()
        dC += dcp.double()
        assert rel(lse.cpu().numpy(), ref["lse"][r * B:(r + 1) * B]) <= LOGIT_RTOL
        assert rel(dq.cpu().numpy(), ref["dQ"][r * B:(r + 1) * B]) <= GRAD_RTOL
    assert abs(loss / (W * B) - ref["loss"]) <= LOSS_RTOL * max(1.0, abs(ref["loss"]))
    got = dC.cpu().numpy().reshape(W, rows_c, d)[:, :n_ctx].reshape(W * n_ctx, d)
    assert rel(got, ref["dC"]) <= GRAD_RTOL


@pytest.mark.parametrize("plan", ["with-dscores", "default-plan"])
@pytest.mark.parametrize("name", golden_names("cfg3"))
@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_train_step_packed_every_rank_cfg3_in_both_wire_formats(wire, name, plan, kn, dev):
    """Every cfg3 fixture the reference wrote (U ragged T 1, U ragged T 0.05, peaky T 1), in the plan that materialises the dScores
    and in the one production takes by default (G == NULL: no dScores launch), in both wire formats of the reduce-scatter.
    dprhot_train_step_packed_f32 -- what the autograd operator runs under DDP -- for EVERY rank of cfg3 against the
    reference's global step: the loss leaves the kernel as the mean (loss_scale = 1 / Nq), the gradients are scaled by a
    DEVICE grad_output (here 8), and dC_part is written as fp32 or, for the bf16 wire of the reduce-scatter, as bf16 by the
    dC epilogue itself (each partial rounded once; summed here in fp64 like the all-pairs reduce-scatter sums in fp32)."""
    meta, g = load_golden(name)
    W, B, K, d, T = meta["W"], meta["B"], meta["K"], meta["d"], meta["T"]
    own, n_ctx = meta["own_rank"], B * K
    parts, rows_c, Cb = _packed_world(meta, kn, dev)
    want_G = plan == "with-dscores"
    if not want_G and kn._wants_g(B, W * rows_c, d):
        pytest.skip("this shape's default plan materialises the dScores")
    inv_T, go = 1.0 / T, 8.0
    dt = torch.float32 if wire == "fp32" else torch.bfloat16
    d_scale = torch.full((1,), go, device=dev)
    Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
    dC = torch.zeros((W * rows_c, d), dtype=torch.float64, device=dev)
    loss, dq_own = 0.0, None
    for r in range(W):
        rl, lse, lo, G, dq, dcp = kn.train_step_packed_f32(t(parts[r][0], dev), Cb, Qb, W, r, n_ctx, t(parts[r][2], dev), inv_T,
                                                           inv_T / (W * B), 1.0 / (W * B), d_scale, dt, want_G=want_G)
        assert dcp.dtype == dt and (G is not None) == want_G
        dC += dcp.double()
        loss += lo[0].item()  # the all-reduce of the per-rank means
        assert abs(lo[0].item() * W * B - rl.double().sum().item()) <= 1e-5 * max(1.0, rl.double().sum().item())
        if r == own:
            dq_own = dq.cpu().numpy() / go
            # the same step with dQ left as split-K slabs, finished by the operator's backward launch (here with a grad_output that
            # differs from the expected one: dQ = go2 * sum of the slabs, dC rescaled by go2 / go)
            go2 = torch.full((1,), 2.0, device=dev)
            _, _, _, _, dq_def, dcp2 = kn.train_step_packed_f32(t(parts[r][0], dev), Cb, Qb, W, r, n_ctx, t(parts[r][2], dev), inv_T,
                                                                inv_T / (W * B), 1.0 / (W * B), d_scale, dt, defer_dq=True, want_G=want_G)
            nsl = kn._lib.train_dq_slabs(B, W * rows_c, d)  # 0 where the step finishes dQ itself (the plan without a dScores launch)
            assert (isinstance(dq_def, tuple) and dq_def[1].shape[0] == nsl) if nsl else not isinstance(dq_def, tuple)
            out2 = kn.rescale_grads(dq_def, dcp2, go2, d_scale)
            assert out2.tolist() == [2.0, 2.0]
            assert rel((dq_def[0] if nsl else dq_def).cpu().numpy() / 2.0, dq_own) <= 1e-5
            assert rel(dcp2.float().cpu().numpy() / 2.0, dcp.float().cpu().numpy() / go) <= (1e-6 if wire == "fp32" else 8e-3)
    assert abs(loss - g["loss"]) <= LOSS_RTOL * max(1.0, abs(g["loss"]))
    assert rel(dq_own, g["dq_own"]) <= GRAD_RTOL
    chunk = dC.cpu().numpy().reshape(W, rows_c, d)[own] / go
    print(f"[plan-error] train-step {name} {plan} {wire}: loss rel {abs(loss - g['loss']) / max(1.0, abs(g['loss'])):.2e}  dq_own {rel(dq_own, g['dq_own']):.2e}  "
          f"dc_own_head {rel(chunk[:64], g['dc_own_head']):.2e}  dc_own_rowsum {rel(chunk[:n_ctx].sum(1), g['dc_own_rowsum']):.2e}  "
          f"dc_own_colsum {rel(chunk[:n_ctx].sum(0), g['dc_own_colsum']):.2e}")
    assert rel(chunk[:64], g["dc_own_head"]) <= GRAD_RTOL
    assert rel(chunk[:n_ctx].sum(1), g["dc_own_rowsum"]) <= GRAD_RTOL
    assert rel(chunk[:n_ctx].sum(0), g["dc_own_colsum"]) <= GRAD_RTOL


@pytest.mark.parametrize("B,K,d,ragged", [(128, 8, 4128, True), (40, 6, 8192, False), (16, 4, 30528, True), (200, 2, 4096, True),
                                          (128, 8, 4160, True), (64, 11, 4096, False), (96, 4, 6208, True), (32, 9, 4096, False), (32, 8, 30528, False)])
def test_vocabulary_wide_fp32_operands_through_the_lds_dma_sim(B, K, d, ragged, kn, dev):
    """csrc/wide.h: fp32 operands with a contraction thousands long (the CITADEL router shape class) go global -> LDS by DMA and are
    rounded to bf16 on the fragment read; split-K slabs, bf16 copy-out for the backward.  Loss / gradients against the fp64 oracle,
    the bf16 images against RNE rounding, and the register-staged kernel (option no_wide) as a second opinion on the logits."""
    from dpr_scale_amd import _lib, hotpath

    q, c, y, m = O.synth_embeddings(4000 + B, B, K, d, "U", ragged)
    ref = O.training_step_global(q, c, y, m, 1.0)
    tq, tc = t(q, dev).requires_grad_(True), t(c, dev).requires_grad_(True)
    loss = hotpath.inbatch_contrastive_loss(tq, tc, t(y, dev), t(m, dev), 1.0)
    loss.backward()
    assert abs(loss.item() - ref["loss"]) <= LOSS_RTOL * max(1.0, abs(ref["loss"]))
    assert rel(tq.grad.cpu().numpy(), ref["dQ"]) <= GRAD_RTOL and rel(tc.grad.cpu().numpy(), ref["dC"]) <= GRAD_RTOL
    Nc = B * K
    if B % 32 == 0 and B <= 128 and d % 64 == 0 and Nc % 8 == 0 and Nc <= 1536:
        # these shapes take the backward units of csrc/skinny.h (dC tiles and unsplit dQ tiles side by side, one launch): the same G
        # through the generic pair kernel (option no_wide_bwd) must give the same gradients up to the summation order of the K ranges
        _lib.set_option("no_wide_bwd", 1)
        try:
            tq2, tc2 = t(q, dev).requires_grad_(True), t(c, dev).requires_grad_(True)
            hotpath.inbatch_contrastive_loss(tq2, tc2, t(y, dev), t(m, dev), 1.0).backward()
        finally:
            _lib.set_option("no_wide_bwd", 0)
        assert rel(tc2.grad.cpu().numpy(), tc.grad.cpu().numpy()) <= 1e-5 and rel(tq2.grad.cpu().numpy(), tq.grad.cpu().numpy()) <= 1e-5
    # the forward alone, both kernels: logits bit-comparable up to the summation order of the slabs
    if Nc % 8 == 0:
        outs = []
        for no_wide in (0, 1):
            _lib.set_option("no_wide", no_wide)
            try:
                Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
                Cb = torch.empty((Nc, d), dtype=torch.bfloat16, device=dev)
                rl, lse, ls, G, S = kn.inbatch_fwd_f32(t(q, dev), t(c, dev), Qb, Cb, t(y, dev), 0, t(m.astype(np.uint8), dev), 1.0, 1.0 / B,
                                                       want_logits=True)
                outs.append((S.cpu().numpy(), Qb.float().cpu().numpy(), Cb.float().cpu().numpy(), ls.item()))
            finally:
                _lib.set_option("no_wide", 0)
        (S1, Q1, C1, l1), (S0, Q0, C0, l0) = outs
        assert np.array_equal(Q1, O.bf16_round(q)) and np.array_equal(C1, O.bf16_round(c))  # (inputs are bf16-representable: identity)
        assert np.array_equal(Q0, Q1) and np.array_equal(C0, C1)
        fin = np.isfinite(ref["S"])
        assert np.array_equal(fin, np.isfinite(S1)) and rel(S1[fin], ref["S"][fin]) <= LOGIT_RTOL and rel(S0[fin], S1[fin]) <= 1e-5
        assert abs(l1 - l0) <= 1e-5 * max(1.0, abs(l0))


def test_selftest_big_passes():
    """The stand-alone C ABI self-test (csrc/selftest.hip) including its cfg3-per-rank (128 x 8192 x 768) and 1M-passage
    search cases, executed as the binary a C caller would link."""
    import os
    import subprocess

    from conftest import ROOT

    exe = os.path.join(ROOT, "dpr_scale_amd", "selftest")
    assert os.path.isfile(exe), "dpr_scale_amd/selftest missing: run __graft_entry__.build()"
    out = subprocess.run([exe, "big"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "SELFTEST PASSED" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_tie_rule_bit_exact(dev):
    from dpr_scale_amd.hotpath import rank_of_gold, topk

    _, g = load_golden("ties")
    S = t(g["S"], dev)
    assert np.array_equal(rank_of_gold(S, t(g["y"], dev)).cpu().numpy(), g["ranks"])
    v, i = topk(S, 16)
    assert np.array_equal(i.cpu().numpy(), g["order16"])
    assert np.array_equal(v.cpu().numpy(), g["topk_values"])


@pytest.mark.parametrize("rows,cols,k,levels", [(5, 1000, 10, 4), (3, 70000, 100, 0), (64, 9000, 128, 50), (2, 3, 3, 2),
                                               (4, 20000, 1, 0), (3, 50000, 1000, 0), (2, 1500, 1024, 7), (5, 30000, 300, 20),
                                               # empty-state bound + counting merges (k <= 256): heavy ties, all-equal scores,
                                               # k at the variant boundary, partial / rotated windows
                                               (8, 65536, 100, 3), (4, 40000, 64, 1), (8, 65536, 256, 0), (6, 4608, 200, 0),
                                               (1030, 16384, 100, 0), (3, 12289, 257, 5)])
def test_streaming_topk_equals_stable_sort_prefix(rows, cols, k, levels, kn, dev):
    """torch.topk of run_retrieval_pytorch.py:149-150 under the frozen tie rule, in one piece and folded over pieces
    (the shard re-merge of :272-277); indices bit-exact."""
    g = torch.Generator().manual_seed(cols)
    S = (torch.randint(0, levels, (rows, cols), generator=g).float() if levels else torch.randn(rows, cols, generator=g)).to(dev)
    if cols > 100:
        S[:, 17] = float("-inf")
    order = torch.sort(S, dim=1, descending=True, stable=True).indices[:, :k]
    v, i = kn.topk(S, k)
    assert torch.equal(i, order) and torch.equal(v, S.gather(1, order))
    v2 = torch.empty_like(v)
    i2 = torch.empty_like(i)
    edges = [0, cols // 5, cols // 5 + 1, cols // 2, cols]
    first = True
    for a, b in zip(edges[:-1], edges[1:]):
        if b > a:
            kn.topk_update(S[:, a:], b - a, 7 + a, v2, i2, first)
            first = False
    assert torch.equal(i2, order + 7) and torch.equal(v2, v)
    # against the oracle fold as well (small cases)
    if rows * cols <= 100000:
        st = None
        for a, b in zip(edges[:-1], edges[1:]):
            if b > a:
                st = O.topk_merge(st, S[:, a:b].cpu().numpy(), 7 + a, k)
        assert np.array_equal(st[1], i2.cpu().numpy()) and np.array_equal(st[0], v2.cpu().numpy())


@pytest.mark.parametrize("k", [100, 256, 300])
@pytest.mark.parametrize("pattern", ["ascending", "descending", "mostly_masked", "sawtooth"])
def test_streaming_topk_adversarial_orders(pattern, k, kn, dev):
    """Orders that defeat the streaming kernel's shortcuts: every new score beats all earlier ones (each window qualifies whole,
    the fold path), none does (pure stream), almost everything masked (-inf ties broken by id), a sawtooth (ties across windows)."""
    rows, cols = 5, 20000
    j = torch.arange(cols, dtype=torch.float32)
    if pattern == "ascending":
        S = j.repeat(rows, 1) + torch.arange(rows).float()[:, None]
    elif pattern == "descending":
        S = (cols - j).repeat(rows, 1)
    elif pattern == "mostly_masked":
        S = torch.full((rows, cols), float("-inf"))
        S[:, 7::997] = torch.randn(rows, len(range(7, cols, 997)), generator=torch.Generator().manual_seed(k))
    else:
        S = (j % 37).repeat(rows, 1)
    S = S.to(dev)
    order = torch.sort(S, dim=1, descending=True, stable=True).indices[:, :k]
    v, i = kn.topk(S, k)
    assert torch.equal(i, order) and torch.equal(v, S.gather(1, order))
    v2, i2 = torch.empty_like(v), torch.empty_like(i)
    first = True
    for a, b in ((0, 3000), (3000, 3001), (3001, 12000), (12000, cols)):
        kn.topk_update(S[:, a:], b - a, a, v2, i2, first)
        first = False
    assert torch.equal(i2, order) and torch.equal(v2, v)


@pytest.mark.parametrize("k", [64, 300, 1000])
def test_topk_never_selects_nan_whatever_the_start_up_path(k, kn, dev):
    """A NaN score never qualifies: with fewer than k other values in a row the remaining slots stay empty (-inf / -1) -- the same
    answer from the counting start (k <= 256), the radix-select start (k > 256, >= 4096 columns) and the streaming update."""
    rows, cols = 6, 8192
    gen = torch.Generator(device="cpu").manual_seed(k)
    S = torch.randn(rows, cols, generator=gen)
    keep = [k + 40, k, k - 1, 5, 0, cols]  # finite values per row
    for r, n in enumerate(keep):
        idx = torch.randperm(cols, generator=gen)[n:]
        S[r, idx] = float("nan")
    v, i = kn.topk(S.to(dev), k)
    v, i = v.cpu(), i.cpu()
    vv, ii = torch.empty((rows, k), device=dev), torch.empty((rows, k), dtype=torch.int64, device=dev)
    Sd = S.to(dev).contiguous()
    kn.topk_update(Sd[:, :4096].contiguous(), 4096, 0, vv, ii, True)   # the same row folded in two pieces
    kn.topk_update(Sd[:, 4096:].contiguous(), 4096, 4096, vv, ii, False)
    for r, n in enumerate(keep):
        fin = torch.nan_to_num(S[r], nan=float("-inf"))
        order = torch.sort(fin, descending=True, stable=True).indices[:min(n, k)]
        m = order.numel()
        assert torch.equal(i[r, :m], order) and torch.equal(v[r, :m], S[r, order]), (k, r)
        assert torch.all(i[r, m:] == -1) and torch.all(v[r, m:] == float("-inf")), (k, r)
        assert torch.equal(ii[r].cpu(), i[r]) and torch.equal(vv[r].cpu(), v[r]), (k, r)


def test_topk_state_with_fewer_columns_than_k(kn, dev):
    S = torch.tensor([[1.0, 3.0, 2.0, 3.0]], device=dev)
    v = torch.empty((1, 6), device=dev)
    i = torch.empty((1, 6), dtype=torch.int64, device=dev)
    kn.topk_update(S, 4, 10, v, i, True)
    assert i.tolist() == [[11, 13, 12, 10, -1, -1]] and v[0, :4].tolist() == [3.0, 3.0, 2.0, 1.0]
    assert torch.isneginf(v[0, 4:]).all()
    kn.topk_update(torch.tensor([[5.0, 0.0, 3.0, 1.0]], device=dev), 3, 0, v, i, False)  # 4th column not scanned
    assert i.tolist() == [[0, 2, 11, 13, 12, 10]]


@pytest.mark.parametrize("nq,shards,d,k,chunk", [(9, (4000, 13, 8, 2051), 64, 20, 1024), (130, (50000,), 768, 100, 8192),
                                                 (512, (100000, 40000), 128, 50, 32768),  # >= 256 tiles per chunk: the gemm8p.h filter epilogue
                                                 (512, (300000,), 128, 50, 32768),  # warm chunks merged in groups of 2, then 4 (dprhot_search)
                                                 (300, (200000, 70000), 128, 1000, 32768),  # ... at k = 1000, a second shard behind them
                                                 (7, (30000, 5000), 128, 1000, 4096),  # --topk 1000: dragon/README recipes
                                                 (40, (60000, 9000), 128, 300, 8192)])  # 48 KB variant, counted candidate merges
def test_corpus_search_equals_topk_of_full_score_matrix(nq, shards, d, k, chunk, kn, dev):
    """search_index + shard loop (run_retrieval_pytorch.py:141-166, :196-243, :272-277): ids bit-exact against
    the stable top-k of the full score matrix computed by the same similarity kernel; scores against fp32 torch."""
    from dpr_scale_amd.hotpath import CorpusSearch, sim_score

    from dpr_scale_amd import _lib

    g = torch.Generator().manual_seed(nq)
    q = torch.randn(nq, d, generator=g).to(dev)
    parts = [torch.randn(n, d, generator=g).to(dev) for n in shards]
    parts[0][100:140] = parts[0][60:100]  # duplicated passages: exact score ties
    s = CorpusSearch(q, k, chunk=chunk, kernels=kn)
    first = 0
    for p in parts:
        s.add(p.to(torch.bfloat16) if first == 0 else p, first)  # bf16-resident and fp32 shards
        first += p.shape[0]
    v, i = s.result()
    C = torch.cat(parts)
    S = sim_score(q, C, kernels=kn)
    order = torch.sort(S, dim=1, descending=True, stable=True).indices[:, :k]
    assert torch.equal(i, order) and torch.equal(v, S.gather(1, order))
    ref = q.to(torch.bfloat16).float() @ C.to(torch.bfloat16).float().T
    assert (v - ref.gather(1, i)).abs().max().item() <= 1e-3 * ref.abs().max().item()


def test_corpus_search_worst_case_order_and_reversed_shards(kn, dev):
    """Every later passage beats everything seen so far (all scores pass the in-GEMM filter), duplicated passages
    tie exactly, and the shards arrive in decreasing id order -- the result is still the stable top-k."""
    from dpr_scale_amd.hotpath import CorpusSearch, sim_score

    d, n, k = 64, 6000, 50
    q = torch.ones(5, d, device=dev)
    scale = torch.arange(n, device=dev).float().div(64).floor()  # groups of 64 identical passages, increasing score
    C = (scale[:, None] * torch.ones(n, d, device=dev) / 64).to(torch.bfloat16)
    S = sim_score(q, C, kernels=kn)
    order = torch.sort(S, dim=1, descending=True, stable=True).indices[:, :k]
    for bounds in ([0, 1000, 2048, 6000], [6000, 2048, 1000, 0]):
        s = CorpusSearch(q, k, chunk=512, kernels=kn)
        for a, b in zip(bounds[:-1], bounds[1:]):
            lo, hi = min(a, b), max(a, b)
            s.add(C[lo:hi], lo)
        v, i = s.result()
        assert torch.equal(i, order) and torch.equal(v, S.gather(1, order))


@pytest.mark.parametrize("B,K,d", [(128, 3, 768), (128, 4, 768), (128, 5, 768), (96, 8, 512), (64, 12, 1024), (32, 40, 768)])
def test_operator_on_the_mid_size_shapes_of_the_few_rows_plan(B, K, d, kn, dev):
    """Shapes the few-rows plan took over at the end of round 3 (csrc/dprhot.hip sk_plan: from 256 / 512 / 1024 columns depending on
    the rows; up to 512 contexts its dQ units are unsplit and dprhot_train_dq_slabs reports no slabs): the autograd operator with a
    non-unit grad_output against the oracle."""
    from dpr_scale_amd import _lib, hotpath

    q, c, y, m = O.synth_embeddings(7300 + B + K, B, K, d, "U", False)
    ref = O.training_step_global(q, c, y, m, 0.5)
    Nc = B * K
    assert (_lib.train_dq_slabs(B, Nc, d) == 0) == ((Nc + 63) // 64 <= 8)  # (every shape of this test is on the few-rows plan)
    tq, tc = t(q, dev).requires_grad_(True), t(c, dev).requires_grad_(True)
    loss = hotpath.inbatch_contrastive_loss(tq, tc, t(y, dev), t(m, dev), 0.5)
    (loss * 64.0).backward()
    assert abs(loss.item() - ref["loss"]) <= LOSS_RTOL * max(1.0, abs(ref["loss"]))
    assert rel(tq.grad.cpu().numpy() / 64.0, ref["dQ"]) <= GRAD_RTOL and rel(tc.grad.cpu().numpy() / 64.0, ref["dC"]) <= GRAD_RTOL


@pytest.mark.parametrize("B,K,d", [(32, 8, 768), (4, 2, 128), (8, 64, 768), (5, 5, 80), (64, 16, 768), (32, 66, 768), (128, 3, 768), (96, 5, 512)])
def test_whole_step_call_equals_forward_plus_backward(B, K, d, kn, dev):
    """dprhot_inbatch_step_f32 (sim + ONE softmax/dScores/dQ/dC kernel at the small shapes; three launches otherwise)
    against dprhot_inbatch_fwd_f32 followed by dprhot_inbatch_bwd, and against the oracle."""
    n = B * K
    Nc = (n + 7) // 8 * 8
    q, c, y, m = O.synth_embeddings(B * 1000 + K, B, K, d, "U", True)
    cp = np.zeros((Nc, d), np.float32)
    cp[:n] = c
    mp = np.ones(Nc, np.uint8)
    mp[:n] = m
    tq, tc, ty, tm = t(q, dev), t(cp, dev), t(y, dev), t(mp, dev)
    T = 0.7
    Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
    Cb = torch.empty((Nc, d), dtype=torch.bfloat16, device=dev)
    rl, lse, ls, G, dQ, dC = kn.inbatch_step_f32(tq, tc, Qb, Cb, ty, 0, tm, 1.0 / T, 1.0 / (T * B))
    Qb2, Cb2 = torch.empty_like(Qb), torch.empty_like(Cb)
    rl2, lse2, ls2, G2, _ = kn.inbatch_fwd_f32(tq, tc, Qb2, Cb2, ty, 0, tm, 1.0 / T, 1.0 / (T * B))
    dQ2, dC2 = kn.inbatch_bwd(G2, Qb2, Cb2, 1.0, None)
    assert torch.equal(Qb, Qb2) and torch.equal(Cb, Cb2)
    assert rel(rl.cpu().numpy(), rl2.cpu().numpy()) <= 1e-6 and abs(ls.item() - ls2.item()) <= 1e-5 * max(1.0, abs(ls2.item()))
    assert rel(G.float().cpu().numpy(), G2.float().cpu().numpy()) <= 8e-3  # one bf16 ulp where the logsumexp differs in its last bit
    assert rel(dQ.cpu().numpy(), dQ2.cpu().numpy()) <= 2e-3 and rel(dC.cpu().numpy(), dC2.cpu().numpy()) <= 2e-3
    r = O.training_step_global(O.bf16_round(q), O.bf16_round(c), y, m, T)
    assert abs(ls.item() / B - r["loss"]) <= 1e-3 * max(1.0, abs(r["loss"]))
    assert rel(dQ.cpu().numpy(), r["dQ"]) <= GRAD_RTOL and rel(dC[:n].cpu().numpy(), r["dC"]) <= GRAD_RTOL


@pytest.mark.parametrize("B,K,d", [(6, 3, 1002), (4, 2, 30522)])
def test_hidden_size_not_a_multiple_of_8_is_zero_padded(B, K, d, dev):
    """Router-style wide vectors (citadel_task.py:249-262 scores [B, vocab = 30522] representations with the same
    sim_score + CrossEntropyLoss): d % 8 != 0 goes through zero padding, gradients come back in the original width."""
    from dpr_scale_amd.hotpath import inbatch_contrastive_loss, sim_score

    q, c, y, m = O.synth_embeddings(11, B, K, d, "U", True)
    r = O.training_step_global(q, c, y, m, 1.0)
    tq, tc = t(q, dev).requires_grad_(True), t(c, dev).requires_grad_(True)
    loss = inbatch_contrastive_loss(tq, tc, t(y, dev), t(m, dev), 1.0, False)
    loss.backward()
    assert abs(loss.item() - r["loss"]) <= 1e-3 * max(1.0, abs(r["loss"]))
    assert tq.grad.shape == (B, d) and tc.grad.shape == (B * K, d)
    assert rel(tq.grad.cpu().numpy(), r["dQ"]) <= GRAD_RTOL and rel(tc.grad.cpu().numpy(), r["dC"]) <= GRAD_RTOL
    S = sim_score(t(q, dev), t(c, dev)).cpu().numpy()
    Sm = np.where(m[None, :], -np.inf, r["S"])
    fin = np.isfinite(Sm)
    assert np.abs(S[:, ~m] - r["S"][:, ~m]).max() <= 1e-3 * np.abs(r["S"][fin]).max()


@pytest.mark.parametrize("name", ["cfg2_U_T1", "cfg2_Ur_T0.05", "cfg1_P_T1"])
def test_fp32_g_debug_mode_meets_1e3_on_gradients(name, dev, monkeypatch):
    """with DPRHOT_FP32_G=1 dScores travel as a bf16 pair (hi + lo) through the same MFMA GEMMs and
    the gradients agree with the fp32 reference to 1e-3 of max (the default single-bf16 G gives ~2e-3, bar 1e-2)."""
    from dpr_scale_amd.hotpath import inbatch_contrastive_loss

    monkeypatch.setenv("DPRHOT_FP32_G", "1")
    meta, g = load_golden(name)
    q, c, y, m = rank_inputs(meta)[0]
    tq, tc = t(q, dev).requires_grad_(True), t(c, dev).requires_grad_(True)
    loss = inbatch_contrastive_loss(tq, tc, t(y, dev), t(m, dev), meta["T"])
    (loss * 8.0).backward()
    assert abs(loss.item() - g["loss"]) <= LOSS_RTOL * max(1.0, abs(g["loss"]))
    assert rel(tq.grad.cpu().numpy() / 8.0, g["dQ"]) <= 1e-3
    assert rel(tc.grad.cpu().numpy() / 8.0, g["dC"]) <= 1e-3


def test_non_finite_loss_is_published_as_non_finite(kn, dev):
    """A masked gold column gives loss = +inf in nn.CrossEntropyLoss; the multi-workgroup loss accumulation (fixed-point
    integer atomics) must publish inf, and a NaN logit row NaN -- never a finite-looking number."""
    B, Nc, d = 512, 2048, 64  # long-row plan, several row blocks
    gen = torch.Generator().manual_seed(3)
    q = torch.randn(B, d, generator=gen).to(torch.bfloat16).to(dev)
    c = torch.randn(Nc, d, generator=gen).to(torch.bfloat16).to(dev)
    y = torch.randint(0, Nc, (B,), generator=gen).to(dev)
    mask = torch.zeros(Nc, dtype=torch.uint8, device=dev)
    _, _, ls0, _, _ = kn.inbatch_fwd(q, c, y, 0, mask, 1.0, 1.0 / B)
    assert torch.isfinite(ls0).all()
    mask[y[300]] = 1  # the gold column of row 300 is a dummy context
    _, _, ls1, _, _ = kn.inbatch_fwd(q, c, y, 0, mask, 1.0, 1.0 / B)
    assert torch.isinf(ls1).all() and ls1.item() > 0
    q2 = q.clone()
    q2[17] = float("nan")
    mask.zero_()
    _, _, ls2, _, _ = kn.inbatch_fwd(q2, c, y, 0, mask, 1.0, 1.0 / B)
    assert torch.isnan(ls2).all()


def test_backward_twice_with_retain_graph(dev):
    """The step runs in forward when a backward will follow; a second backward (retain_graph=True) must give the same
    gradients again (accumulated), not fail on missing state."""
    from dpr_scale_amd.hotpath import inbatch_contrastive_loss

    q, c, y, m = O.synth_embeddings(21, 8, 4, 64, "U", False)
    tq, tc = t(q, dev).requires_grad_(True), t(c, dev).requires_grad_(True)
    loss = inbatch_contrastive_loss(tq, tc, t(y, dev), t(m, dev), 1.0, False)
    loss.backward(retain_graph=True)
    g1q, g1c = tq.grad.clone(), tc.grad.clone()
    loss.backward()
    assert torch.allclose(tq.grad, 2 * g1q) and torch.allclose(tc.grad, 2 * g1c)


def test_non_inbatch_window_branch(kn, dev):
    """in_batch_negatives=False (dpr_task.py:198-207): row i sees only its own K columns."""
    meta, g = load_golden("nib")
    q, c, y, m = O.synth_embeddings(meta["seed"], meta["B"], meta["K"], meta["d"], meta["dist"], meta["ragged"])
    B, K = meta["B"], meta["K"]
    Qb, Cb = bf16(q, dev), bf16(c, dev)
    S = kn.sim(Qb, Cb, t(m.astype(np.uint8), dev), 1.0)
    row_loss, _, G = kn.softmax_ce(S, t(y, dev), 0, 1.0 / B, want_G=True, row_win_start=t(y, dev), win_len=K)
    assert abs(row_loss.mean().item() - g["loss"]) <= LOSS_RTOL * max(1.0, abs(g["loss"]))
    assert rel(kn.dq(G, Cb).cpu().numpy(), g["dQ"]) <= GRAD_RTOL
    assert rel(kn.dc(G, Qb).cpu().numpy(), g["dC"]) <= GRAD_RTOL


# ---- size-independent properties at full size -----------------------------------------------------------
@pytest.mark.parametrize("B,Nc,d", [(128, 8192, 768), (1024, 8192, 768), (64, 1024, 1024), (32, 2112, 768), (16, 4096, 128),
                                    (48, 264, 64)])
def test_properties_full_size(B, Nc, d, kn, dev):
    gen = torch.Generator(device="cpu").manual_seed(7)
    q = (torch.randn(B, d, generator=gen) * d ** -0.25).to(torch.bfloat16)
    c = (torch.randn(Nc, d, generator=gen) * d ** -0.25).to(torch.bfloat16)
    y = torch.randint(0, Nc, (B,), generator=gen)
    mask = (torch.rand(Nc, generator=gen) < 0.05)
    mask[y] = False
    Qb, Cb, yd, md = q.to(dev), c.to(dev), y.to(dev), mask.to(torch.uint8).to(dev)
    row_loss, lse, loss_sum, G, S = kn.inbatch_fwd(Qb, Cb, yd, 0, md, 1.0, 1.0 / B, want_logits=True)
    # (1) logits vs a plain fp32 torch matmul on the device (same bf16-representable inputs)
    S_ref = Qb.float() @ Cb.float().T
    S_ref[:, mask.to(dev)] = float("-inf")
    fin = torch.isfinite(S_ref)
    assert torch.equal(fin, torch.isfinite(S))
    assert ((S[fin] - S_ref[fin]).abs().max() / S_ref[fin].abs().max()).item() <= LOGIT_RTOL
    # (2) lse >= max logit, loss >= 0, loss_sum == sum(row_loss), lse vs torch.logsumexp
    assert torch.all(lse >= S.max(dim=1).values - 1e-5) and torch.all(row_loss >= -1e-5)
    assert abs(loss_sum.item() - row_loss.double().sum().item()) <= 1e-4 * max(1.0, row_loss.double().sum().item())
    assert ((lse - torch.logsumexp(S_ref, dim=1)).abs().max() / lse.abs().max()).item() <= LOGIT_RTOL
    # (3) rows of G sum to zero (softmax - onehot), masked columns carry exactly zero
    Gf = G.float()
    assert Gf.sum(dim=1).abs().max().item() <= 2e-2 / B
    assert torch.all(Gf[:, mask.to(dev)] == 0)
    # (4) backward is linear in grad_output and equals fp32 torch GEMMs on the same G
    one = torch.ones(1, device=dev)
    two = torch.full((1,), 2.0, device=dev)
    dq1, dc1 = kn.inbatch_bwd(G, Qb, Cb, 1.0, one)
    dq2, dc2 = kn.inbatch_bwd(G, Qb, Cb, 1.0, two)
    assert torch.equal(dq2, 2 * dq1) and torch.equal(dc2, 2 * dc1)
    dq_ref, dc_ref = Gf @ Cb.float(), Gf.T @ Qb.float()
    assert ((dq1 - dq_ref).abs().max() / dq_ref.abs().max()).item() <= 1e-4
    assert ((dc1 - dc_ref).abs().max() / dc_ref.abs().max()).item() <= 1e-4
    # (5) a column permutation of C (labels and mask permuted with it) changes neither loss nor ranks
    perm = torch.randperm(Nc, generator=gen)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(Nc)
    rl2, _, ls2, _, S2 = kn.inbatch_fwd(Qb, Cb[perm.to(dev)].contiguous(), inv[y].to(dev), 0, md[perm.to(dev)].contiguous(),
                                        1.0, 1.0 / B, want_logits=True)
    assert abs(ls2.item() - loss_sum.item()) <= 1e-5 * max(1.0, abs(loss_sum.item()))
    greater1 = (S > S.gather(1, yd[:, None])).sum(1)
    greater2 = (S2 > S2.gather(1, inv[y].to(dev)[:, None])).sum(1)
    assert torch.equal(greater1, greater2)
    # (6) rank of gold == position in a stable descending sort; top-k == stable sort prefix
    order = torch.sort(S, dim=1, descending=True, stable=True).indices
    ref_rank = (order == yd[:, None]).nonzero()[:, 1] + 1
    assert torch.equal(kn.rank_of_gold(S, yd), ref_rank)
    v, i = kn.topk(S, 10)
    assert torch.equal(i, order[:, :10]) and torch.equal(v, S.gather(1, order[:, :10]))


# ---- edge cases -------------------------------------------------------------------------------------------
def test_edge_single_row_and_minimal_columns(kn, dev):
    q, c, y, m = O.synth_embeddings(3, 1, 8, 64, "U", False)
    r = O.training_step_global(q, c, y, m, 1.0)
    from dpr_scale_amd.hotpath import inbatch_contrastive_loss

    tq, tc = t(q, dev).requires_grad_(True), t(c, dev).requires_grad_(True)
    loss = inbatch_contrastive_loss(tq, tc, t(y, dev), t(m, dev), 1.0)
    loss.backward()
    assert abs(loss.item() - r["loss"]) <= LOSS_RTOL * max(1.0, abs(r["loss"]))
    assert rel(tq.grad.cpu().numpy(), r["dQ"]) <= GRAD_RTOL and rel(tc.grad.cpu().numpy(), r["dC"]) <= GRAD_RTOL


def test_edge_ragged_column_count_is_padded(dev):
    """B*K not a multiple of 8 (K=3): the wrapper pads with masked columns; result equals the oracle."""
    from dpr_scale_amd.hotpath import inbatch_contrastive_loss

    q, c, y, m = O.synth_embeddings(11, 5, 3, 128, "U", True)
    r = O.training_step_global(q, c, y, m, 0.5)
    tq, tc = t(q, dev).requires_grad_(True), t(c, dev).requires_grad_(True)
    loss = inbatch_contrastive_loss(tq, tc, t(y, dev), t(m, dev), 0.5)
    loss.backward()
    assert abs(loss.item() - r["loss"]) <= LOSS_RTOL * max(1.0, abs(r["loss"]))
    assert tc.grad.shape == (15, 128)
    assert rel(tq.grad.cpu().numpy(), r["dQ"]) <= GRAD_RTOL and rel(tc.grad.cpu().numpy(), r["dC"]) <= GRAD_RTOL


def test_edge_everything_but_gold_masked(kn, dev):
    q, c, y, m = O.synth_embeddings(5, 8, 4, 128, "U", False)
    m[:] = True
    m[y] = False
    Qb, Cb = bf16(q, dev), bf16(c, dev)
    row_loss, lse, loss_sum, G, S = kn.inbatch_fwd(Qb, Cb, t(y, dev), 0, t(m.astype(np.uint8), dev), 1.0, 1.0 / 8,
                                                   want_logits=True)
    r = O.training_step_global(q, c, y, m, 1.0)
    assert abs(loss_sum.item() / 8 - r["loss"]) <= LOSS_RTOL * max(1.0, abs(r["loss"]))
    assert torch.isfinite(G.float()).all() and torch.isfinite(row_loss).all()


def test_edge_peaky_logits_and_small_temperature(kn, dev):
    """dist P (logit sigma ~ 28) at T = 0.05: |logits| in the thousands; no overflow, loss matches oracle."""
    q, c, y, m = O.synth_embeddings(9, 32, 8, 768, "P", True)
    r = O.training_step_global(q, c, y, m, 0.05)
    Qb, Cb = bf16(q, dev), bf16(c, dev)
    row_loss, lse, loss_sum, G, _ = kn.inbatch_fwd(Qb, Cb, t(y, dev), 0, t(m.astype(np.uint8), dev), 20.0, 20.0 / 32)
    assert torch.isfinite(row_loss).all() and torch.isfinite(G.float()).all()
    assert abs(loss_sum.item() / 32 - r["loss"]) <= LOSS_RTOL * abs(r["loss"])
    assert rel(row_loss.cpu().numpy(), r["row_loss"]) <= LOSS_RTOL


def test_bad_arguments_raise(kn, dev):
    from dpr_scale_amd._lib import DprhotError

    Qb = torch.zeros((4, 60), dtype=torch.bfloat16, device=dev)  # d % 8 != 0
    Cb = torch.zeros((8, 60), dtype=torch.bfloat16, device=dev)
    with pytest.raises(DprhotError, match="multiple of 8"):
        kn.sim(Qb, Cb)


# ---- operator-level variants ---------------------------------------------------------------------------------
@pytest.mark.parametrize("qdt,cdt", [(torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16), (torch.float16, torch.float32)])
def test_operator_accepts_reduced_precision_inputs(qdt, cdt, dev):
    """Encoder heads with a projection return bf16/fp16 under autocast: the operator takes them (values are
    bf16-representable here, so the result is the fp32 fixture's)."""
    from dpr_scale_amd.hotpath import inbatch_contrastive_loss

    meta, g = load_golden("cfg2_Ur_T0.05")
    q, c, y, m = rank_inputs(meta)[0]
    tq = t(q, dev).to(qdt).requires_grad_(True)
    tc = t(c, dev).to(cdt).requires_grad_(True)
    loss = inbatch_contrastive_loss(tq, tc, t(y, dev), t(m, dev), meta["T"])
    loss.backward()
    assert tq.grad.dtype == qdt and tc.grad.dtype == cdt
    assert abs(loss.item() - g["loss"]) <= LOSS_RTOL * max(1.0, abs(g["loss"]))
    assert rel(tq.grad.float().cpu().numpy(), g["dQ"]) <= 2 * GRAD_RTOL
    assert rel(tc.grad.float().cpu().numpy(), g["dC"]) <= 2 * GRAD_RTOL


def test_operator_with_only_query_gradient_and_noncontiguous_inputs(dev):
    from dpr_scale_amd.hotpath import inbatch_contrastive_loss

    meta, g = load_golden("cfg2_U_T1")
    q, c, y, m = rank_inputs(meta)[0]
    big = torch.zeros((q.shape[0], 2 * q.shape[1]), device=dev)
    big[:, ::2] = t(q, dev)
    tq = big[:, ::2].detach().requires_grad_(True)  # strided view
    tc = t(c, dev)  # no grad wanted for the contexts
    loss = inbatch_contrastive_loss(tq, tc, t(y, dev), t(m, dev), meta["T"])
    loss.backward()
    assert tc.grad is None
    assert rel(tq.grad.cpu().numpy(), g["dQ"]) <= GRAD_RTOL


def test_large_batch_takes_the_cast_then_bf16_path(kn, dev):
    """B > 128 through the fp32 entry point (internally dprhot_prep + the bf16-operand GEMM) vs the oracle."""
    q, c, y, m = O.synth_embeddings(21, 160, 4, 256, "U", True)
    r = O.training_step_global(q, c, y, m, 1.0)
    Qb = torch.empty((160, 256), dtype=torch.bfloat16, device=dev)
    Cb = torch.empty((640, 256), dtype=torch.bfloat16, device=dev)
    row_loss, lse, loss_sum, G, S = kn.inbatch_fwd_f32(t(q, dev), t(c, dev), Qb, Cb, t(y, dev), 0, t(m.astype(np.uint8), dev),
                                                       1.0, 1.0 / 160, want_logits=True)
    assert np.array_equal(Qb.float().cpu().numpy(), q) and np.array_equal(Cb.float().cpu().numpy(), c)
    assert abs(loss_sum.item() / 160 - r["loss"]) <= LOSS_RTOL * max(1.0, abs(r["loss"]))
    fin = np.isfinite(r["S"])
    assert rel(S.cpu().numpy()[fin], r["S"][fin]) <= LOGIT_RTOL
    dq, dcp = kn.inbatch_bwd(G, Qb, Cb, 1.0, torch.ones(1, device=dev))
    assert rel(dq.cpu().numpy(), r["dQ"]) <= GRAD_RTOL and rel(dcp.cpu().numpy(), r["dC"]) <= GRAD_RTOL


def test_short_and_long_forward_plans_agree(kn, dev, monkeypatch):
    """The two forward plans (split-K slabs + in-register softmax vs per-tile statistics + streaming pass) on the
    same inputs: B = 64 (short) against the same rows inside a B = 72 problem (long plan: B > 64)."""
    q, c, y, m = O.synth_embeddings(33, 72, 8, 128, "U", True)
    Cb, mask = bf16(c, dev), t(m.astype(np.uint8), dev)
    outs = []
    for B in (64, 72):
        Qb = bf16(q[:B], dev)
        row_loss, lse, _, G, S = kn.inbatch_fwd(Qb, Cb, t(y[:B], dev), 0, mask, 1.0, 1.0 / 72, want_logits=True)
        outs.append((row_loss[:64].cpu().numpy(), lse[:64].cpu().numpy(), G[:64].float().cpu().numpy(), S[:64].cpu().numpy()))
    assert rel(outs[0][0], outs[1][0]) <= 1e-5 and rel(outs[0][1], outs[1][1]) <= 1e-6
    assert np.abs(outs[0][2] - outs[1][2]).max() <= 2.0 ** -8 * np.abs(outs[1][2]).max()  # bf16 G: at most 1 ulp apart
    fin = np.isfinite(outs[1][3])
    assert np.array_equal(fin, np.isfinite(outs[0][3])) and rel(outs[0][3][fin], outs[1][3][fin]) <= 1e-6


# ---- large shapes: the no-logits forward (gemm8p.h) and the score-free rank --------------------------------------------------
# >= 256 tiles of 256 x 256; ragged rows and columns.  From the fourth on: the widths BASELINE uses (d = 768: 12 K steps per tile, d = 1024:
# 16 -- the persistent K pipeline wraps into the next tile differently than at 2 / 4 steps), ragged rows and a column count that is no
# multiple of 256 included (VERDICT r5 "what's weak" #1)
NL_SHAPES = [(4096, 4096, 256), (1000, 16392, 128), (300, 70000, 128),
             (4096, 4096, 768), (2048, 16384, 768), (1024, 32768, 1024), (4100, 4360, 768),
             (1024, 8192, 768), (1500, 6216, 256)]  # (the last two: 128 / 150 tiles -- one workgroup per tile, the gate of round 6: option nl_min)


def _nl_problem(B, Nc, d, seed, dev, dup=False):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    q = (torch.randn(B, d, generator=gen) * d ** -0.25).to(torch.bfloat16)
    c = (torch.randn(Nc, d, generator=gen) * d ** -0.25).to(torch.bfloat16)
    if dup:  # exact score ties: copies of context rows spread over the matrix
        src = torch.randint(0, Nc, (Nc // 16,), generator=gen)
        dst = torch.randint(0, Nc, (Nc // 16,), generator=gen)
        c[dst] = c[src]
    y = torch.randint(0, Nc, (B,), generator=gen)
    mask = torch.rand(Nc, generator=gen) < 0.05
    mask[y[: B // 2]] = False  # half of the gold columns may be masked (-inf gold logit)
    return q.to(dev), c.to(dev), y.to(dev), mask.to(torch.uint8).to(dev)


@pytest.mark.parametrize("B,Nc,d", NL_SHAPES)
def test_no_logits_forward_equals_the_two_launch_plan(B, Nc, d, kn, dev):
    """dprhot_inbatch_fwd without S_out at a large shape runs statistics pass -> logsumexp -> dScores pass (logits recomputed, never
    stored); with S_out it runs the round-1 plan (logits stored, streaming softmax).  Same loss, logsumexp, G."""
    Qb, Cb, y, m8 = _nl_problem(B, Nc, d, 11, dev)
    m8[y] = 0
    assert kn._lib.workspace_bytes(B, Nc, d) < B * Nc * 4, "the workspace of a no-logits shape holds no logit buffer"
    rl0, lse0, ls0, G0, S0 = kn.inbatch_fwd(Qb, Cb, y, 0, m8, 0.5, 1.0 / B, want_logits=True)
    rl1, lse1, ls1, G1, S1 = kn.inbatch_fwd(Qb, Cb, y, 0, m8, 0.5, 1.0 / B, want_logits=False)
    assert S1 is None
    assert ((lse1 - lse0).abs().max() / lse0.abs().max()).item() <= 1e-6
    assert ((rl1 - rl0).abs().max() / rl0.abs().max()).item() <= 1e-5
    assert abs(ls1.item() - ls0.item()) <= 1e-5 * abs(ls0.item())
    g0, g1 = G0.float(), G1.float()
    assert ((g1 - g0).abs() <= 2.0 ** -7 * g0.abs() + 1e-12).all()  # bf16: at most one ulp apart (exp2 vs exp formulation)
    assert torch.all(g1[:, m8.bool()] == 0) and g1.sum(dim=1).abs().max().item() <= 2e-2 / B
    # against fp32 torch on the stored logits of the other plan
    ref = torch.softmax(S0, dim=1)
    ref[torch.arange(B, device=dev), y] -= 1.0
    assert ((g1 - ref / B).abs().max() / (ref / B).abs().max()).item() <= 2.0 ** -8


@pytest.mark.parametrize("B,Nc,d", [(4096, 4096, 768), (1000, 16392, 128), (1024, 32768, 1024), (4100, 4360, 768)])
def test_one_pass_forward_equals_the_two_pass_forward(B, Nc, d, kn, dev):
    """Option nl_p16 (round 6): the no-logits forward with the dScores wanted as ONE pass of the GEMM (strip statistics + fp16 softmax
    numerators, then the row kernel that rescales them into G in place) against the two-pass forward (statistics GEMM,
    then the logits recomputed into G): row logsumexp, row loss and the loss sum BIT-IDENTICAL (same statistics, same order), G at most
    one bf16 ulp apart, masked columns exactly 0, and against fp32 torch on stored logits."""
    from dpr_scale_amd import _lib

    Qb, Cb, y, m8 = _nl_problem(B, Nc, d, 41, dev)
    m8[y[B // 2:]] = 0  # (the first half of the gold columns may stay masked: -inf gold logit, loss +inf, G[gold] = -scale)
    outs = {}
    try:
        for mode in (0, 1):
            _lib.set_option("nl_p16", mode)
            outs[mode] = kn.inbatch_fwd(Qb, Cb, y, 0, m8, 0.5, 1.0 / B, want_logits=False)
    finally:
        _lib.set_option("nl_p16", 1)
    rl0, lse0, ls0, G0, _ = outs[0]
    for mode in (1,):
        rl, lse, ls, G, _ = outs[mode]
        assert torch.equal(lse, lse0) and torch.equal(rl, rl0)
        assert torch.equal(ls, ls0) or (not math.isfinite(ls0.item()) and not math.isfinite(ls.item()))
        g0, g1 = G0.float(), G.float()
        assert ((g1 - g0).abs() <= 2.0 ** -7 * g0.abs() + 1e-12).all()
        cols = torch.nonzero(m8).flatten()
        keep = cols[~torch.isin(cols, y)]
        assert torch.all(g1[:, keep] == 0)
    rerun = kn.inbatch_fwd(Qb, Cb, y, 0, m8, 0.5, 1.0 / B, want_logits=False)
    assert torch.equal(outs[1][3], rerun[3]) and torch.equal(outs[1][1], rerun[1])  # (reruns are bit-identical)
    S = kn.sim(Qb, Cb, m8, 0.5)
    ref = torch.softmax(S, dim=1)
    fin = torch.isfinite(lse0)
    ref[torch.arange(B, device=dev), y] -= 1.0
    assert ((outs[1][3].float()[fin] - ref[fin] / B).abs().max() / (ref[fin] / B).abs().max()).item() <= 2.0 ** -8


@pytest.mark.parametrize("B,Nc,d,mode", [(1024, 4096, 768, 1), (2048, 4096, 256, 1), (512, 2048, 128, 1), (320, 4160, 768, 1), (256, 1088, 1024, 1),
                                          (512, 8192, 768, 2), (1024, 16448, 128, 2), (64, 8192, 72, 2),
                                          # context counts that are multiples of 8, not of 64 (the packed layout of 2 or 4 ranks): the last K step of
                                          # the last dQ slice is partial (8 .. 56 contexts deep)
                                          (512, 4128, 768, 1), (256, 2080, 768, 1), (512, 2064, 256, 1), (320, 4104, 128, 1), (1024, 4152, 768, 1),
                                          # query rows that are no multiple of 64: the last K step of the dC tiles is partial
                                          (160, 4096, 768, 1), (300, 8200, 128, 1), (1000, 4104, 256, 1), (100, 2048, 768, 1), (65, 2056, 64, 1)])
def test_backward_pair_on_the_lds_dma_tile(B, Nc, d, mode, kn, dev):
    """Option pair128 (round 6, gemm128d_pair_kernel): dC tiles next to split-K dQ tiles in ONE launch on the 128 x 128 LDS-DMA tile -- the
    plan of the shapes under the 256 x 256 gate from 256 rows on (mode 1), forced onto other shapes for A/B (mode 2: column counts of the
    packed layout, a vector width that is no multiple of 128, 64 rows).  dC (one K range, the same MFMA order) BIT-identical to the
    register-staged plan it replaces, dQ to fp32 rounding of a different slice count, both against fp32 torch (dpr_task.py:209-212 backward)."""
    from dpr_scale_amd import _lib

    gen = torch.Generator(device="cpu").manual_seed(B + Nc + d)
    G = (torch.randn(B, Nc, generator=gen) * 0.01).to(torch.bfloat16).to(dev)
    Qb = torch.randn(B, d, generator=gen).to(torch.bfloat16).to(dev)
    Cb = torch.randn(Nc, d, generator=gen).to(torch.bfloat16).to(dev)
    one = torch.full((1,), 0.5, device=dev)
    outs = {}
    try:
        for m in (0, mode):
            _lib.set_option("pair128", m)
            dQ, dC = kn.inbatch_bwd(G, Qb, Cb, 2.0, one)
            outs[m] = (dQ.clone(), dC.clone())
    finally:
        _lib.set_option("pair128", 1)
    ref_dq = G.float() @ Cb.float()
    ref_dc = G.float().t() @ Qb.float()
    if mode == 1:  # (against the register-staged engine: the same 16 x 16 x 32 chain per element; the 256 x 256 family chains 32 x 32 x 16)
        assert torch.equal(outs[0][1], outs[mode][1])
    assert ((outs[mode][1] - outs[0][1]).abs().max() / ref_dc.abs().max()).item() <= 1e-5
    assert ((outs[mode][0] - outs[0][0]).abs().max() / ref_dq.abs().max()).item() <= 1e-5
    assert ((outs[mode][0] - ref_dq).abs().max() / ref_dq.abs().max()).item() <= 1e-3
    assert ((outs[mode][1] - ref_dc).abs().max() / ref_dc.abs().max()).item() <= 1e-3


@pytest.mark.parametrize("B,Nc,d", [(256, 8192, 768), (512, 8192, 768), (1024, 8192, 768), (300, 8200, 128), (520, 4104, 256), (136, 16392, 64),
                                    (1024, 1024, 768), (256, 2048, 128), (2048, 8192, 192)])  # (the last: K = 3 steps of 64 -- the 256 x 256 forward wants multiples of 128 -- at 1024 tiles)
def test_one_pass_forward_on_the_128_tile(B, Nc, d, kn, dev):
    """Option nl128 (round 6): the shapes whose 256-wide tiles cannot fill the chip (a few hundred query rows against thousands of
    contexts) run the training forward in ONE pass on the 128 x 128 LDS-DMA tile -- 64-column strip statistics + fp16 softmax numerators
    (EpiSimP), then the row kernel that rescales them into G in place -- instead of storing fp32 logits for a streaming softmax
    (dpr_task.py:209-212).  Against that logits-storing plan on the same operands: loss / logsumexp to fp32 rounding, G at most one bf16
    ulp apart, masked columns exactly 0 (gold columns excepted), rows summing to ~0; against fp32 torch on the stored logits; reruns
    bit-identical.  Half of the gold columns may be masked (loss +inf, G[gold] = -scale)."""
    from dpr_scale_amd import _lib

    assert _lib.fwd_one_pass(B, Nc, d) == 2
    Qb, Cb, y, m8 = _nl_problem(B, Nc, d, 29, dev)
    m8[y[B // 2:]] = 0
    try:
        _lib.set_option("nl128", 0)
        _lib.set_option("nl_min", 1 << 30)  # (the reference leg stores its logits at every shape here)
        assert _lib.fwd_one_pass(B, Nc, d) == 0
        rl0, lse0, ls0, G0, S0 = kn.inbatch_fwd(Qb, Cb, y, 0, m8, 0.5, 1.0 / B, want_logits=True)
    finally:
        _lib.set_option("nl128", 1)
        _lib.set_option("nl_min", 128)
    rl1, lse1, ls1, G1, S1 = kn.inbatch_fwd(Qb, Cb, y, 0, m8, 0.5, 1.0 / B, want_logits=False)
    assert S1 is None
    fin = torch.isfinite(rl0)
    assert torch.equal(torch.isfinite(rl1), fin) and fin.sum().item() >= B // 2
    assert ((lse1 - lse0).abs().max() / lse0.abs().max()).item() <= 1e-6
    assert ((rl1[fin] - rl0[fin]).abs().max() / rl0[fin].abs().max()).item() <= 1e-5
    assert (not math.isfinite(ls0.item()) and not math.isfinite(ls1.item())) or abs(ls1.item() - ls0.item()) <= 1e-5 * abs(ls0.item())
    g0, g1 = G0.float(), G1.float()
    assert ((g1 - g0).abs() <= 2.0 ** -7 * g0.abs() + 1e-12).all()
    cols = torch.nonzero(m8).flatten()
    keep = cols[~torch.isin(cols, y)]
    assert torch.all(g1[:, keep] == 0)
    ref = torch.softmax(S0, dim=1)
    ref[torch.arange(B, device=dev), y] -= 1.0
    assert ((g1[fin] - ref[fin] / B).abs().max() / (ref[fin] / B).abs().max()).item() <= 2.0 ** -8
    rerun = kn.inbatch_fwd(Qb, Cb, y, 0, m8, 0.5, 1.0 / B, want_logits=False)
    assert torch.equal(G1, rerun[3]) and torch.equal(lse1, rerun[1]) and torch.equal(rl1, rerun[0])
    # no column mask at all (colmask == NULL), against fp32 torch
    rl2, lse2, ls2, G2, _ = kn.inbatch_fwd(Qb, Cb, y, 0, None, 0.5, 1.0 / B, want_logits=False)
    S2 = (Qb.float() @ Cb.float().t()) * 0.5
    ref_lse = torch.logsumexp(S2, dim=1)
    assert ((lse2 - ref_lse).abs().max() / ref_lse.abs().max()).item() <= 1e-5
    ref2 = torch.softmax(S2, dim=1)
    ref2[torch.arange(B, device=dev), y] -= 1.0
    assert ((G2.float() - ref2 / B).abs().max() / (ref2 / B).abs().max()).item() <= 2.0 ** -8
    assert abs(ls2.item() - (ref_lse - S2[torch.arange(B, device=dev), y]).sum().item()) <= 1e-4 * abs(ls2.item())


@pytest.mark.parametrize("B,Nc,d", [(1024, 8192, 768), (512, 16384, 256), (2048, 4104, 128), (4096, 4096, 256), (256, 8192, 128)])
def test_step_forms_its_loss_in_the_slab_combining_launch(B, Nc, d, kn, dev):
    """Option loss_with_dq (round 6): in the one-call step on the no-logits forward the sum of the row losses is formed by one more
    workgroup of the launch that combines the dQ slabs, not by a launch of its own between forward and backward.  Same arithmetic:
    loss, row losses, dQ and dC BIT-IDENTICAL with the option off; the loss against fp32 torch on the same operands."""
    from dpr_scale_amd import _lib

    assert _lib.fwd_one_pass(B, Nc, d) > 0  # (both tile families: 4096 x 4096 is the 256 x 256 one)
    gen = torch.Generator(device="cpu").manual_seed(77)
    q = (torch.randn(B, d, generator=gen) * d ** -0.25).to(torch.bfloat16).float().to(dev)
    c = (torch.randn(Nc, d, generator=gen) * d ** -0.25).to(torch.bfloat16).float().to(dev)
    y = torch.randperm(Nc, generator=gen)[:B].to(dev)
    m8 = (torch.rand(Nc, generator=gen) < 0.01).to(torch.uint8).to(dev)
    m8[y] = 0
    one = torch.ones(1, device=dev)
    outs = {}
    try:
        for mode in (0, 1):
            _lib.set_option("loss_with_dq", mode)
            Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
            Cb = torch.empty((Nc, d), dtype=torch.bfloat16, device=dev)
            outs[mode] = kn.train_step_f32(q, c, Qb, Cb, y, 0, m8, 0.5, 0.5 / B, 1.0 / B, one)
            torch.cuda.synchronize()
    finally:
        _lib.set_option("loss_with_dq", 1)
    rl0, lse0, lo0, _, dQ0, dC0 = outs[0]
    rl1, lse1, lo1, _, dQ1, dC1 = outs[1]
    assert torch.equal(lo0[:1], lo1[:1]) and torch.equal(rl0, rl1) and torch.equal(lse0, lse1)
    assert torch.equal(dQ0, dQ1) and torch.equal(dC0, dC1)
    rq, rc = q.clone().requires_grad_(True), c.clone().requires_grad_(True)
    S = (rq @ rc.T).masked_fill(m8.bool()[None, :], float("-inf")) * 0.5
    ref = torch.nn.functional.cross_entropy(S, y)
    ref.backward()  # (grad_scale = inv_T / B: the mean loss's gradient, as the operator asks for it)
    assert abs(lo1[0].item() - ref.item()) <= LOSS_RTOL * abs(ref.item())
    assert ((dQ1 - rq.grad).abs().max() / rq.grad.abs().max()).item() <= GRAD_RTOL
    assert ((dC1 - rc.grad).abs().max() / rc.grad.abs().max()).item() <= GRAD_RTOL


@pytest.mark.parametrize("B,Nc,d", [(4096, 4096, 768), (4100, 4360, 768), (1024, 32768, 1024), (1000, 16392, 128)])
def test_one_pass_forward_numerators_through_the_lds_patch(B, Nc, d, kn, dev):
    """Option p16_staged: the fp16 numerators of the 256 x 256 one-pass forward leave through the per-wave LDS patch (64-byte pieces of 16
    rows per store instruction) instead of straight from the registers (32-byte pieces of 32 rows).  Same values, another way out: G,
    logsumexp, row losses and the loss sum BIT-identical, ragged rows / columns included."""
    from dpr_scale_amd import _lib

    assert _lib.fwd_one_pass(B, Nc, d) == 1
    Qb, Cb, y, m8 = _nl_problem(B, Nc, d, 53, dev)
    outs = {}
    default = _lib.get_option("p16_staged")
    try:
        for mode in (0, 2):
            _lib.set_option("p16_staged", mode)
            outs[mode // 2] = kn.inbatch_fwd(Qb, Cb, y, 0, m8, 0.5, 1.0 / B, want_logits=False)
            torch.cuda.synchronize()
    finally:
        _lib.set_option("p16_staged", default)
    for k in (0, 1, 3):
        assert torch.equal(outs[0][k], outs[1][k])
    assert torch.equal(outs[0][2], outs[1][2]) or (not math.isfinite(outs[0][2].item()) and not math.isfinite(outs[1][2].item()))


def test_no_logits_forward_from_fp32_inputs_and_autograd(dev):
    """The fp32 entry point and the autograd operator at a no-logits shape against fp32 torch (reference formulation,
    dpr_task.py:197-212)."""
    from dpr_scale_amd.hotpath import inbatch_contrastive_loss

    B, K, d = 2048, 16, 128  # Nc = 32768: 8 x 128 tiles
    gen = torch.Generator(device="cpu").manual_seed(5)
    q = (torch.randn(B, d, generator=gen) * d ** -0.25).to(torch.bfloat16).float().to(dev)
    c = (torch.randn(B * K, d, generator=gen) * d ** -0.25).to(torch.bfloat16).float().to(dev)
    y = (torch.arange(B) * K).to(dev)
    mask = (torch.rand(B * K, generator=gen) < 0.02).to(dev)
    mask[y] = False
    tq, tc = q.clone().requires_grad_(True), c.clone().requires_grad_(True)
    loss = inbatch_contrastive_loss(tq, tc, y, mask, 0.7)
    loss.backward()
    rq, rc = q.clone().requires_grad_(True), c.clone().requires_grad_(True)
    S = (rq @ rc.T).masked_fill(mask[None, :], float("-inf")) / 0.7
    ref = torch.nn.functional.cross_entropy(S, y)
    ref.backward()
    assert abs(loss.item() - ref.item()) <= LOSS_RTOL * abs(ref.item())
    assert ((tq.grad - rq.grad).abs().max() / rq.grad.abs().max()).item() <= GRAD_RTOL
    assert ((tc.grad - rc.grad).abs().max() / rc.grad.abs().max()).item() <= GRAD_RTOL


def test_no_logits_step_at_bert_base_width_against_the_oracle(kn, dev):
    """A 2048 x 8192 x 768 world (exactly 256 tiles of 256 x 256: the no-logits forward -- statistics GEMM, logsumexp, dScores GEMM -- and
    the 256 x 256 backward pair, twelve K steps per tile) against the numpy restatement of dpr_task.py:153-214 in fp64: loss and every
    row's logsumexp to 1e-3 (measured ~1e-6), dQ / dC to the gradient bar 1e-2 of max |grad| (bf16 dScores)."""
    from dpr_scale_amd.hotpath import inbatch_contrastive_loss

    B, K, d, T = 2048, 4, 768, 0.5
    q, c, y, m = O.synth_embeddings(77, B, K, d, "U", True)
    r = O.training_step_global(q, c, y, m, T)
    assert kn._lib.workspace_bytes(B, B * K, d) < B * B * K * 4, "2048 x 8192 x 768 must be a no-logits shape"
    tq, tc = t(q, dev).requires_grad_(True), t(c, dev).requires_grad_(True)
    loss = inbatch_contrastive_loss(tq, tc, t(y, dev), t(m, dev), T)
    loss.backward()
    assert abs(loss.item() - r["loss"]) <= LOSS_RTOL * max(1.0, abs(r["loss"]))
    assert rel(tq.grad.cpu().numpy(), r["dQ"]) <= GRAD_RTOL and rel(tc.grad.cpu().numpy(), r["dC"]) <= GRAD_RTOL
    # the C-ABI forward on the bf16 operands: row logsumexp, row loss, dScores
    rl, lse, ls, G, S = kn.inbatch_fwd(bf16(q, dev), bf16(c, dev), t(y, dev), 0, t(m.astype(np.uint8), dev), 1.0 / T, 1.0 / (B * T), want_logits=False)
    assert S is None
    assert rel(lse.cpu().numpy(), r["lse"]) <= LOSS_RTOL and rel(rl.cpu().numpy(), r["row_loss"]) <= LOSS_RTOL
    assert abs(ls.item() / B - r["loss"]) <= LOSS_RTOL * max(1.0, abs(r["loss"]))
    assert rel(G.float().cpu().numpy(), r["G"]) <= 2.0 ** -7  # bf16 dScores: half an ulp of the largest element + the fp32 logit error


@pytest.mark.parametrize("B,Nc,d", NL_SHAPES + [(64, 1000, 64)])
def test_score_free_rank_is_bit_exact(B, Nc, d, kn, dev):
    """dprhot_sim_rank (gold logits from the gathered mini GEMM + count-greater in the GEMM epilogue; no score matrix at large
    shapes) == rank_of_gold of the stored scores == position in the reference's stable descending sort, ties included."""
    Nc8 = (Nc + 7) // 8 * 8
    Qb, Cb, y, m8 = _nl_problem(B, Nc8, d, 23, dev, dup=True)
    S = kn.sim(Qb, Cb, m8, 1.0)
    want = kn.rank_of_gold(S, y)
    got = kn.sim_rank(Qb, Cb, y, m8, 1.0)
    assert torch.equal(got, want)
    sub = torch.arange(0, B, max(1, B // 64), device=dev)
    order = torch.sort(S[sub], dim=1, descending=True, stable=True).indices
    ref_rank = (order == y[sub][:, None]).nonzero()[:, 1] + 1
    assert torch.equal(got[sub], ref_rank)


@pytest.mark.parametrize("B,Nc,d", NL_SHAPES + [(64, 1000, 64)])
def test_one_pass_rank_and_loss_equals_the_two_passes(B, Nc, d, kn, dev):
    """dprhot_sim_rank_loss (validation: ranks AND cross-entropy from ONE pass of the similarity GEMM at no-logits shapes -- count and
    softmax statistics in the same epilogue) against dprhot_sim_rank + dprhot_inbatch_fwd: ranks bit-exact (ties included), loss to
    1e-6; and against the stored score matrix."""
    Nc8 = (Nc + 7) // 8 * 8
    Qb, Cb, y, m8 = _nl_problem(B, Nc8, d, 29, dev, dup=True)
    rank1, loss1 = kn.sim_rank_loss(Qb, Cb, y, m8, 1.0)
    rank2 = kn.sim_rank(Qb, Cb, y, m8, 1.0)
    _, _, loss2, _, _ = kn.inbatch_fwd(Qb, Cb, y, 0, m8, 1.0, 1.0, want_logits=False, want_G=False)
    assert torch.equal(rank1, rank2)
    l1, l2 = loss1.item(), loss2.item()  # (a masked gold column makes the sum +inf in both: equal, not close)
    assert l1 == l2 or abs(l1 - l2) <= 1e-6 * max(1.0, abs(l2))
    S = kn.sim(Qb, Cb, m8, 1.0)
    assert torch.equal(rank1, kn.rank_of_gold(S, y))
    ref = torch.nn.functional.cross_entropy(S, y, reduction="sum").item()
    assert l1 == ref or abs(l1 - ref) <= 1e-4 * max(1.0, abs(ref))
    # and on a problem whose loss is finite
    Qb, Cb, y, m8 = _nl_problem(B, Nc8, d, 31, dev)
    m8[y] = 0
    rank1, loss1 = kn.sim_rank_loss(Qb, Cb, y, m8, 1.0)
    _, _, loss2, _, _ = kn.inbatch_fwd(Qb, Cb, y, 0, m8, 1.0, 1.0, want_logits=False, want_G=False)
    assert torch.equal(rank1, kn.sim_rank(Qb, Cb, y, m8, 1.0))
    assert math.isfinite(loss2.item()) and abs(loss1.item() - loss2.item()) <= 1e-6 * max(1.0, abs(loss2.item()))


def test_rank_and_loss_helper_matches_score_matrix_path(kn, dev):
    from dpr_scale_amd import hotpath

    B, Nc, d = 1024, 16390, 128  # ragged column count: padded with masked columns inside
    gen = torch.Generator(device="cpu").manual_seed(3)
    q = (torch.randn(B, d, generator=gen) * d ** -0.25).to(torch.bfloat16).float().to(dev)
    c = (torch.randn(Nc, d, generator=gen) * d ** -0.25).to(torch.bfloat16).float().to(dev)
    y = torch.randint(0, Nc, (B,), generator=gen).to(dev)
    mask = (torch.rand(Nc, generator=gen) < 0.05).to(dev)
    mask[y] = False
    ranks, loss = hotpath.rank_and_loss(q, c, y, mask, 1.0, kn)
    S = hotpath.sim_score(q, c, mask, 1.0, kn)
    assert torch.equal(ranks, hotpath.rank_of_gold(S, y, kn))
    ref = torch.nn.functional.cross_entropy(S, y)
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())


def test_operator_applies_any_grad_output_with_one_fixup_launch(dev):
    """Under AMP backward() receives the loss scale, not 1.  The operator computes its gradients in the forward call for the
    grad_output the PREVIOUS backward saw and fixes them up in backward (dprhot_rescale_grads) only when the scale changed: every
    sequence of scales -- first step, unchanged, changed, zero, a second backward through a retained graph -- must give
    grad_output x the reference's gradients."""
    from dpr_scale_amd import hotpath

    meta, g = load_golden("cfg2_Ur_T0.05")
    q, c, y, m = rank_inputs(meta)[0]
    ty, tm = t(y, dev), t(m, dev)

    def step(scale, retain=False):
        tq, tc = t(q, dev).requires_grad_(True), t(c, dev).requires_grad_(True)
        loss = hotpath.inbatch_contrastive_loss(tq, tc, ty, tm, meta["T"])
        (loss * scale).backward(retain_graph=retain)
        return loss, tq, tc

    for scale in (1.0, 65536.0, 65536.0, 32768.0, 0.0, 3.0, 1.0):
        loss, tq, tc = step(scale)
        assert abs(loss.item() - g["loss"]) <= LOSS_RTOL * max(1.0, abs(g["loss"]))
        if scale == 0.0:
            assert float(tq.grad.abs().max()) == 0.0 and float(tc.grad.abs().max()) == 0.0
        else:
            assert rel(tq.grad.cpu().numpy() / scale, g["dQ"]) <= GRAD_RTOL, scale
            assert rel(tc.grad.cpu().numpy() / scale, g["dC"]) <= GRAD_RTOL, scale
    # the expectation never becomes a value gradients cannot be rescaled from
    exp = hotpath._ExpectedGradScale.get(dev)
    assert float(exp.item()) == 1.0
    # two backwards through one graph with different scales: gradients accumulate as 2 x + 5 x
    tq, tc = t(q, dev).requires_grad_(True), t(c, dev).requires_grad_(True)
    loss = hotpath.inbatch_contrastive_loss(tq, tc, ty, tm, meta["T"])
    (loss * 2.0).backward(retain_graph=True)
    (loss * 5.0).backward()
    assert rel(tq.grad.cpu().numpy() / 7.0, g["dQ"]) <= GRAD_RTOL and rel(tc.grad.cpu().numpy() / 7.0, g["dC"]) <= GRAD_RTOL
    # an overflowed loss scale (inf) propagates like the reference's autograd does (non-finite gradients), and is not remembered
    loss, tq, tc = step(float("inf"))
    assert not torch.isfinite(tq.grad).all()
    assert float(hotpath._ExpectedGradScale.get(dev).item()) == 1.0
    loss, tq, tc = step(4.0)
    assert rel(tq.grad.cpu().numpy() / 4.0, g["dQ"]) <= GRAD_RTOL


def test_operator_in_an_amp_loop_with_a_moving_loss_scale(dev):
    """torch.amp.GradScaler multiplies the loss by a scale that GROWS every `growth_interval` good steps and is halved on overflow; the
    operator computes its gradients for the scale the previous backward saw.  Ten optimiser steps of two tiny towers with
    growth_interval = 2 (the scale changes every other step) must follow the same trajectory as the reference's formulation in
    torch ops driven by an identical scaler."""
    from dpr_scale_amd import hotpath

    B, K, d_in, d = 32, 4, 48, 64
    gen = torch.Generator(device="cpu").manual_seed(21)
    xq = torch.randn(10, B, d_in, generator=gen).to(dev)
    xc = torch.randn(10, B * K, d_in, generator=gen).to(dev)
    pos = (torch.arange(B) * K).to(dev)
    mask = torch.zeros(B * K, dtype=torch.bool, device=dev)
    mask[1::7] = True
    mask[pos] = False

    def run(use_op):
        torch.manual_seed(5)
        tq, tc = torch.nn.Linear(d_in, d).to(dev), torch.nn.Linear(d_in, d).to(dev)
        opt = torch.optim.SGD(list(tq.parameters()) + list(tc.parameters()), lr=0.05)
        scaler = torch.amp.GradScaler("cuda", init_scale=256.0, growth_interval=2)
        losses = []
        for s in range(10):
            q, c = tq(xq[s]), tc(xc[s])
            q, c = q.to(torch.bfloat16).float(), c.to(torch.bfloat16).float()  # both formulations see bf16-representable embeddings
            if use_op:
                loss = hotpath.inbatch_contrastive_loss(q, c, pos, mask, 1.0, False)
            else:
                scores = q @ c.t()
                scores = scores.masked_fill(mask[None, :], float("-inf"))
                loss = torch.nn.functional.cross_entropy(scores, pos)
            opt.zero_grad(set_to_none=True)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            losses.append(loss.item())
        return losses, torch.cat([p.detach().flatten() for p in list(tq.parameters()) + list(tc.parameters())]), scaler.get_scale()

    l_ref, p_ref, s_ref = run(False)
    l_op, p_op, s_op = run(True)
    assert s_ref == s_op and s_ref > 256.0  # the scale moved, identically
    assert max(abs(a - b) for a, b in zip(l_op, l_ref)) <= 2e-3 * max(1.0, max(l_ref))
    assert (p_op - p_ref).abs().max().item() <= 2e-2 * p_ref.abs().max().item()  # bf16 dScores over ten steps


def test_operator_step_launches_only_library_kernels(dev):
    """SURVEY.md section 8 b2 / VERDICT round 2: one training step through the autograd operator (forward + backward with a loss
    scale, as under AMP) must not add torch elementwise passes to the hand-written step."""
    from torch.profiler import ProfilerActivity, profile

    from dpr_scale_amd import hotpath

    meta, g = load_golden("cfg2_U_T1")
    q, c, y, m = rank_inputs(meta)[0]
    tq, tc, ty, tm = t(q, dev).requires_grad_(True), t(c, dev).requires_grad_(True), t(y, dev), t(m, dev)
    scale = torch.full((), 1024.0, device=dev)

    def one():
        tq.grad = tc.grad = None
        loss = hotpath.inbatch_contrastive_loss(tq, tc, ty, tm, 1.0)
        loss.backward(scale)

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        one()
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and "Memcpy" not in e.name and "Memset" not in e.name]
    if not names:
        pytest.skip("the profiler recorded no device activity on this box")
    foreign = [n for n in names if "dprhot" not in n]
    assert not foreign, foreign
    assert len(names) <= 3, names  # sim, softmax + backward, grad_output check


def test_rank_and_loss_with_a_hidden_size_that_is_not_a_multiple_of_8(kn, dev):
    """d = 100 (tiny encoders, projection heads): validation must pad like training does (it used to hit `n % 8 == 0`)."""
    from dpr_scale_amd import hotpath

    B, Nc, d = 24, 50, 100
    gen = torch.Generator(device="cpu").manual_seed(11)
    q = (torch.randn(B, d, generator=gen) * d ** -0.25).to(torch.bfloat16).float()
    c = (torch.randn(Nc, d, generator=gen) * d ** -0.25).to(torch.bfloat16).float()
    y = torch.randint(0, Nc, (B,), generator=gen)
    ranks, loss = hotpath.rank_and_loss(q.to(dev), c.to(dev), y.to(dev), None, 1.0, kn)
    S = O.sim_score(q.numpy(), c.numpy(), None)
    assert np.array_equal(ranks.cpu().numpy(), O.rank_of_gold(S.astype(np.float32), y.numpy()))
    ref = O.log_softmax_ce(S, y.numpy())[0]
    assert abs(loss.item() - ref) <= LOSS_RTOL * max(1.0, abs(ref))


@pytest.mark.parametrize("W,B,K,d", [(2, 1024, 16, 128), (2, 512, 32, 256), (8, 1024, 8, 256), (2, 1024, 16, 768), (4, 512, 16, 1024),
                                     (8, 256, 8, 768), (4, 384, 8, 256)])  # (the last two: the one-pass forward on the 128 x 128 tile, mask bytes and padding rows read from the packed buffer)
def test_packed_multi_rank_step_at_a_no_logits_shape(W, B, K, d, kn, dev):
    """Large per-rank batch over several ranks, emulated on one GPU: dprhot_inbatch_step_packed_f32 then runs the no-logits forward
    with the column mask read from the gathered buffer's mask rows (Epi8Base::mask_byte, packed layout) and stamps the loss numerator
    into dC_part.  Checked against the unpacked formulation of the same step (mask vector from dprhot_unpack_mask, logits stored)."""
    # (2, 1024, 16, 128): Nc = 2 * packed_rows(16384, 128) ~ 33 k columns, 4 x 129 tiles.  The other two are long-context-axis shapes of
    # the 256 x 256 backward (Nc >= 32 B): dC -- and with it the loss stamp -- leaves through the 128 x 128 engine in a launch of its own
    n_ctx = B * K
    rows_c = kn.packed_rows(n_ctx, d)
    gen = torch.Generator(device="cpu").manual_seed(9)
    qs = [(torch.randn(B, d, generator=gen) * d ** -0.25).to(torch.bfloat16).float().to(dev) for _ in range(W)]
    cs = [(torch.randn(n_ctx, d, generator=gen) * d ** -0.25).to(torch.bfloat16).float().to(dev) for _ in range(W)]
    ms = [(torch.rand(n_ctx, generator=gen) < 0.03) for _ in range(W)]
    y = (torch.arange(B) * K)
    for m in ms:
        m[y] = False
    sends = []
    for r in range(W):
        send = torch.empty((rows_c, d), dtype=torch.bfloat16, device=dev)
        kn.pack_ctx(cs[r], ms[r].to(torch.uint8).to(dev), send)
        sends.append(send)
    Cb = torch.cat(sends, 0).contiguous()
    assert kn._lib.fwd_one_pass(B, W * rows_c, d) == (2 if B < 512 else 1)  # no logits stored in the fused forward; B >= 512 here: not in the workspace either
    if B >= 512:
        assert kn._lib.workspace_bytes(B, W * rows_c, d) < B * W * rows_c * 4
    colmask = torch.empty(W * rows_c, dtype=torch.uint8, device=dev)
    kn.unpack_mask(Cb, W, n_ctx, colmask)
    yd = y.to(dev)
    Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
    one = torch.ones(1, dtype=torch.float32, device=dev)
    for r in range(W):
        _, _, ls, _, dq, dcp = kn.inbatch_step_packed_f32(qs[r], Cb, Qb, W, r, n_ctx, yd, 1.0, 1.0 / (W * B))
        rl, lse, ls_ref, G, S = kn.inbatch_fwd_f32(qs[r], None, Qb, Cb, yd, r * rows_c, colmask, 1.0, 1.0 / (W * B), want_logits=True)
        dq_ref, dc_ref = kn.inbatch_bwd(G, Qb, Cb, 1.0, one)
        assert abs(ls.item() - ls_ref.item()) <= 1e-5 * abs(ls_ref.item())
        assert ((dq - dq_ref).abs().max() / dq_ref.abs().max()).item() <= 2e-3
        stamped = dcp.clone()
        for k in range(W):
            assert abs(stamped[k * rows_c + n_ctx, 0].item() - ls.item()) <= 1e-6 * abs(ls.item())  # the loss numerator rides here
            stamped[k * rows_c + n_ctx, 0] = 0.0
        assert ((stamped - dc_ref).abs().max() / dc_ref.abs().max()).item() <= 2e-3
        assert torch.all(stamped.view(W, rows_c, d)[:, n_ctx:] == 0)  # mask rows: masked columns have G == 0 exactly

@pytest.mark.parametrize("W,B,K,d", [(8, 32, 8, 768), (4, 64, 16, 256), (8, 32, 9, 768), (8, 128, 8, 128)])
def test_packed_skinny_step_tiles_over_real_rows(W, B, K, d, kn, dev):
    """Few rows x thousands of gathered contexts (skinny.h) in the packed multi-rank layout: when a rank's n_ctx is a whole number
    of sim tiles the tiles cover the real rows only and the header columns of the logits are written as -inf (cfg2 over 8 ranks:
    64-column tiles; 4 x 64 x 16: 128-column tiles; K = 9: no whole number of tiles -> tiles over all packed rows).  Checked against
    the unpacked formulation of the same step (explicit mask vector, label offset)."""
    n_ctx = B * K
    rows_c = kn.packed_rows(n_ctx, d)
    gen = torch.Generator(device="cpu").manual_seed(W * 1000 + K)
    qs = [(torch.randn(B, d, generator=gen) * d ** -0.25).to(torch.bfloat16).float().to(dev) for _ in range(W)]
    cs = [(torch.randn(n_ctx, d, generator=gen) * d ** -0.25).to(torch.bfloat16).float().to(dev) for _ in range(W)]
    ms = [(torch.rand(n_ctx, generator=gen) < 0.05) for _ in range(W)]
    y = torch.arange(B) * K
    for m in ms:
        m[y] = False
    sends = []
    for r in range(W):
        send = torch.empty((rows_c, d), dtype=torch.bfloat16, device=dev)
        kn.pack_ctx(cs[r], ms[r].to(torch.uint8).to(dev), send)
        sends.append(send)
    Cb = torch.cat(sends, 0).contiguous()
    colmask = torch.empty(W * rows_c, dtype=torch.uint8, device=dev)
    kn.unpack_mask(Cb, W, n_ctx, colmask)
    yd = y.to(dev)
    Qb = torch.empty((B, d), dtype=torch.bfloat16, device=dev)
    one = torch.ones(1, dtype=torch.float32, device=dev)
    for r in (0, W - 1):
        rl, lse_p, ls, _, dq, dcp = kn.inbatch_step_packed_f32(qs[r], Cb, Qb, W, r, n_ctx, yd, 1.0, 1.0 / (W * B))
        rl_ref, lse_ref, ls_ref, G, S = kn.inbatch_fwd_f32(qs[r], None, Qb, Cb, yd, r * rows_c, colmask, 1.0, 1.0 / (W * B), want_logits=True)
        dq_ref, dc_ref = kn.inbatch_bwd(G, Qb, Cb, 1.0, one)
        assert abs(ls.item() - ls_ref.item()) <= 1e-5 * abs(ls_ref.item())
        assert torch.allclose(lse_p, lse_ref, rtol=1e-5, atol=1e-5) and torch.allclose(rl, rl_ref, rtol=1e-4, atol=1e-5)
        assert ((dq - dq_ref).abs().max() / dq_ref.abs().max()).item() <= 2e-3
        stamped = dcp.clone()
        for k in range(W):
            stamped[k * rows_c + n_ctx, 0] = 0.0
        assert ((stamped - dc_ref).abs().max() / dc_ref.abs().max()).item() <= 2e-3
        assert torch.all(stamped.view(W, rows_c, d)[:, n_ctx:] == 0)  # header rows: their columns of G are exactly 0


@pytest.mark.gpu
def test_cpp_autograd_node_equals_the_python_operator(dev):
    """hotpath.inbatch_contrastive_loss runs the plain single-rank fp32 step as a C++ autograd node (csrc/opx.cpp: InBatchFn); the
    Python operator (InBatchContrastive, every other case) makes the same two library calls: same bits, for a unit and a non-unit
    grad_output, and for a second backward through a retained graph."""
    from dpr_scale_amd import hotpath

    if not getattr(hotpath, "_OPX_NODE", False):
        pytest.skip("the optional host extension _opx.so is not built")
    meta, g = load_golden("cfg2_Ur_T0.05")
    q, c, y, m = rank_inputs(meta)[0]
    ty, tm = t(y, dev), t(m, dev)

    def run(fn, scale, twice=False):
        tq, tc = t(q, dev).requires_grad_(True), t(c, dev).requires_grad_(True)
        loss = fn(tq, tc)
        assert loss.grad_fn is not None
        (loss * scale).backward(retain_graph=twice)
        if twice:
            (loss * 3.0).backward()
        return loss.detach().clone(), tq.grad.clone(), tc.grad.clone()

    cpp = lambda tq, tc: hotpath.inbatch_contrastive_loss(tq, tc, ty, tm, meta["T"])  # noqa: E731
    py = lambda tq, tc: hotpath.InBatchContrastive.apply(tq, tc, ty, tm, meta["T"], None, None, None, None)  # noqa: E731
    for scale, twice in ((1.0, False), (1024.0, False), (1024.0, False), (2.0, True), (1.0, False)):
        a, b = run(cpp, scale, twice), run(py, scale, twice)
        assert "InBatchFn" in cpp(t(q, dev).requires_grad_(True), t(c, dev)).grad_fn.name()
        for x, z in zip(a, b):
            assert torch.equal(x, z), (scale, twice)
    # no gradient wanted: the node is not taken (forward only, the general path)
    with torch.no_grad():
        loss = hotpath.inbatch_contrastive_loss(t(q, dev), t(c, dev), ty, tm, meta["T"])
    assert loss.grad_fn is None and abs(loss.item() - g["loss"]) <= LOSS_RTOL * max(1.0, abs(g["loss"]))


@pytest.mark.gpu
@pytest.mark.parametrize("B,Nc,d,engine128,k256,pair128", [(1024, 49152, 768, True, True, False), (1024, 65536, 768, False, True, False),
                                                             (512, 16384, 768, False, False, True), (256, 57344, 768, False, False, True),
                                                             (2048, 65536, 768, False, True, False), (1024, 16384, 768, False, False, True),
                                                             (2048, 32768, 768, False, True, False), (128, 32768, 768, False, False, True),
                                                             # units with an ODD number of 64-deep K steps on the 256 x 256 kernel (packed-layout column
                                                             # counts are multiples of 64, rarely of 128): dC K = 1088 = 17 steps, dQ slices of odd length; 1025 steps
                                                             (1088, 32832, 768, False, True, False), (1024, 65600, 768, True, True, False),
                                                             (320, 8256, 768, False, False, True)])
def test_backward_over_a_very_long_context_axis(B, Nc, d, engine128, k256, pair128, kn, dev):
    """dprhot_inbatch_bwd's plan rules (round 5; re-measured in round 6 with the LDS-DMA 128 x 128 tile and its pair launch,
    profiles/r06_bwd_plan_ab.txt, r06_dc_alone_ab.txt, r06_pair128_ab.txt):
      B < 1024, or B x Nc <= 2^25     the pair launch on the 128 x 128 LDS-DMA tile (gemm128d_pair_kernel), long context axes included
      1024 <= B <= 2048, Nc >= 32 B   dC on the 128 x 128 tile in a launch of its own, dQ on the 256 x 256 kernel's units -- except where G's
                                      row pitch is a multiple of 128 KiB (Nc = 65536): there the dC tiles run ALONE on the 256 x 256 kernel
      otherwise                       the 256 x 256 pair launch
    Both gradients against fp32 matmuls of the same bf16 operands, and the launches that run are the ones the rule names.  Autograd of
    dpr_task.py:98-105 into q and c."""
    from torch.profiler import ProfilerActivity, profile

    gen = torch.Generator().manual_seed(B + Nc)
    G = (torch.randn(B, Nc, generator=gen) * 0.01).to(torch.bfloat16).to(dev)
    Qb = torch.randn(B, d, generator=gen).to(torch.bfloat16).to(dev)
    Cb = torch.randn(Nc, d, generator=gen).to(torch.bfloat16).to(dev)
    go = torch.full((1,), 0.5, device=dev)
    dQ, dC = kn.inbatch_bwd(G, Qb, Cb, 2.0, go)
    torch.cuda.synchronize()
    ref_dq = G.float() @ Cb.float()
    ref_dc = G.float().t() @ Qb.float()
    assert float((dQ - ref_dq).abs().max() / ref_dq.abs().max()) <= 2e-5
    assert float((dC - ref_dc).abs().max() / ref_dc.abs().max()) <= 2e-5
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        kn.inbatch_bwd(G, Qb, Cb, 2.0, go)
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    if names:
        assert any("gemm_bf16_kernel" in n or "gemm128d_kernel" in n for n in names) == engine128, names  # (the 128 x 128 tile, either staging)
        assert any("gemm8p_bwd_kernel" in n for n in names) == k256, names
        assert any("gemm128d_pair_kernel" in n for n in names) == pair128, names


@pytest.mark.gpu
@pytest.mark.parametrize("B,Nc,d", [(512, 16384, 768), (320, 8200, 512), (1024, 32768, 1024), (192, 4096, 128), (300, 4104, 256)])  # (the last: partial last K steps in dQ and dC)
def test_lds_dma_staged_128_tile_is_bit_identical(B, Nc, d, kn, dev):
    """Option g128_dma (round 6, csrc/gemm128d.h): the 128 x 128 x 64 tile of the GEMM engine with its operands staged by LDS-DMA instead
    of global -> VGPR -> ds_write.  Same LDS images, same fragments, same MFMA order, same epilogues: dQ = G x C (A k-major, B mn-major),
    dC = G^T x Q (both mn-major) and the stored-logits similarity GEMM (both k-major, statistics epilogue) must be BIT-identical to the
    register-staged kernel's, ragged edges included; and equal to fp32 torch within the gradient bar."""
    from dpr_scale_amd import _lib

    gen = torch.Generator(device="cpu").manual_seed(B + d)
    G = (torch.randn(B, Nc, generator=gen) * 0.01).to(torch.bfloat16).to(dev)
    Qb = torch.randn(B, d, generator=gen).to(torch.bfloat16).to(dev)
    Cb = torch.randn(Nc, d, generator=gen).to(torch.bfloat16).to(dev)
    m8 = (torch.rand(Nc, generator=gen) < 0.05).to(torch.uint8).to(dev)
    outs = {}
    defaults = {k: _lib.get_option(k) for k in ("g128_dma", "tile")}
    try:
        _lib.set_option("tile", 0)  # (pins the 128 x 128 tile for the single-GEMM launches; the large-shape kernels stand aside)
        for mode in (0, 1):
            _lib.set_option("g128_dma", mode)
            outs[mode] = (kn.dq(G, Cb, 1.0).clone(), kn.dc(G, Qb, 1.0).clone(), kn.sim(Qb, Cb, m8, 0.5).clone())
    finally:
        for k, v in defaults.items():
            _lib.set_option(k, v)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    ref_dq = G.float() @ Cb.float()
    ref_dc = G.float().t() @ Qb.float()
    assert ((outs[1][0] - ref_dq).abs().max() / ref_dq.abs().max()).item() <= 1e-3
    assert ((outs[1][1] - ref_dc).abs().max() / ref_dc.abs().max()).item() <= 1e-3
