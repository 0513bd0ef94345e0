"""The numpy oracle against the fixtures the REFERENCE produced (oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle import inbatch_oracle as O


def _global_inputs(meta):
    W, B, K, d = meta["W"], meta["B"], meta["K"], meta["d"]
    parts = [O.synth_embeddings(meta["seed"] + r, B, K, d, meta["dist"], meta["ragged"]) for r in range(W)]
    Q = np.concatenate([p[0] for p in parts])
    C = np.concatenate([p[1] for p in parts])
    y = O.gathered_labels(np.stack([p[2] for p in parts]), B * K)
    m = np.concatenate([p[3] for p in parts])
    return parts, Q, C, y, m


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("name", golden_names("cfg1") + golden_names("cfg2"))
def test_full_tensor_cases(name):
    meta, g = load_golden(name)
    _, Q, C, y, m = _global_inputs(meta)
    r = O.training_step_global(Q, C, y, m, meta["T"])
    assert abs(r["loss"] - g["loss"]) <= 2e-6 * max(1, abs(g["loss"]))
    S_ref = g["S"] / meta["T"]
    fin = np.isfinite(S_ref)
    assert np.array_equal(fin, np.isfinite(r["S"]))
    assert rel(r["S"][fin], S_ref[fin]) < 2e-6
    assert rel(r["lse"], g["lse"]) < 2e-6
    assert rel(r["dQ"], g["dQ"]) < 3e-4
    assert rel(r["dC"], g["dC"]) < 3e-4
    S_unscaled = O.sim_score(Q, C, m)
    assert np.array_equal(O.rank_of_gold(S_unscaled, y), g["ranks"])
    rs, mrr, sc = O.rank_metrics(S_unscaled, y, k=1)
    assert rs == int(g["rank_metrics"][0]) and sc == int(g["rank_metrics"][2])
    assert abs(mrr - g["rank_metrics"][1]) < 1e-9 * max(1, mrr)


@pytest.mark.parametrize("name", golden_names("cfg3") + golden_names("cfg5"))
def test_summary_cases_local_rows_identity(name):
    """cfg3/cfg5: the local-rows formulation summed over ranks == the reference's global step."""
    meta, g = load_golden(name)
    W, B, K = meta["W"], meta["B"], meta["K"]
    parts, Q, C, y, m = _global_inputs(meta)
    own = meta["own_rank"]
    loss_sum, dC = 0.0, 0.0
    ranks, lse = [], []
    for r in range(W):
        out = O.training_step_rank(parts[r][0], C, y[r * B:(r + 1) * B], m, meta["T"], W * B)
        loss_sum += out["loss_sum"]
        dC = dC + out["dC_part"]
        lse.append(out["lse"])
        ranks.append(O.rank_of_gold(out["S"] * meta["T"], y[r * B:(r + 1) * B]))
        if r == own:
            assert rel(out["dq"], g["dq_own"]) < 3e-4
            si, sj = g["sample_i"], g["sample_j"]
            sel = (si >= r * B) & (si < (r + 1) * B)
            mine = out["S"][si[sel] - r * B, sj[sel]] * meta["T"]
            ref = g["sample_S"][sel]
            fin = np.isfinite(ref)
            assert np.array_equal(fin, np.isfinite(mine)) and rel(mine[fin], ref[fin]) < 2e-6
    assert abs(loss_sum / (W * B) - g["loss"]) <= 2e-6 * max(1, abs(g["loss"]))
    assert rel(np.concatenate(lse), g["lse"]) < 2e-6
    assert np.array_equal(np.concatenate(ranks), g["ranks"])
    dc_own = dC[own * B * K:(own + 1) * B * K]
    assert rel(dc_own.sum(1), g["dc_own_rowsum"]) < 2e-3  # cancelling sums: fp32 noise of the reference
    assert rel(dc_own.sum(0), g["dc_own_colsum"]) < 2e-3
    assert rel(dc_own[:64], g["dc_own_head"]) < 3e-4
    # sum_j dC_j == sum_i (sum_j G_ij) q_i == 0 analytically: the fixture holds only fp32 noise there
    assert np.abs(dC.sum(0)).max() < 1e-6 and np.abs(g["dC_colsum"]).max() < 1e-5


@pytest.mark.parametrize("name", ["w2_ddp", "w4_ddp", "cfg4_ddp"])
def test_real_ddp_branch(name):
    """Fixtures from the reference DDP branch on real gloo ranks: per-rank loss, q.grad, c.grad."""
    meta, g = load_golden(name)
    W, B, K = meta["W"], meta["B"], meta["K"]
    parts, Q, C, y, m = _global_inputs(meta)
    outs = [O.training_step_rank(parts[r][0], C, y[r * B:(r + 1) * B], m, meta["T"], W * B) for r in range(W)]
    loss = sum(o["loss_sum"] for o in outs) / (W * B)
    dC = sum(o["dC_part"] for o in outs)
    for r in range(W):
        assert abs(loss - g["loss_per_rank"][r]) <= 2e-6 * max(1, abs(loss))
        assert rel(outs[r]["dq"], g["dq_per_rank"][r]) < 3e-4
        assert rel(dC[r * B * K:(r + 1) * B * K], g["dc_per_rank"][r]) < 3e-4


def test_ties_rule():
    _, g = load_golden("ties")
    assert np.array_equal(O.rank_of_gold(g["S"], g["y"]), g["ranks"])
    v, i = O.topk_stable(g["S"], 16)
    assert np.array_equal(i, g["order16"])
    assert np.array_equal(v, g["topk_values"])


def test_non_inbatch_branch():
    meta, g = load_golden("nib")
    q, c, y, m = O.synth_embeddings(meta["seed"], meta["B"], meta["K"], meta["d"], meta["dist"], meta["ragged"])
    qc = O.non_inbatch_query_ctx_mask(y, m, q.shape[0])
    S = np.asarray(q, np.float64) @ np.asarray(c, np.float64).T
    S[qc] = -np.inf
    loss, _, lse = O.log_softmax_ce(S, y)
    G = O.dscores(S, y, lse, 1.0 / q.shape[0])
    assert abs(loss - g["loss"]) < 2e-6 * max(1, abs(g["loss"]))
    assert rel(G @ c.astype(np.float64), g["dQ"]) < 3e-4
    assert rel(G.T @ q.astype(np.float64), g["dC"]) < 3e-4


def test_bf16_helpers_roundtrip():
    x = np.random.default_rng(0).standard_normal(1000).astype(np.float32)
    r = O.bf16_round(x)
    assert np.array_equal(O.from_bf16_bits(O.to_bf16_bits(x)), r)
    assert np.array_equal(O.bf16_round(r), r)
    assert np.abs(r - x).max() <= np.abs(x).max() * 2 ** -8


@pytest.mark.parametrize("name", golden_names("cfg1") + golden_names("cfg2"))
def test_torch_ops_restatement_is_bit_identical_to_the_reference_fixture(name):
    """oracle/torch_steps.py (bench.py's `cpu_baseline_reference` leg: dpr_task.py:197-212 in torch CPU ops) against the
    loss / q.grad / c.grad the reference itself produced: same library, same op order -> the same bits."""
    import torch

    from oracle.torch_steps import reference_step_torch

    meta, g = load_golden(name)
    _, Q, C, y, m = _global_inputs(meta)
    torch.set_num_threads(8)
    loss, dq, dc = reference_step_torch(torch.from_numpy(Q), torch.from_numpy(C), torch.from_numpy(y), torch.from_numpy(m), meta["T"])
    assert abs(loss.item() - g["loss"]) <= 1e-6 * max(1.0, abs(g["loss"]))
    assert rel(dq.numpy(), g["dQ"]) <= 1e-6 and rel(dc.numpy(), g["dC"]) <= 1e-6
