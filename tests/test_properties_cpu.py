"""Property tests of the oracle identities and of the host-side planning code (CPU only, hypothesis)."""
import ctypes

import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import inbatch_oracle as O


@settings(max_examples=25, deadline=None)
@given(W=st.integers(1, 4), B=st.integers(1, 6), K=st.integers(1, 4), d=st.sampled_from([8, 16, 64]),
       T=st.sampled_from([0.5, 1.0, 4.0]), seed=st.integers(0, 10_000), ragged=st.booleans())
def test_local_rows_plus_reduce_scatter_equals_global_step(W, B, K, d, T, seed, ragged):
    """The identity the whole multi-rank design rests on, for arbitrary small shapes."""
    parts = [O.synth_embeddings(seed + r, B, K, d, "U", ragged) for r in range(W)]
    Q = np.concatenate([p[0] for p in parts])
    C = np.concatenate([p[1] for p in parts])
    y = O.gathered_labels(np.stack([p[2] for p in parts]), B * K)
    m = np.concatenate([p[3] for p in parts])
    ref = O.training_step_global(Q, C, y, m, T)
    loss_sum, dC = 0.0, 0.0
    for r in range(W):
        o = O.training_step_rank(parts[r][0], C, y[r * B:(r + 1) * B], m, T, W * B)
        loss_sum += o["loss_sum"]
        dC = dC + o["dC_part"]
        assert np.allclose(o["dq"], ref["dQ"][r * B:(r + 1) * B], rtol=1e-10, atol=1e-14)
    assert abs(loss_sum / (W * B) - ref["loss"]) <= 1e-12 * max(1.0, abs(ref["loss"]))
    assert np.allclose(dC, ref["dC"], rtol=1e-10, atol=1e-14)


@settings(max_examples=50, deadline=None)
@given(rows=st.integers(1, 8), cols=st.integers(1, 40), levels=st.integers(1, 5), seed=st.integers(0, 10_000))
def test_rank_of_gold_is_position_in_stable_descending_sort(rows, cols, levels, seed):
    rng = np.random.default_rng(seed)
    S = rng.integers(0, levels, (rows, cols)).astype(np.float32)  # few distinct values: many ties
    y = rng.integers(0, cols, rows)
    order = np.argsort(-S, axis=1, kind="stable")
    want = np.array([int(np.nonzero(order[i] == y[i])[0][0]) + 1 for i in range(rows)])
    assert np.array_equal(O.rank_of_gold(S, y), want)
    k = min(cols, 5)
    v, idx = O.topk_stable(S, k)
    assert np.array_equal(idx, order[:, :k])


@settings(max_examples=50, deadline=None)
@given(rows=st.integers(1, 5), cols=st.integers(1, 60), k=st.integers(1, 12), levels=st.integers(1, 6),
       cuts=st.lists(st.integers(0, 60), max_size=4), seed=st.integers(0, 10_000))
def test_topk_fold_over_pieces_equals_topk_of_the_concatenation(rows, cols, k, levels, cuts, seed):
    """run_retrieval_pytorch.py:196-243 + :272-277 as a fold (oracle.topk_merge): any split into pieces, many ties."""
    rng = np.random.default_rng(seed)
    S = rng.integers(0, levels, (rows, cols)).astype(np.float32)
    S[rng.random(S.shape) < 0.1] = -np.inf
    edges = sorted({0, cols, *[c for c in cuts if c < cols]})
    state = None
    for a, b in zip(edges[:-1], edges[1:]):
        state = O.topk_merge(state, S[:, a:b], 100 + a, k)
    kk = min(k, cols)
    v, i = O.topk_stable(S, kk)
    assert np.array_equal(state[0][:, :kk], v) and np.array_equal(state[1][:, :kk], i + 100)
    assert np.all(state[1][:, kk:] == -1) and np.all(np.isneginf(state[0][:, kk:]))


def test_corpus_search_host_logic_with_ragged_shards():
    """CorpusSearch (host side of dprhot_search): shards of any length, id offsets, chunking -- against one big top-k."""
    import torch
    from _oracle_kernels import OracleKernels
    from dpr_scale_amd.hotpath import CorpusSearch

    rng = np.random.default_rng(5)
    q = torch.from_numpy(rng.integers(-2, 3, (7, 16)).astype(np.float32))
    shards = [torch.from_numpy(rng.integers(-2, 3, (n, 16)).astype(np.float32)) for n in (37, 8, 5, 120)]
    s = CorpusSearch(q, 9, chunk=16, kernels=OracleKernels())
    first = 0
    for sh in shards:
        s.add(sh, first)
        first += sh.shape[0]
    v, i = s.result()
    S = (q.double() @ torch.cat(shards).double().T).float().numpy()
    wv, wi = O.topk_stable(S, 9)
    assert np.array_equal(i.numpy(), wi) and np.array_equal(v.numpy(), wv)


@settings(max_examples=100, deadline=None)
@given(st.floats(allow_nan=False, allow_infinity=False, width=32))
def test_bf16_round_is_nearest_even(x):
    x = np.float32(x)
    r = O.bf16_round(np.array([x]))[0]
    bits = np.array([r]).view(np.uint32)[0]
    assert bits & 0xFFFF == 0  # representable in bf16
    if np.isfinite(r) and x != 0 and abs(float(x)) > 1e-30:
        assert abs(float(r) - float(x)) <= abs(float(x)) * 2.0 ** -8


def test_host_side_plans_are_consistent():
    """Pure host code of the C ABI: packed row count, workspace growth, argument validation paths."""
    from dpr_scale_amd import _lib

    for n_ctx, d in [(256, 768), (8, 8), (1024, 1024), (264, 136), (2048, 64)]:
        rows = _lib.packed_rows(n_ctx, d)
        assert rows % 8 == 0 and rows * d * 2 >= n_ctx * d * 2 + n_ctx and rows - n_ctx <= -(-n_ctx // (2 * d)) + (63 if n_ctx >= 2048 else 7)
        assert n_ctx < 2048 or rows % 64 == 0  # (round 6: whole 64-deep K steps over the gathered axis for the 256 x 256 backward)
    last = 0
    for B in [8, 32, 128, 512]:  # (1024 x 8192 is a no-logits shape since round 6 -- 128 tiles, option nl_min: no logit buffer there)
        w = _lib.workspace_bytes(B, 8192, 768)
        assert w >= B * 8192 * 4 and w > last
        last = w
    out = ctypes.c_int(0)
    assert _lib.lib.dprhot_packed_rows(0, 768, ctypes.byref(out)) == -1
    assert _lib.lib.dprhot_pack_ctx(None, None, 8, 8, None, None) == -1


def test_search_over_pickled_shards_follows_the_reference_flow(tmp_path):
    """dpr_scale_amd.retrieval.search_shards (reps_* pickles, sorted glob, build_index sizing rule, shard id offsets)
    against a direct restatement of run_retrieval_pytorch.py:177-243 + :272-277 in numpy."""
    import pickle

    import torch
    from _oracle_kernels import OracleKernels
    from dpr_scale_amd.retrieval import search_shards

    rng = np.random.default_rng(3)
    d, sizes = 16, [40, 40, 40, 25]  # the short last file leaves zero rows in its shard, as in the reference
    files = []
    for i, n in enumerate(sizes):
        v = rng.integers(-3, 4, (n, d)).astype(np.float32)
        files.append(v)
        with open(tmp_path / f"reps_{i:04}.pkl", "wb") as f:
            pickle.dump(v, f)
    q = rng.integers(-3, 4, (6, d)).astype(np.float32)
    with open(tmp_path / "q.pkl", "wb") as f:
        pickle.dump(q, f)
    for shard in (1, 2):
        v, i = search_shards(str(tmp_path / "q.pkl"), str(tmp_path), 7, shard=shard, device="cpu", chunk=16, kernels=OracleKernels())
        # the reference: per shard an index of rows(first file) * files rows, per-shard top-k, ids offset by len(index), re-merge
        per = len(files) // shard
        cand_s, cand_i, offset = [], [], 0
        for s in range(shard):
            fs = files[s * per:(s + 1) * per]
            index = np.zeros((fs[0].shape[0] * len(fs), d), np.float32)
            n = 0
            for f in fs:
                index[n:n + f.shape[0]] = f
                n += f.shape[0]
            sv, si = O.topk_stable(q @ index.T, 7)
            cand_s.append(sv)
            cand_i.append(si + offset)
            offset += index.shape[0]
        cs, ci = np.concatenate(cand_s, 1), np.concatenate(cand_i, 1)
        order = np.lexsort((ci, -cs), axis=1)[:, :7]
        assert np.array_equal(i.numpy(), np.take_along_axis(ci, order, 1))
        assert np.array_equal(v.numpy(), np.take_along_axis(cs, order, 1))


def test_sim_score_refuses_to_return_detached_scores_to_a_training_caller():
    """A kernels object without backward GEMMs must not hand scores without a grad_fn to a caller whose inputs require grad
    (the encoders would silently receive no gradient)."""
    import pytest
    import torch

    from _oracle_kernels import OracleKernels
    from dpr_scale_amd import hotpath

    class NoBackward(OracleKernels):
        dq = property()  # hasattr(...) is False
        dc = property()

    q = torch.randn(4, 16, requires_grad=True)
    c = torch.randn(8, 16)
    with pytest.raises(RuntimeError, match="requires grad"):
        hotpath.sim_score(q, c, None, 1.0, NoBackward())
    assert hotpath.sim_score(q.detach(), c, None, 1.0, NoBackward()).shape == (4, 8)  # inference callers are unaffected
