"""The plain-C restatement (oracle/inbatch_oracle.c) against the reference-generated fixtures.  CPU only."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle import c_oracle
from oracle import inbatch_oracle as O


@pytest.fixture(scope="module")
def lib():
    return c_oracle.load()


def rel(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("name", golden_names("cfg1") + golden_names("cfg2"))
def test_c_port_full_tensor_cases(name, lib):
    meta, g = load_golden(name)
    q, c, y, m = O.synth_embeddings(meta["seed"], meta["B"], meta["K"], meta["d"], meta["dist"], meta["ragged"])
    r = c_oracle.train_step(lib, q, c, y, 0, m, meta["T"], meta["B"])
    assert abs(r["loss_sum"] / meta["B"] - g["loss"]) <= 3e-6 * max(1, abs(g["loss"]))
    fin = np.isfinite(g["S"])
    assert np.array_equal(fin, np.isfinite(r["S"]))
    assert rel(r["S"][fin] * meta["T"], g["S"][fin]) < 3e-6
    assert rel(r["dQ"], g["dQ"]) < 3e-4 and rel(r["dC"], g["dC"]) < 3e-4
    ranks = np.empty(meta["B"], np.int64)
    lib.oracle_rank_of_gold(c_oracle.ptr(r["S"]), meta["B"], r["S"].shape[1], c_oracle.ptr(y), 0, c_oracle.ptr(ranks))
    assert np.array_equal(ranks, g["ranks"])


def test_c_port_ddp_case(lib):
    meta, g = load_golden("w4_ddp")
    W, B, K = meta["W"], meta["B"], meta["K"]
    parts = [O.synth_embeddings(meta["seed"] + r, B, K, meta["d"], meta["dist"], meta["ragged"]) for r in range(W)]
    C = np.concatenate([p[1] for p in parts])
    m = np.concatenate([p[3] for p in parts])
    outs = [c_oracle.train_step(lib, parts[r][0], C, parts[r][2], r * B * K, m, meta["T"], W * B) for r in range(W)]
    loss = sum(o["loss_sum"] for o in outs) / (W * B)
    dC = sum(o["dC"].astype(np.float64) for o in outs)
    for r in range(W):
        assert abs(loss - g["loss_per_rank"][r]) <= 3e-6 * max(1, abs(loss))
        assert rel(outs[r]["dQ"], g["dq_per_rank"][r]) < 3e-4
        assert rel(dC[r * B * K:(r + 1) * B * K], g["dc_per_rank"][r]) < 3e-4


def test_c_port_ties(lib):
    _, g = load_golden("ties")
    S = np.ascontiguousarray(g["S"], np.float32)
    ranks = np.empty(S.shape[0], np.int64)
    lib.oracle_rank_of_gold(c_oracle.ptr(S), S.shape[0], S.shape[1], c_oracle.ptr(np.ascontiguousarray(g["y"])), 0,
                            c_oracle.ptr(ranks))
    assert np.array_equal(ranks, g["ranks"])
