"""The C-ABI library loads on a GPU-less box and exports every symbol include/dprhot.h declares; argument
validation (pure host code, no kernel launch) behaves as documented."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "dprhot.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dprhot_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from dpr_scale_amd import _lib

    syms = declared_symbols()
    assert len(syms) >= 13
    for s in syms:
        assert hasattr(_lib.lib, s), f"libdprhot.so does not export {s}"
        assert s in _lib.SIGNATURES, f"dpr_scale_amd/_lib.py does not bind {s}"
    assert sorted(_lib.SIGNATURES) == syms, "binding lists symbols the header does not declare"


def test_version_and_header_agree():
    from dpr_scale_amd import _lib

    m = re.search(r"#define\s+DPRHOT_VERSION\s+(\d+)", open(HEADER).read())
    assert _lib.version() == int(m.group(1))


def test_argument_validation_is_host_side():
    from dpr_scale_amd import _lib

    lib = _lib.lib
    out = ctypes.c_size_t(0)
    assert lib.dprhot_workspace_bytes(128, 8192, 768, ctypes.byref(out)) == 0 and out.value >= 128 * 8192 * 4
    assert lib.dprhot_workspace_bytes(128, 8190, 768, ctypes.byref(out)) == -1  # Nc % 8
    assert b"multiple of 8" in lib.dprhot_last_error()
    assert lib.dprhot_workspace_bytes(0, 8, 8, ctypes.byref(out)) == -1
    assert lib.dprhot_sim_fwd(None, 4, None, 8, 128, None, 1.0, None, None) == -1  # NULL pointers
    assert lib.dprhot_cast_bf16(None, None, 8, None) == -1
    assert lib.dprhot_topk(None, 1, 1, 1, None, None, None) == -1
    # the newer entry points validate on the host as well (nothing below touches a device)
    assert lib.dprhot_topk_update(None, 1, 8, 8, 0, 1, None, None, 1, None) == -1
    assert lib.dprhot_search(None, 1, None, 8, 64, 0, 1, 8, None, None, 1, None, 0, None) == -1
    assert lib.dprhot_search_workspace_bytes(4, 12, ctypes.byref(out)) == -1  # chunk % 8
    assert lib.dprhot_search_workspace_bytes(4, 16, ctypes.byref(out)) == 0 and out.value >= 4 * 16 * 8
    null = None
    assert lib.dprhot_inbatch_step_f32(null, null, null, null, 4, 8, 64, null, 0, null, 1.0, 1.0, 1.0, null, null, null, null,
                                       null, null, null, null, null, 0, null) == -1
    assert b"required" in lib.dprhot_last_error()
    assert lib.dprhot_inbatch_step_packed_f32(null, null, null, 4, 2, 5, 8, 64, null, 1.0, 1.0, 1.0, null, null, null, null, null,
                                              null, null, null, 0, null) == -1  # rank >= W, NULL pointers
    assert lib.dprhot_comm_init(null, 2, 0, null) == -1
    assert lib.dprhot_allgather_ctx(null, null, null, 0, null) == -1
    assert lib.dprhot_comm_destroy(null) == 0  # destroying nothing is fine
    # round 2: the no-logits forward and the score-free rank
    assert lib.dprhot_dscores(null, 8192, null, 8192, 768, null, 0, null, 1.0, 1.0, null, null, null, 0, null) == -1  # NULL pointers
    assert lib.dprhot_sim_rank(null, 8, null, 8, 64, null, 0, null, 1.0, null, null, 0, null) == -1
    big, small = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert lib.dprhot_workspace_bytes(8192, 65536, 768, ctypes.byref(big)) == 0
    assert big.value < 8192 * 65536 * 4, "a no-logits shape must not reserve a logit buffer"
    assert lib.dprhot_workspace_bytes(512, 8192, 768, ctypes.byref(small)) == 0 and small.value >= 512 * 8192 * 4  # logits stored here (64 tiles: below option nl_min = 128)
    # round 3: the operator's step, the grad_output fix-up, the gradient-hook legs
    assert lib.dprhot_train_step_f32(null, null, null, null, 4, 8, 64, null, 0, null, 1.0, 1.0, 1.0, null, null, null, null, null, null,
                                     null, null, 2, null, 0, null) == -1
    assert lib.dprhot_train_step_f32(null, null, null, null, 4, 8, 64, null, 0, null, 1.0, 1.0, 1.0, null, null, null, null, null, null,
                                     null, null, 1, null, 0, null) == -3  # fp16 partials: no such epilogue
    assert lib.dprhot_train_step_f32(null, null, null, null, 4, 8, 64, null, 0, null, 1.0, 1.0, 1.0, null, null, null, null, null, null,
                                     null, null, 0, null, 0, null) == -3  # bf16 partials: only the skinny plan writes them
    assert lib.dprhot_train_step_packed_f32(null, null, null, 4, 2, 5, 8, 64, null, 1.0, 1.0, 1.0, null, null, null, null, null, null,
                                            null, null, 2, null, 0, null) == -1
    assert lib.dprhot_rescale_grads(null, 8, null, 0, null, 0, 2, null, null, null, null) == -1
    n = ctypes.c_int(-1)
    # cfg3 per rank: the step can run without the dScores (round 4) and then finishes dQ itself -- nothing deferred, G optional; with
    # that plan switched off it is the split-K slabs of round 3 again
    w = ctypes.c_int(-1)
    assert lib.dprhot_train_dq_slabs(128, 8256, 768, ctypes.byref(n)) == 0 and n.value == 0
    assert lib.dprhot_step_wants_g(128, 8256, 768, ctypes.byref(w)) == 0 and w.value == 0
    assert lib.dprhot_set_option(b"sk_fused", 0) == 0
    assert lib.dprhot_train_dq_slabs(128, 8256, 768, ctypes.byref(n)) == 0 and 1 <= n.value <= 64
    assert lib.dprhot_step_wants_g(128, 8256, 768, ctypes.byref(w)) == 0 and w.value == 1
    assert lib.dprhot_set_option(b"sk_fused", 1) == 0
    assert lib.dprhot_step_wants_g(32, 256, 768, ctypes.byref(w)) == 0 and w.value == 1             # cfg2: G is written by the fused small step
    assert lib.dprhot_step_wants_g(128, 1032, 768, ctypes.byref(w)) == 0 and w.value == 1           # narrow sim units: the four-launch plan
    assert lib.dprhot_train_dq_slabs(32, 256, 768, ctypes.byref(n)) == 0 and n.value == 0          # cfg2: dQ comes out whole
    # where the plan without the dScores launch is chosen (round 4, eight-wave units): from 2^19 scores up, where the few-rows plan's
    # own dQ slices fit a unit's factor table; sk_fused = 2 takes it wherever it exists (it then cuts its own slices)
    for shape, want_g in (((128, 4160, 768), 0), ((64, 8256, 512), 0), ((96, 6208, 768), 0), ((128, 4096, 768), 0), ((128, 2112, 768), 1),
                          ((64, 4160, 768), 1), ((128, 8256, 1024), 1), ((128, 12384, 768), 1), ((128, 16512, 768), 1)):
        assert lib.dprhot_step_wants_g(*shape, ctypes.byref(w)) == 0 and w.value == want_g, (shape, w.value)
    assert lib.dprhot_set_option(b"sk_fused", 2) == 0
    big = ctypes.c_size_t(0)
    for shape, want_g in (((128, 2112, 768), 1), ((128, 8256, 1024), 0), ((128, 12384, 768), 0), ((128, 16512, 768), 1)):  # (2112: narrow sim units)
        assert lib.dprhot_step_wants_g(*shape, ctypes.byref(w)) == 0 and w.value == want_g, (shape, w.value)
        # its slabs fit the workspace: ceil(ceil(Nc / 64) / 13) slices of B x d floats (plus everything else in there)
        assert lib.dprhot_workspace_bytes(*shape, ctypes.byref(big)) == 0
        assert big.value >= -(-(-(-shape[1] // 64)) // 13) * shape[0] * shape[2] * 4
    assert lib.dprhot_set_option(b"sk_fused", 1) == 0
    for name, default in ((b"sk_w8", 1), (b"sk_sim_w8", 1), (b"sk_pair", 0), (b"sk_tail", 0), (b"sk_sim_priv", 1), (b"sk_fused", 1), (b"nt_stores", 1)):
        assert lib.dprhot_get_option(name, ctypes.byref(n)) == 0 and n.value == default, (name, n.value)
    # the few-rows plan's measured boundaries ("Mid-size steps"), seen through the slab count: split-K slabs where
    # the plan applies with more than 512 contexts, none where its dQ units are unsplit or another plan has the shape
    for shape, want_slabs in (((128, 1032, 768), True), ((64, 1088, 1024), True), ((32, 1056, 768), True), ((32, 2112, 768), True),
                              ((128, 264, 768), False), ((128, 520, 768), True), ((96, 512, 768), False), ((64, 392, 768), False),
                              ((32, 528, 768), False), ((32, 1024, 768), False), ((64, 256, 768), False), ((100, 1032, 768), False),
                              ((128, 1032, 800), False)):
        assert lib.dprhot_train_dq_slabs(*shape, ctypes.byref(n)) == 0 and (n.value > 1) == want_slabs, (shape, n.value)
    # explicit option table instead of environment variables
    assert lib.dprhot_set_option(b"no_skinny", 1) == 0
    assert lib.dprhot_train_dq_slabs(128, 8256, 768, ctypes.byref(n)) == 0 and n.value == 0
    assert lib.dprhot_get_option(b"no_skinny", ctypes.byref(n)) == 0 and n.value == 1
    assert lib.dprhot_set_option(b"no_skinny", 0) == 0
    assert lib.dprhot_set_option(b"no_such_option", 1) == -1 and b"unknown option" in lib.dprhot_last_error()
    assert b"getenv" not in open(_lib.LIB_PATH, "rb").read()  # the library reads no environment variable
    assert lib.dprhot_grad_pack(null, 8, 1.0, 0, null, 8, null) == -1
    buf = (ctypes.c_float * 16)()
    assert lib.dprhot_grad_pack(buf, 9, 1.0, 0, buf, 12, null) == -1 and b"n_padded" in lib.dprhot_last_error()  # n_padded % 8
    assert lib.dprhot_grad_pack(buf, 9, 1.0, 5, buf, 16, null) == -1  # wire kind
    assert lib.dprhot_grad_sum_shards(buf, 2, 12, 0, 0, buf, null) == -1  # shard % 8
    assert lib.dprhot_grad_sum_shards(buf, 2, 8, 0, 1, buf, null) == -1   # bf16 in, fp16 out
    assert lib.dprhot_grad_unpack(null, 0, null, 8, null) == -1
    rows = ctypes.c_int(0)
    assert lib.dprhot_packed_rows(256, 768, ctypes.byref(rows)) == 0 and rows.value == 264  # 256 rows + 1 mask row -> 8-row multiple
    with pytest.raises(_lib.DprhotError):
        _lib.check(-1, "x")


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import importlib
    import sys

    monkeypatch.setenv("DPRHOT_LIB", str(tmp_path / "nope.so"))
    saved = sys.modules.pop("dpr_scale_amd._lib", None)
    try:
        with pytest.raises(ImportError, match="no CPU fallback"):
            importlib.import_module("dpr_scale_amd._lib")
    finally:
        sys.modules.pop("dpr_scale_amd._lib", None)
        if saved is not None:
            sys.modules["dpr_scale_amd._lib"] = saved

def test_plans_are_total_functions_of_the_shape():
    """Every entry point derives its plans (tiles, splits, workspace layout) from (B, Nc, d) on the host before anything is launched:
    they must be defined for EVERY accepted shape -- a hidden size below 64 once divided by zero inside the skinny-step plan that
    the workspace layout consults for all shapes (caught by scripts/fuzz_step.py on the GPU; reproduced here without one)."""
    import itertools

    from dpr_scale_amd import _lib

    n = 0
    for B, Nc, d in itertools.product([1, 8, 32, 33, 64, 128, 129, 300, 1024, 8192],
                                      [8, 64, 256, 1152, 2048, 2112, 4800, 8256, 65536],
                                      [8, 16, 24, 56, 64, 80, 128, 768, 1024, 4096, 30528]):
        assert _lib.workspace_bytes(B, Nc, d) > 0
        n += 1
    assert n == 990
    for nq, chunk in itertools.product([1, 7, 1024], [8, 1024, 65536]):
        assert _lib.search_workspace_bytes(nq, chunk) >= nq * chunk * 8
