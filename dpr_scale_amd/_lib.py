"""ctypes binding of libdprhot.so (include/dprhot.h).  There is no CPU fallback: if the library is missing
or fails to load, importing this module raises -- the product path must fail loudly (it never routes
through oracle/)."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DPRHOT_LIB", os.path.join(_HERE, "libdprhot.so"))

if not os.path.isfile(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C dpr_scale_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback."
    )

try:  # torch first, so that libamdhip64.so.7 resolves to the runtime torch already loaded (one HIP runtime)
    import torch  # noqa: F401
except Exception:  # pragma: no cover - the library also works without torch (system ROCm runtime)
    torch = None

lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)

# name -> (restype, argtypes); must list every symbol include/dprhot.h declares (tests/test_abi.py checks)
SIGNATURES = {
    "dprhot_version": (c_int, []),
    "dprhot_last_error": (c_char_p, []),
    "dprhot_set_option": (c_int, [c_char_p, c_int]),
    "dprhot_get_option": (c_int, [c_char_p, POINTER(c_int)]),
    "dprhot_options_epoch": (ctypes.c_longlong, []),
    "dprhot_workspace_bytes": (c_int, [c_int, c_int, c_int, POINTER(c_size_t)]),
    "dprhot_cast_bf16": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "dprhot_prep": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "dprhot_packed_rows": (c_int, [c_int, c_int, POINTER(c_int)]),
    "dprhot_pack_ctx": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "dprhot_unpack_mask": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dprhot_sim_fwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p]),
    "dprhot_softmax_ce_fwd_bwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_float, c_void_p, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p]),
    "dprhot_reduce_sum": (c_int, [c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "dprhot_dq": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                          c_size_t, c_void_p]),
    "dprhot_dc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "dprhot_rank_of_gold": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "dprhot_topk": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dprhot_topk_update": (c_int, [c_void_p, c_int, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "dprhot_topk_wide_workspace_bytes": (c_int, [c_int, c_int, POINTER(c_size_t)]),
    "dprhot_topk_update_wide": (c_int, [c_void_p, c_int, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "dprhot_search_workspace_bytes": (c_int, [c_int, c_int, POINTER(c_size_t)]),
    "dprhot_search": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_int,
                              c_void_p, c_size_t, c_void_p]),
    "dprhot_inbatch_fwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_float,
                                   c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                   c_void_p]),
    "dprhot_sim_stats": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_float, c_void_p,
                                 c_void_p, c_size_t, c_void_p]),
    "dprhot_softmax_finish": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int64, c_float, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dprhot_dscores": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_float, c_float, c_void_p,
                               c_void_p, c_void_p, c_size_t, c_void_p]),
    "dprhot_sim_rank": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_float, c_void_p, c_void_p,
                                c_size_t, c_void_p]),
    "dprhot_sim_rank_loss": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_float, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dprhot_sim_stats_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int64, c_void_p,
                                     c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dprhot_inbatch_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int64,
                                       c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_size_t, c_void_p]),
    "dprhot_inbatch_step_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int64,
                                        c_void_p, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dprhot_inbatch_step_packed_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_float,
                                               c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_size_t, c_void_p]),
    "dprhot_pairwise_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dprhot_pairwise_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dprhot_train_dq_slabs": (c_int, [c_int, c_int, c_int, POINTER(c_int)]),
    "dprhot_step_wants_g": (c_int, [c_int, c_int, c_int, POINTER(c_int)]),
    "dprhot_train_step_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_float,
                                      c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int, c_void_p, c_size_t, c_void_p]),
    "dprhot_train_step_packed_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_float, c_float,
                                             c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_int, c_void_p, c_size_t, c_void_p]),
    "dprhot_rescale_grads": (c_int, [c_void_p, c_size_t, c_void_p, c_int, c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "dprhot_grad_pack": (c_int, [c_void_p, c_size_t, c_float, c_int, c_void_p, c_size_t, c_void_p]),
    "dprhot_grad_sum_shards": (c_int, [c_void_p, c_int, c_size_t, c_int, c_int, c_void_p, c_void_p]),
    "dprhot_grad_unpack": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "dprhot_comm_unique_id": (c_int, [c_void_p]),
    "dprhot_comm_init": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "dprhot_comm_destroy": (c_int, [c_void_p]),
    "dprhot_allgather_ctx": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dprhot_reducescatter_dc": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dprhot_reducescatter_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "dprhot_allreduce_sum": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "dprhot_allgather_allpairs": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dprhot_comm_has_allpairs": (c_int, [c_void_p]),
    "dprhot_fwd_no_logits": (c_int, [c_int, c_int, c_int, POINTER(c_int)]),
    "dprhot_fwd_one_pass": (c_int, [c_int, c_int, c_int, POINTER(c_int)]),
    "dprhot_reducescatter_allpairs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "dprhot_inbatch_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here == ABI mismatch: fail at import
    _fn.restype = _res
    _fn.argtypes = _args


class DprhotError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib.dprhot_last_error()
        raise DprhotError(f"{what or 'dprhot'} failed with code {rc}: {msg.decode() if msg else ''}")


def version() -> int:
    return lib.dprhot_version()




def options_epoch() -> int:
    """The LIBRARY's own counter (dprhot_options_epoch, bumped by every dprhot_set_option whoever calls it): host-side caches of plan
    facts (slab counts, whether a step wants a G buffer) key on it."""
    return int(lib.dprhot_options_epoch())


def set_option(name: str, value: int):
    """Process-wide test / A-B switch of the plans (include/dprhot.h); production never calls this."""
    check(lib.dprhot_set_option(name.encode(), int(value)), f"dprhot_set_option({name})")


def get_option(name: str) -> int:
    out = c_int(0)
    check(lib.dprhot_get_option(name.encode(), ctypes.byref(out)), f"dprhot_get_option({name})")
    return out.value


def search_workspace_bytes(nq, chunk):
    out = c_size_t(0)
    check(lib.dprhot_search_workspace_bytes(int(nq), int(chunk), ctypes.byref(out)), "dprhot_search_workspace_bytes")
    return out.value


def packed_rows(n_ctx: int, d: int) -> int:
    out = c_int(0)
    check(lib.dprhot_packed_rows(n_ctx, d, ctypes.byref(out)), "dprhot_packed_rows")
    return out.value


def train_dq_slabs(B: int, Nc: int, d: int) -> int:
    out = c_int(0)
    check(lib.dprhot_train_dq_slabs(B, Nc, d, ctypes.byref(out)), "dprhot_train_dq_slabs")
    return out.value


def step_wants_g(B: int, Nc: int, d: int) -> bool:
    out = c_int(1)
    check(lib.dprhot_step_wants_g(B, Nc, d, ctypes.byref(out)), "dprhot_step_wants_g")
    return bool(out.value)


def fwd_no_logits(B: int, Nc: int, d: int) -> bool:
    """True when this shape's forward never stores the logits (dprhot_fwd_no_logits)."""
    out = c_int(0)
    check(lib.dprhot_fwd_no_logits(B, Nc, d, ctypes.byref(out)), "dprhot_fwd_no_logits")
    return bool(out.value)


def fwd_one_pass(B: int, Nc: int, d: int) -> int:
    """How the fused forward forms G without stored logits (dprhot_fwd_one_pass): 0 logits stored, 1 one pass on the 256 x 256 tile,
    2 one pass on the 128 x 128 LDS-DMA tile."""
    out = c_int(0)
    check(lib.dprhot_fwd_one_pass(B, Nc, d, ctypes.byref(out)), "dprhot_fwd_one_pass")
    return int(out.value)


def workspace_bytes(B: int, Nc: int, d: int) -> int:
    out = c_size_t(0)
    check(lib.dprhot_workspace_bytes(B, Nc, d, ctypes.byref(out)), "dprhot_workspace_bytes")
    return out.value
