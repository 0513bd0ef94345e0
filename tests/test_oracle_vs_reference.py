"""The oracle against the reference itself, run live (only where /root/reference is mounted: the build container; the
fixtures under tests/golden/ carry the same pinning to the GPU box).  Also SURVEY.md section 8 c5's secondary check: what
changes when the reference runs under torch.autocast(bf16), as it does in production (`precision: 16`)."""
import numpy as np
import pytest
import torch

from oracle import inbatch_oracle as O
from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="needs /root/reference (build container only)")


def rel(a, b):
    return np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("B,K,d,dist,ragged,T", [(4, 2, 128, "P", False, 1.0), (16, 4, 256, "U", True, 0.5), (32, 8, 768, "U", True, 0.05)])
def test_numpy_oracle_and_torch_restatement_equal_the_live_reference(B, K, d, dist, ragged, T):
    q, c, y, m = O.synth_embeddings(900 + B, B, K, d, dist, ragged)
    tq, tc, ty, tm = torch.from_numpy(q), torch.from_numpy(c), torch.from_numpy(y), torch.from_numpy(m)
    loss, dq, dc = ref_shim.reference_training_step(tq, tc, ty, tm, temperature=T)
    r = O.training_step_global(q, c, y, m, T)
    assert abs(r["loss"] - loss.item()) <= 2e-6 * max(1.0, abs(loss.item()))
    assert rel(r["dQ"], dq.numpy()) <= 3e-4 and rel(r["dC"], dc.numpy()) <= 3e-4
    from oracle.torch_steps import reference_step_torch

    l2, dq2, dc2 = reference_step_torch(tq, tc, ty, tm, T)
    assert abs(l2.item() - loss.item()) <= 1e-6 * max(1.0, abs(loss.item()))
    assert rel(dq2.numpy(), dq.numpy()) <= 1e-6 and rel(dc2.numpy(), dc.numpy()) <= 1e-6
    S = ref_shim.reference_sim_score(tq, tc, tm).numpy()
    So = O.sim_score(q, c, m)
    fin = np.isfinite(S)
    assert np.array_equal(fin, np.isfinite(So)) and rel(So[fin], S[fin]) <= 2e-6


def test_reference_under_bf16_autocast_informational():
    """The reference trains with AMP: its matmul runs in reduced precision and the LOGITS are rounded to bf16 before the
    cross-entropy (fp32).  The parity protocol (bf16-representable inputs, fp32 reference) removes the input rounding;
    what is left -- and what this test quantifies -- is the rounding of the logits themselves, which the HIP path does NOT
    do (fp32 accumulate, fp32 logits).  Loose bars: the point is the order of magnitude, recorded in DESIGN.md section 7."""
    q, c, y, m = O.synth_embeddings(77, 32, 8, 768, "U", True)
    tq, tc, ty, tm = torch.from_numpy(q), torch.from_numpy(c), torch.from_numpy(y), torch.from_numpy(m)
    loss32, dq32, dc32 = ref_shim.reference_training_step(tq, tc, ty, tm, temperature=1.0)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        loss16, dq16, dc16 = ref_shim.reference_training_step(tq, tc, ty, tm, temperature=1.0)
    e_loss = abs(loss16.item() - loss32.item()) / max(1.0, abs(loss32.item()))
    e_dq, e_dc = rel(dq16.float().numpy(), dq32.numpy()), rel(dc16.float().numpy(), dc32.numpy())
    print(f"autocast(bf16) vs fp32 reference: loss rel {e_loss:.2e}, dQ rel-to-max {e_dq:.2e}, dC rel-to-max {e_dc:.2e}")
    assert e_loss <= 2e-2 and e_dq <= 1e-1 and e_dc <= 1e-1   # bf16 logits: ~1e-3 .. 1e-2, an order above the HIP path's deviation
    assert e_loss > 0 or e_dq > 0  # it IS a different computation


@pytest.mark.parametrize("in_batch,teacher_coef", [(True, 0.0), (False, 0.0), (True, 0.3), (False, 1.0)])
def test_router_restatement_equals_the_live_citadel_task(in_batch, teacher_coef):
    """oracle/router_oracle.py against /root/reference/dpr_scale/task/citadel_task.py run live."""
    from oracle import router_oracle as R

    q, c, mask, pos, teacher = R.synth_router(41, 6, 4, 1000)
    t = torch.from_numpy
    loss, dq, dc, logged = ref_shim.reference_router_loss(t(q), t(c), t(mask), t(pos), t(teacher), in_batch, teacher_coef, 2.0)
    l2, dq2, dc2 = R.router_step(q, c, mask, pos, teacher, in_batch, teacher_coef, 2.0)
    assert abs(l2 - loss.item()) <= 2e-6 * max(1.0, abs(loss.item()))
    assert rel(dq2, dq.numpy()) <= 2e-6 and rel(dc2, dc.numpy()) <= 2e-6
    for pairwise in (False, True):
        S = ref_shim.reference_citadel_sim_score(t(q), t(c), t(mask), pairwise).numpy()
        So = R.sim_score(t(q).double(), t(c).double(), t(mask), pairwise).numpy()
        fin = np.isfinite(S)
        assert np.array_equal(fin, np.isfinite(So)) and rel(So[fin], S[fin]) <= 2e-6
