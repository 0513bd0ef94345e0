"""The in-batch contrastive step of dpr-scale on MI355X: host side of the operator boundary.

Replaces, for one rank, dpr_scale/task/dpr_task.py:163-212 (gather -> sim_score -> /T -> CrossEntropyLoss) and
its autograd backward by:  fp32->bf16 cast | RCCL all-gather of context rows | HIP sim + softmax-CE + dScores |
HIP dQ / dC GEMMs | RCCL reduce-scatter of dC.  All device work goes through the C ABI of libdprhot.so
(include/dprhot.h); this file only moves pointers.  There is NO CPU path here: CPU tensors raise.

`kernels` arguments exist so that the CPU test-suite can exercise the distributed orchestration (gather
layout, label offsets, reduce-scatter, loss all-reduce) on gloo with a stand-in; the default -- and the only
thing the product ever passes -- is the HIP library.
"""
import ctypes
import os

import torch

from . import dist as D

_BF16 = torch.bfloat16
_KIND = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}  # storage kinds of include/dprhot.h


class _ExpectedGradScale:
    """Per device: the DEVICE scalar the next training step should scale its gradients by -- the grad_output the previous backward
    saw (AMP's loss scale; 1 for a plain backward()).  Every backward publishes a FRESH one-float tensor (written once by
    dprhot_rescale_grads, never modified afterwards), so a step that read an older one can still compare against it."""

    _cur = {}

    @classmethod
    def get(cls, dev, site=None):
        """`site`: what tells the operator's call sites of one model apart -- the step's shape (B, Nc, d).  CITADEL's router loss and the
        main loss receive different constant upstream factors in the same step; one shared prediction made each of them mispredict
        the other every step (and pay the full rescale pass)."""
        k = (dev, site)
        t = cls._cur.get(k)
        if t is None:
            t = cls._cur[k] = torch.ones(1, dtype=torch.float32, device=dev)
        return t

    @classmethod
    def publish(cls, dev, t, site=None):
        cls._cur[(dev, site)] = t


def _fp32_g_mode():
    """DPRHOT_FP32_G=1: debug mode of SURVEY.md section 8 c5 -- dScores no longer travel as ONE bf16 value (2^-9 relative
    rounding, the source of the ~2e-3 gradient deviation) but as a bf16 pair hi + lo (G = hi + lo to 2^-17), and each
    backward GEMM runs twice on the same MFMA kernels; gradients then agree with the fp32 reference to <= 1e-3.  Single
    rank only; costs a second pass over the logits and two more GEMM launches."""
    return os.environ.get("DPRHOT_FP32_G") == "1"


def _dc_wire_dtype(group=None):
    """The format the dC partials travel in.  DPRHOT_DC_WIRE=bf16: the reduce-scatter ships bf16 (10.5 MiB per rank at cfg3 instead of
    21 MiB; one more rounding per partial); =fp32: never.  Unset / auto: what dist.choose_path_collectives
    MEASURED for this group when it was given both wires (DenseRetrieverTask.on_pretrain_routine_start; bf16 only where it won by 5 %),
    fp32 where no probe ran."""
    env = os.environ.get("DPRHOT_DC_WIRE", "auto")
    if env == "bf16":
        return _BF16
    if env == "fp32" or group is False:
        return torch.float32
    return D.path_wire(group, torch.float32)


_ZERO_ROW = {}  # device -> one fp32 zero, never written: `expand`ed it is a gradient of zeros that costs no launch


def _zeros_view(shape, device):
    z = _ZERO_ROW.get(device)
    if z is None:
        z = _ZERO_ROW[device] = torch.zeros((1, 1), dtype=torch.float32, device=device)
    return z.expand(*shape)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class HipKernels:
    """Thin, allocation-aware wrapper over libdprhot.  Every tensor must live on a HIP device."""

    name = "hip"

    def __init__(self):
        from . import _lib  # raises ImportError if libdprhot.so is missing -- loudly, by design

        self._lib = _lib
        self.lib = _lib.lib
        self._ws = {}
        self._nslabs = {}
        self._wg = {}

    # -- plumbing --------------------------------------------------------------------------------------
    @staticmethod
    def _require_gpu(*ts):
        for t in ts:
            if t is not None and not t.is_cuda:
                raise RuntimeError(
                    "dpr_scale_amd hot path needs HIP device tensors (got a CPU tensor); there is no CPU fallback")

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _workspace(self, device, nbytes):
        ws = self._ws.get(device)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
            self._ws[device] = ws
        return ws

    def empty(self, shape, dtype, like):
        return torch.empty(shape, dtype=dtype, device=like.device)

    def _g_buffer(self, want_G, B, Nc, d, dev):
        """The dScores buffer of the one-call steps: True -> always (the plan then materialises G); "auto" -> only where the shape's
        plan needs one (dprhot_step_wants_g; elsewhere the step never writes the dScores: one launch less at cfg3 per rank);
        False -> none (raises at shapes that need it)."""
        if want_G is True or (want_G == "auto" and self._wants_g(B, Nc, d)):
            return torch.empty((B, Nc), dtype=_BF16, device=dev)
        return None

    def _wants_g(self, B, Nc, d):
        key = (B, Nc, d, self._lib.options_epoch())
        w = self._wg.get(key)
        if w is None:
            w = self._wg[key] = self._lib.step_wants_g(B, Nc, d)
        return w

    # -- ops -------------------------------------------------------------------------------------------
    def cast_bf16(self, src, dst):
        """dst (bf16, contiguous, may be a row-slice of the gather buffer) <- src (fp32 or bf16)."""
        self._require_gpu(src, dst)
        src = src.detach()
        if src.dtype == _BF16:
            dst.copy_(src)
            return dst
        src = src.contiguous().float()
        n = src.numel()
        assert dst.is_contiguous() and dst.numel() == n and n % 8 == 0
        self._lib.check(self.lib.dprhot_cast_bf16(_ptr(src), _ptr(dst), n, self._stream()), "dprhot_cast_bf16")
        return dst

    def prep(self, q, Qb, c, Cdst):
        """Both producer-side casts in one launch (fp32 inputs); other dtypes take the per-tensor path."""
        self._require_gpu(q, Qb, c, Cdst)
        if q.dtype != torch.float32 or c.dtype != torch.float32:
            self.cast_bf16(q, Qb)
            self.cast_bf16(c, Cdst)
            return
        q = q.detach().contiguous()
        c = c.detach().contiguous()
        assert Qb.is_contiguous() and Cdst.is_contiguous() and Qb.numel() == q.numel() and Cdst.numel() == c.numel()
        self._lib.check(self.lib.dprhot_prep(_ptr(q), q.numel(), _ptr(Qb), _ptr(c), c.numel(), _ptr(Cdst), self._stream()),
                        "dprhot_prep")

    def packed_rows(self, n_ctx, d):
        return self._lib.packed_rows(n_ctx, d)

    def pack_ctx(self, c, m8, send):
        """send [rows_c, d] bf16 <- context rows (fp32 -> bf16) followed by the mask bytes (one all-gather moves both)."""
        self._require_gpu(c, m8, send)
        c = c.detach().float().contiguous()
        n_ctx, d = c.shape
        assert send.is_contiguous() and send.shape == (self.packed_rows(n_ctx, d), d)
        self._lib.check(self.lib.dprhot_pack_ctx(_ptr(c), _ptr(m8.contiguous()), n_ctx, d, _ptr(send), self._stream()),
                        "dprhot_pack_ctx")

    def unpack_mask(self, gathered, W, n_ctx, colmask):
        self._require_gpu(gathered, colmask)
        d = gathered.shape[1]
        assert colmask.numel() == W * self.packed_rows(n_ctx, d)
        self._lib.check(self.lib.dprhot_unpack_mask(_ptr(gathered), W, n_ctx, d, _ptr(colmask), self._stream()),
                        "dprhot_unpack_mask")

    def inbatch_fwd(self, Qb, Cb, y, y_offset, colmask, inv_T, grad_scale, want_logits=False, want_G=True):
        self._require_gpu(Qb, Cb, y, colmask)
        B, d = Qb.shape
        Nc = Cb.shape[0]
        dev = Qb.device
        row_loss = torch.empty(B, dtype=torch.float32, device=dev)
        row_lse = torch.empty(B, dtype=torch.float32, device=dev)
        loss_sum = torch.empty(1, dtype=torch.float32, device=dev)
        G = torch.empty((B, Nc), dtype=_BF16, device=dev) if want_G else None
        S = torch.empty((B, Nc), dtype=torch.float32, device=dev) if want_logits else None
        nbytes = self._lib.workspace_bytes(B, Nc, d)
        ws = self._workspace(dev, nbytes)
        self._lib.check(self.lib.dprhot_inbatch_fwd(
            _ptr(Qb), B, _ptr(Cb), Nc, d, _ptr(y), int(y_offset), _ptr(colmask), float(inv_T), float(grad_scale),
            _ptr(S), _ptr(row_loss), _ptr(row_lse), _ptr(loss_sum), _ptr(G), _ptr(ws), ws.numel(), self._stream()),
            "dprhot_inbatch_fwd")
        return row_loss, row_lse, loss_sum, G, S

    def inbatch_fwd_f32(self, q, c, Qb, Cb, y, y_offset, colmask, inv_T, grad_scale, want_logits=False, want_G=True):
        """Forward straight from the fp32 encoder outputs: q [B,d] fp32; c [Nc,d] fp32 or None (Cb already holds
        the gathered bf16 rows).  Fills Qb (and Cb when c is given) with the bf16 images the backward reads."""
        self._require_gpu(q, c, Qb, Cb, y, colmask)
        B, d = Qb.shape
        Nc = Cb.shape[0]
        dev = Qb.device
        q = q.detach().contiguous()
        c = c.detach().contiguous() if c is not None else None
        assert q.dtype == torch.float32 and (c is None or (c.dtype == torch.float32 and c.shape[0] == Nc))
        row_loss = torch.empty(B, dtype=torch.float32, device=dev)
        row_lse = torch.empty(B, dtype=torch.float32, device=dev)
        loss_sum = torch.empty(1, dtype=torch.float32, device=dev)
        G = torch.empty((B, Nc), dtype=_BF16, device=dev) if want_G else None
        S = torch.empty((B, Nc), dtype=torch.float32, device=dev) if want_logits else None
        ws = self._workspace(dev, self._lib.workspace_bytes(B, Nc, d))
        self._lib.check(self.lib.dprhot_inbatch_fwd_f32(
            _ptr(q), _ptr(c), _ptr(Qb), _ptr(Cb), B, Nc, d, _ptr(y), int(y_offset), _ptr(colmask), float(inv_T),
            float(grad_scale), _ptr(S), _ptr(row_loss), _ptr(row_lse), _ptr(loss_sum), _ptr(G), _ptr(ws), ws.numel(),
            self._stream()), "dprhot_inbatch_fwd_f32")
        return row_loss, row_lse, loss_sum, G, S

    def inbatch_step_f32(self, q, c, Qb, Cb, y, y_offset, colmask, inv_T, grad_scale, want_G=True):
        """Forward AND backward in one library call (dprhot_inbatch_step_f32; two launches at the BASELINE training
        shapes).  Gradients come back for grad_output = 1: returns (row_loss, row_lse, loss_sum, G, dQ, dC_part)."""
        self._require_gpu(q, c, Qb, Cb, y, colmask)
        B, d = Qb.shape
        Nc = Cb.shape[0]
        dev = Qb.device
        q = q.detach().contiguous()
        c = c.detach().contiguous() if c is not None else None
        assert q.dtype == torch.float32 and (c is None or (c.dtype == torch.float32 and c.shape[0] == Nc))
        f32 = torch.float32
        row_loss = torch.empty(B, dtype=f32, device=dev)
        row_lse = torch.empty(B, dtype=f32, device=dev)
        loss_sum = torch.empty(1, dtype=f32, device=dev)
        G = self._g_buffer(want_G, B, Nc, d, dev)
        dQ = torch.empty((B, d), dtype=f32, device=dev)
        dC = torch.empty((Nc, d), dtype=f32, device=dev)
        ws = self._workspace(dev, self._lib.workspace_bytes(B, Nc, d))
        self._lib.check(self.lib.dprhot_inbatch_step_f32(
            _ptr(q), _ptr(c), _ptr(Qb), _ptr(Cb), B, Nc, d, _ptr(y), int(y_offset), _ptr(colmask), float(inv_T),
            float(grad_scale), 1.0, None, None, _ptr(row_loss), _ptr(row_lse), _ptr(loss_sum), _ptr(G), _ptr(dQ), _ptr(dC),
            _ptr(ws), ws.numel(), self._stream()), "dprhot_inbatch_step_f32")
        return row_loss, row_lse, loss_sum, G, dQ, dC

    def inbatch_step_packed_f32(self, q, gathered, Qb, W, rank, n_ctx, y, inv_T, grad_scale, want_G=True):
        """World size > 1: everything between the all-gather and the reduce-scatter in one library call.  `gathered` is
        the all-gathered packed buffer; the mask is read from it, and dC_part carries this rank's loss numerator at
        [k * rows_c + n_ctx][0] of every chunk k (see include/dprhot.h).  Returns (row_loss, row_lse, loss_sum, G, dQ,
        dC_part)."""
        self._require_gpu(q, gathered, Qb, y)
        B, d = Qb.shape
        Nc = gathered.shape[0]
        dev = Qb.device
        q = q.detach().contiguous()
        assert q.dtype == torch.float32 and gathered.dtype == _BF16 and Nc == W * self.packed_rows(n_ctx, d)
        f32 = torch.float32
        row_loss = torch.empty(B, dtype=f32, device=dev)
        row_lse = torch.empty(B, dtype=f32, device=dev)
        loss_sum = torch.empty(1, dtype=f32, device=dev)
        G = self._g_buffer(want_G, B, Nc, d, dev)
        dQ = torch.empty((B, d), dtype=f32, device=dev)
        dC = torch.empty((Nc, d), dtype=f32, device=dev)
        ws = self._workspace(dev, self._lib.workspace_bytes(B, Nc, d))
        self._lib.check(self.lib.dprhot_inbatch_step_packed_f32(
            _ptr(q), _ptr(gathered), _ptr(Qb), B, int(W), int(rank), int(n_ctx), d, _ptr(y), float(inv_T), float(grad_scale),
            1.0, None, _ptr(row_loss), _ptr(row_lse), _ptr(loss_sum), _ptr(G), _ptr(dQ), _ptr(dC), _ptr(ws), ws.numel(),
            self._stream()), "dprhot_inbatch_step_packed_f32")
        return row_loss, row_lse, loss_sum, G, dQ, dC

    def _dq_slabs(self, B, Nc, d, dev):
        """Caller-owned buffer for the split-K partial sums of dQ, or None when this shape's plan leaves none."""
        key = (B, Nc, d, self._lib.options_epoch())  # the plan depends on process-wide options (dprhot_set_option)
        n = self._nslabs.get(key)
        if n is None:
            n = self._nslabs[key] = self._lib.train_dq_slabs(B, Nc, d)
        return torch.empty((n, B, d), dtype=torch.float32, device=dev) if n > 0 else None

    def train_step_f32(self, q, c, Qb, Cb, y, y_offset, colmask, inv_T, grad_scale, loss_scale, d_scale, dc_dtype=torch.float32,
                       defer_dq=False, want_G=True):
        """The operator's step (dprhot_train_step_f32): forward AND backward in one library call, the loss already multiplied by
        loss_scale, the gradients scaled by the DEVICE scalar d_scale (the grad_output backward() is expected to deliver).
        Returns (row_loss, row_lse, loss_out [2], G, dQ, dC_part); loss_out[0] is the loss.  defer_dq: where the plan splits dQ over
        the contexts, dQ comes back as (dQ buffer, slabs) and rescale_grads() adds the slabs up (one launch less in the step)."""
        self._require_gpu(q, c, Qb, Cb, y, colmask, d_scale)
        B, d = Qb.shape
        Nc = Cb.shape[0]
        dev = Qb.device
        q = q.detach().contiguous()
        c = c.detach().contiguous() if c is not None else None
        assert q.dtype == torch.float32 and (c is None or (c.dtype == torch.float32 and c.shape[0] == Nc))
        f32 = torch.float32
        row_loss = torch.empty(B, dtype=f32, device=dev)
        row_lse = torch.empty(B, dtype=f32, device=dev)
        loss_out = torch.empty(2, dtype=f32, device=dev)
        G = self._g_buffer(want_G, B, Nc, d, dev)
        dQ = torch.empty((B, d), dtype=f32, device=dev)
        dC = torch.empty((Nc, d), dtype=dc_dtype, device=dev)
        part = self._dq_slabs(B, Nc, d, dev) if defer_dq else None
        ws = self._workspace(dev, self._lib.workspace_bytes(B, Nc, d))
        self._lib.check(self.lib.dprhot_train_step_f32(
            _ptr(q), _ptr(c), _ptr(Qb), _ptr(Cb), B, Nc, d, _ptr(y), int(y_offset), _ptr(colmask), float(inv_T), float(grad_scale),
            float(loss_scale), _ptr(d_scale), _ptr(row_loss), _ptr(row_lse), _ptr(loss_out), _ptr(G), _ptr(dQ), _ptr(part), _ptr(dC),
            _KIND[dc_dtype], _ptr(ws), ws.numel(), self._stream()), "dprhot_train_step_f32")
        return row_loss, row_lse, loss_out, G, (dQ if part is None else (dQ, part)), dC

    def train_step_packed_f32(self, q, gathered, Qb, W, rank, n_ctx, y, inv_T, grad_scale, loss_scale, d_scale, dc_dtype=torch.float32,
                              defer_dq=False, want_G=True):
        """World size > 1 (dprhot_train_step_packed_f32): as train_step_f32 on the all-gathered packed buffer."""
        self._require_gpu(q, gathered, Qb, y, d_scale)
        B, d = Qb.shape
        Nc = gathered.shape[0]
        dev = Qb.device
        q = q.detach().contiguous()
        assert q.dtype == torch.float32 and gathered.dtype == _BF16 and Nc == W * self.packed_rows(n_ctx, d)
        f32 = torch.float32
        row_loss = torch.empty(B, dtype=f32, device=dev)
        row_lse = torch.empty(B, dtype=f32, device=dev)
        loss_out = torch.empty(2, dtype=f32, device=dev)
        G = self._g_buffer(want_G, B, Nc, d, dev)
        dQ = torch.empty((B, d), dtype=f32, device=dev)
        dC = torch.empty((Nc, d), dtype=dc_dtype, device=dev)
        part = self._dq_slabs(B, Nc, d, dev) if defer_dq else None
        ws = self._workspace(dev, self._lib.workspace_bytes(B, Nc, d))
        self._lib.check(self.lib.dprhot_train_step_packed_f32(
            _ptr(q), _ptr(gathered), _ptr(Qb), B, int(W), int(rank), int(n_ctx), d, _ptr(y), float(inv_T), float(grad_scale),
            float(loss_scale), _ptr(d_scale), _ptr(row_loss), _ptr(row_lse), _ptr(loss_out), _ptr(G), _ptr(dQ), _ptr(part), _ptr(dC),
            _KIND[dc_dtype], _ptr(ws), ws.numel(), self._stream()), "dprhot_train_step_packed_f32")
        return row_loss, row_lse, loss_out, G, (dQ if part is None else (dQ, part)), dC

    def rescale_grads(self, dQ, dC, go, used):
        """backward() of the operator (dprhot_rescale_grads): the gradients were computed for grad_output = used; multiply by
        go / used only if they differ (decided on the device).  Returns out2: [0] what the gradients are scaled by now, [1] the
        scale the next forward should expect."""
        part = None
        if isinstance(dQ, tuple):  # (dQ buffer, split-K slabs left by a train step with defer_dq): summed here, scaled by go
            dQ, part = dQ
        self._require_gpu(dQ, part, dC, go, used)
        out2 = torch.empty(2, dtype=torch.float32, device=go.device)
        self._lib.check(self.lib.dprhot_rescale_grads(
            _ptr(dQ), dQ.numel() if dQ is not None else 0, _ptr(part), part.shape[0] if part is not None else 0, _ptr(dC),
            dC.numel() if dC is not None else 0, _KIND[dC.dtype] if dC is not None else 2, _ptr(go), _ptr(used), _ptr(out2),
            self._stream()), "dprhot_rescale_grads")
        return out2

    def widen(self, src, dst):
        """dst (fp32, contiguous) <- src (bf16 / fp16 / fp32, contiguous), one pass (dprhot_grad_unpack)."""
        self._require_gpu(src, dst)
        assert src.is_contiguous() and dst.is_contiguous() and dst.dtype == torch.float32 and src.numel() >= dst.numel()
        self._lib.check(self.lib.dprhot_grad_unpack(_ptr(src), _KIND[src.dtype], _ptr(dst), dst.numel(), self._stream()), "dprhot_grad_unpack")
        return dst

    def inbatch_bwd(self, G, Qb, Cb, h_scale, d_scale, need_dq=True, need_dc=True):
        self._require_gpu(G, Qb, Cb, d_scale)
        B, d = Qb.shape
        Nc = Cb.shape[0]
        dev = Qb.device
        dQ = torch.empty((B, d), dtype=torch.float32, device=dev) if need_dq else None
        dC = torch.empty((Nc, d), dtype=torch.float32, device=dev) if need_dc else None
        ws = self._workspace(dev, self._lib.workspace_bytes(B, Nc, d))
        self._lib.check(self.lib.dprhot_inbatch_bwd(
            _ptr(G), _ptr(Qb), _ptr(Cb), B, Nc, d, float(h_scale), _ptr(d_scale), _ptr(dQ), _ptr(dC), _ptr(ws),
            ws.numel(), self._stream()), "dprhot_inbatch_bwd")
        return dQ, dC

    def sim(self, Qb, Cb, colmask=None, inv_T=1.0):
        self._require_gpu(Qb, Cb, colmask)
        B, d = Qb.shape
        Nc = Cb.shape[0]
        S = torch.empty((B, Nc), dtype=torch.float32, device=Qb.device)
        self._lib.check(self.lib.dprhot_sim_fwd(_ptr(Qb), B, _ptr(Cb), Nc, d, _ptr(colmask), float(inv_T), _ptr(S),
                                                self._stream()), "dprhot_sim_fwd")
        return S

    def softmax_ce(self, S, y, y_offset=0, grad_scale=1.0, want_G=False, row_win_start=None, win_len=0):
        self._require_gpu(S, y, row_win_start)
        B, Nc = S.shape
        dev = S.device
        row_loss = torch.empty(B, dtype=torch.float32, device=dev)
        row_lse = torch.empty(B, dtype=torch.float32, device=dev)
        G = torch.empty((B, Nc), dtype=_BF16, device=dev) if want_G else None
        self._lib.check(self.lib.dprhot_softmax_ce_fwd_bwd(
            _ptr(S), B, Nc, _ptr(y), int(y_offset), float(grad_scale), _ptr(row_win_start), int(win_len),
            _ptr(row_loss), _ptr(row_lse), _ptr(G), self._stream()), "dprhot_softmax_ce_fwd_bwd")
        return row_loss, row_lse, G

    def reduce_sum(self, x, scale=1.0):
        self._require_gpu(x)
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        self._lib.check(self.lib.dprhot_reduce_sum(_ptr(x), x.numel(), float(scale), _ptr(out), self._stream()),
                        "dprhot_reduce_sum")
        return out

    def dq(self, G, Cb, h_scale=1.0, d_scale=None):
        B, Nc = G.shape
        d = Cb.shape[1]
        out = torch.empty((B, d), dtype=torch.float32, device=G.device)
        ws = self._workspace(G.device, self._lib.workspace_bytes(B, Nc, d))
        self._lib.check(self.lib.dprhot_dq(_ptr(G), _ptr(Cb), B, Nc, d, float(h_scale), _ptr(d_scale), _ptr(out), _ptr(ws),
                                           ws.numel(), self._stream()), "dprhot_dq")
        return out

    def dc(self, G, Qb, h_scale=1.0, d_scale=None):
        B, Nc = G.shape
        d = Qb.shape[1]
        out = torch.empty((Nc, d), dtype=torch.float32, device=G.device)
        self._lib.check(self.lib.dprhot_dc(_ptr(G), _ptr(Qb), B, Nc, d, float(h_scale), _ptr(d_scale), _ptr(out),
                                           self._stream()), "dprhot_dc")
        return out

    def pairwise_fwd(self, q, c, m8):
        self._require_gpu(q, c, m8)
        B, d = q.shape
        M = c.shape[0] // B
        S = torch.empty((B, M), dtype=torch.float32, device=q.device)
        self._lib.check(self.lib.dprhot_pairwise_fwd(_ptr(q), _ptr(c), _ptr(m8), B, M, d, _ptr(S), self._stream()), "dprhot_pairwise_fwd")
        return S

    def pairwise_bwd(self, g, q, c, need_dq=True, need_dc=True):
        self._require_gpu(g, q, c)
        B, d = q.shape
        M = c.shape[0] // B
        dq = torch.empty_like(q) if need_dq else None
        dc = torch.empty_like(c) if need_dc else None
        self._lib.check(self.lib.dprhot_pairwise_bwd(_ptr(g), _ptr(q), _ptr(c), B, M, d, _ptr(dq), _ptr(dc), self._stream()),
                        "dprhot_pairwise_bwd")
        return dq, dc

    def rank_of_gold(self, S, y, y_offset=0):
        self._require_gpu(S, y)
        rows, cols = S.shape
        out = torch.empty(rows, dtype=torch.int64, device=S.device)
        self._lib.check(self.lib.dprhot_rank_of_gold(_ptr(S), rows, cols, _ptr(y), int(y_offset), _ptr(out), self._stream()),
                        "dprhot_rank_of_gold")
        return out

    def sim_rank(self, Qb, Cb, y, colmask=None, inv_T=1.0, y_offset=0):
        """rank_of_gold of the scores Qb x Cb^T * inv_T (masked columns -inf) without the score matrix at large shapes."""
        self._require_gpu(Qb, Cb, y, colmask)
        B, d = Qb.shape
        Nc = Cb.shape[0]
        rank = torch.empty(B, dtype=torch.int64, device=Qb.device)
        ws = self._workspace(Qb.device, self._lib.workspace_bytes(B, Nc, d))
        self._lib.check(self.lib.dprhot_sim_rank(_ptr(Qb), B, _ptr(Cb), Nc, d, _ptr(y), int(y_offset), _ptr(colmask), float(inv_T),
                                                 _ptr(rank), _ptr(ws), ws.numel(), self._stream()), "dprhot_sim_rank")
        return rank

    def sim_rank_loss(self, Qb, Cb, y, colmask=None, inv_T=1.0, y_offset=0):
        """(rank [B] int64, loss_sum [1]) of the scores Qb x Cb^T * inv_T -- dprhot_sim_rank_loss: one GEMM pass at large shapes."""
        self._require_gpu(Qb, Cb, y, colmask)
        B, d = Qb.shape
        Nc = Cb.shape[0]
        rank = torch.empty(B, dtype=torch.int64, device=Qb.device)
        loss_sum = torch.empty(1, dtype=torch.float32, device=Qb.device)
        ws = self._workspace(Qb.device, self._lib.workspace_bytes(B, Nc, d))
        self._lib.check(self.lib.dprhot_sim_rank_loss(_ptr(Qb), B, _ptr(Cb), Nc, d, _ptr(y), int(y_offset), _ptr(colmask), float(inv_T),
                                                      _ptr(rank), None, None, _ptr(loss_sum), _ptr(ws), ws.numel(), self._stream()),
                        "dprhot_sim_rank_loss")
        return rank, loss_sum

    def topk(self, S, k):
        self._require_gpu(S)
        rows, cols = S.shape
        v = torch.empty((rows, k), dtype=torch.float32, device=S.device)
        i = torch.empty((rows, k), dtype=torch.int64, device=S.device)
        self._lib.check(self.lib.dprhot_topk(_ptr(S), rows, cols, k, _ptr(v), _ptr(i), self._stream()), "dprhot_topk")
        return v, i

    def topk_update(self, S, cols, col_offset, values, indices, first):
        self._require_gpu(S, values, indices)
        rows, k = values.shape
        self._lib.check(self.lib.dprhot_topk_update(_ptr(S), rows, int(cols), S.stride(0), int(col_offset), k, _ptr(values),
                                                    _ptr(indices), int(bool(first)), self._stream()), "dprhot_topk_update")

    def topk_update_wide(self, S, cols, col_offset, values, indices, first, ws):
        """Any k (csrc/wideselect.h): state in HBM.  ws: uint8 workspace of dprhot_topk_wide_workspace_bytes(rows, k) bytes.  No host sync:
        the ABI reserves word 5 of every row's record as an error word ("keeps its old state"), but no such condition is defined (always 0,
        include/dprhot.h) -- CorpusSearch.result() checks the words of the LAST call once instead of syncing after every chunk."""
        self._require_gpu(S, values, indices, ws)
        rows, k = values.shape
        self._lib.check(self.lib.dprhot_topk_update_wide(_ptr(S), rows, int(cols), S.stride(0), int(col_offset), k, _ptr(values), _ptr(indices),
                                                         1 if first else 0, _ptr(ws), ws.numel(), self._stream()), "dprhot_topk_update_wide")

    @staticmethod
    def topk_wide_errors(ws, rows):
        """The error words of the last dprhot_topk_update_wide call on `ws` (one int32 per row; host sync when inspected)."""
        return ws[: rows * 32].view(torch.int32).view(rows, 8)[:, 5]

    def topk_wide_workspace(self, rows, k, like):
        n = ctypes.c_size_t(0)
        self._lib.check(self.lib.dprhot_topk_wide_workspace_bytes(int(rows), int(k), ctypes.byref(n)), "dprhot_topk_wide_workspace_bytes")
        return torch.empty(n.value, dtype=torch.uint8, device=like.device)

    def search_workspace(self, nq, chunk, like):
        return torch.empty(self._lib.search_workspace_bytes(nq, chunk), dtype=torch.uint8, device=like.device)

    def search(self, Qb, Cb, id_offset, values, indices, first, chunk, ws):
        self._require_gpu(Qb, Cb, values, indices, ws)
        nq, d = Qb.shape
        k = values.shape[1]
        self._lib.check(self.lib.dprhot_search(_ptr(Qb), nq, _ptr(Cb), Cb.shape[0], d, int(id_offset), k, int(chunk),
                                               _ptr(values), _ptr(indices), int(bool(first)), _ptr(ws), ws.numel(),
                                               self._stream()), "dprhot_search")


_DEFAULT = None


def _load_opx():
    """csrc/opx.cpp, built next to libdprhot.so by `make -C dpr_scale_amd/csrc` / __graft_entry__.build(); DPRHOT_OPX=0 keeps the
    Python host path (A/B).  Optional: without it the operator is the same operator, ~150 us per step slower on the host."""
    if os.environ.get("DPRHOT_OPX", "1") == "0":
        return None
    try:
        from . import _lib, _opx
    except ImportError:
        return None
    _opx.init(_lib.LIB_PATH)
    return _opx


try:
    _OPX = _load_opx()
except Exception:  # pragma: no cover - a broken optional extension must not take the operator down
    _OPX = None


# the multi-rank packed step's host side in C++ (csrc/opx.cpp: packed_train_step / packed_backward); DPRHOT_OPX_PACKED=0: the Python wrappers
_OPX_PACKED = _OPX is not None and hasattr(_OPX, "packed_train_step") and os.environ.get("DPRHOT_OPX_PACKED", "1") != "0"


def default_kernels():
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = HipKernels()
    return _DEFAULT


def _pad_cols(n):
    return (n + 7) // 8 * 8


class ContextGather:
    """The forward collective, started as soon as the context tower has produced its rows:
    pack (bf16 rows + mask bytes) on the compute stream, then ONE all-gather with async_op=True -- it runs on RCCL's
    own stream while the query tower keeps the compute stream busy.  InBatchContrastive waits on it right before
    the sim GEMM.  Usage (what DenseRetrieverTask.training_step does):

        c  = encode_contexts(...)
        c, pending = defer_context_grad(c)            # see below: lets the reduce-scatter overlap the query-tower backward
        g  = ContextGather(c, ctx_mask, group)        # all-gather in flight ...
        q  = encode_queries(...)                      # ... under the query tower
        loss = inbatch_contrastive_loss(q, c, pos, ctx_mask, T, group, gather=g, pending=pending)
    """

    def __init__(self, c, ctx_mask, group=None, kernels=None):
        kn = kernels if kernels is not None else default_kernels()
        W, r = D.world(group)
        assert W > 1 or D.force_dist(), "ContextGather is for world size > 1"
        n_ctx, d = c.shape
        m8 = ctx_mask.view(torch.uint8) if ctx_mask.dtype == torch.bool else ctx_mask.to(torch.uint8)
        self.rows_c = kn.packed_rows(n_ctx, d)
        self.send = kn.empty((self.rows_c, d), _BF16, c)
        kn.pack_ctx(c.detach(), m8, self.send)
        self.Cb = kn.empty((W * self.rows_c, d), _BF16, c)
        self.work = D.all_gather_rows(self.send, self.Cb, group, async_op=True)
        self.shape = (n_ctx, d)

    def wait(self):
        if self.work is not None:
            self.work.wait()  # the compute stream waits for RCCL's stream; the host does not block
            self.work = None
        return self.Cb


class PendingGrad:
    """Carries the reduce-scatter started in InBatchContrastive.backward to the point where its result is consumed."""

    def __init__(self):
        self.work = None
        self.post = None  # run on the gradient after the wait (half-width wire: widen the received rows into it)


class _DeferContextGrad(torch.autograd.Function):
    """Identity on the context rows whose backward first waits for the pending reduce-scatter.  Placed right after
    the context tower and BEFORE the query tower runs, its autograd node is older than every query-tower node, so
    the engine runs the whole query-tower backward (it only needs dq) before it gets here: the collective overlaps
    it on RCCL's stream."""

    @staticmethod
    def forward(ctx, c, pending):
        ctx.pending = pending
        return c.view_as(c)

    @staticmethod
    def backward(ctx, grad):
        w, ctx.pending.work = ctx.pending.work, None
        if w is not None:
            w.wait()
        post, ctx.pending.post = ctx.pending.post, None
        if post is not None:
            out = post(grad)  # (may hand on a different tensor: the half-width wire's widened rows)
            grad = out if out is not None else grad
        return grad, None


def defer_context_grad(c):
    pending = PendingGrad()
    return _DeferContextGrad.apply(c, pending), pending


class InBatchContrastive(torch.autograd.Function):
    """loss = InBatchContrastive.apply(q_local, c_local, pos_idx, ctx_mask, temperature, group, kernels)

    q_local [B,d], c_local [B*K,d] (fp32 or bf16, requires_grad), pos_idx [B] int64 rank-local positive
    indices (dpr_transform.py:164-166), ctx_mask [B*K] bool (dummy-context mask, dpr_transform.py:143-157).
    Returns the scalar fp32 loss, identical on every rank (mean over the W*B global queries), exactly the
    value dpr_task.py:212 returns.  backward returns (dq_local, dc_local): the same tensors the reference's
    autograd leaves in q.grad / c.grad on this rank.
    """

    @staticmethod
    def forward(ctx, q, c, pos_idx, ctx_mask, temperature, group, kernels, gather=None, pending=None):
        ctx.fast = None
        ctx.opx_packed = False
        if _OPX is not None and kernels is None and gather is None and q.is_cuda and c.is_cuda and (group is False or D.world(group)[0] == 1):
            # The step a single-GPU run issues every iteration (dpr_task.py:197-212): host side in C++ (csrc/opx.cpp).  Taken when
            # nothing needs the general code below: fp32 contiguous encoder outputs, a whole number of 8-column groups, a byte mask.
            n_ctx, d = c.shape
            if (q.dtype == torch.float32 and c.dtype == torch.float32 and q.is_contiguous() and c.is_contiguous() and n_ctx % 8 == 0 and d % 8 == 0
                    and q.shape[1] == d and ctx_mask.is_contiguous() and ctx_mask.element_size() == 1 and pos_idx.dtype == torch.int64
                    and pos_idx.is_contiguous() and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and not _fp32_g_mode()
                    and not (D.force_dist() and group is not False)):
                B = q.shape[0]
                inv_T = 1.0 / float(temperature)
                site = (B, n_ctx, d)
                used = _ExpectedGradScale.get(q.device, site)
                loss_out, row_lse, dQ, dC, Qb, Cb, G, part = _OPX.train_step(q, c, pos_idx, ctx_mask, inv_T, inv_T / B, 1.0 / B, used,
                                                                             default_kernels()._lib.options_epoch())
                ctx.fast = (dQ, dC, part, used, site)
                ctx.regen = (Qb, Cb, G, pos_idx, ctx_mask, inv_T, inv_T / B)
                ctx.row_lse = row_lse
                return loss_out[0]
        kn = kernels if kernels is not None else default_kernels()
        W, r = (1, 0) if group is False else D.world(group)  # group=False: single-device strategy, never gather
        B, d = q.shape
        n_ctx = c.shape[0]  # contexts on this rank (B*K)
        Nq = W * B
        assert pos_idx.shape[0] == B and ctx_mask.shape[0] == n_ctx
        if d % 8 != 0:
            raise ValueError(f"hidden size {d} must be a multiple of 8")
        m8 = ctx_mask.view(torch.uint8) if ctx_mask.dtype == torch.bool else ctx_mask.to(torch.uint8)
        q_f32 = q.dtype == torch.float32 and hasattr(kn, "inbatch_fwd_f32")
        Qb = kn.empty((B, d), _BF16, q)

        wants_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        multi = W > 1 or (group is not False and D.force_dist())  # (forced: the multi-rank code path on a one-rank world, for timelines)
        if not multi:
            # columns = this rank's contexts (padded to a multiple of 8 with masked zero rows)
            rows_c = _pad_cols(n_ctx)
            Nc = rows_c
            Cb = kn.empty((Nc, d), _BF16, c)
            if Nc == n_ctx:
                colmask = m8 if m8.is_contiguous() else m8.contiguous()  # the batch's own bool mask, read in place
            else:
                colmask = kn.empty((Nc,), torch.uint8, c)
                colmask[:n_ctx].copy_(m8)
                Cb[n_ctx:].zero_()
                colmask[n_ctx:].fill_(1)
            # fp32 encoder outputs are consumed as they are by the sim kernel (it rounds to bf16 while staging
            # and leaves the bf16 images in Qb / Cb)
            c_direct = q_f32 and c.dtype == torch.float32 and Nc == n_ctx
            if not c_direct:
                kn.prep(q, Qb, c, Cb[:n_ctx])
                q_f32 = False
        else:
            # ONE all-gather: context rows (bf16) + mask bytes in trailing rows of the same buffer; the trailing
            # rows become always-masked extra columns (the reference issues 4 fp32 all_gathers, :169-176)
            rows_c = kn.packed_rows(n_ctx, d)
            Nc = W * rows_c
            if gather is not None:  # started under the query tower (ContextGather): only wait here
                assert gather.shape == (n_ctx, d)
                Cb = gather.wait()
            else:
                send = kn.empty((rows_c, d), _BF16, c)
                kn.pack_ctx(c, m8, send)
                Cb = kn.empty((Nc, d), _BF16, c)
                D.all_gather_rows(send, Cb, group)
            c_direct = False
            # with a backward to follow, the one-call step reads the mask straight from the gathered buffer: no unpack launch
            packed_step = q_f32 and wants_grad and (hasattr(kn, "train_step_packed_f32") or hasattr(kn, "inbatch_step_packed_f32"))
            if not packed_step:
                colmask = kn.empty((Nc,), torch.uint8, c)
                kn.unpack_mask(Cb, W, n_ctx, colmask)
            if not q_f32:
                kn.cast_bf16(q, Qb)

        inv_T = 1.0 / float(temperature)
        grad_scale = inv_T / Nq  # d loss / d S of the global mean, before grad_output
        y_off = r * rows_c       # dpr_task.py:189-190 (label offset of this rank's columns)
        eager = None  # gradients computed in this call, scaled by ctx.used (a device scalar)
        used = None
        S_dbg = None
        loss_is_mean = False  # the kernel already multiplied the numerator by 1 / Nq
        dc_dtype = _dc_wire_dtype(group) if multi else torch.float32
        if not multi and _fp32_g_mode() and wants_grad:
            if q_f32:
                row_loss, row_lse, loss_sum, G, S_dbg = kn.inbatch_fwd_f32(q, c if c_direct else None, Qb, Cb, pos_idx, y_off, colmask,
                                                                           inv_T, grad_scale, want_logits=True)
            else:
                row_loss, row_lse, loss_sum, G, S_dbg = kn.inbatch_fwd(Qb, Cb, pos_idx, y_off, colmask, inv_T, grad_scale, want_logits=True)
        elif multi and packed_step and hasattr(kn, "train_step_packed_f32"):
            # forward and backward of this rank's rows in ONE library call; gradients scaled by the grad_output the last backward saw
            used = _ExpectedGradScale.get(q.device, (B, Nc, d))
            if _OPX_PACKED and kernels is None and q.is_contiguous() and pos_idx.dtype == torch.int64 and pos_idx.is_contiguous() \
                    and dc_dtype in (torch.float32, _BF16):
                # host side in C++ (csrc/opx.cpp: packed_train_step): the allocations, the library call and its fall-back to fp32
                # partials in one function; backward's counterpart is packed_backward
                loss_sum, row_lse, dQ, dC_part, Qb, G, part = _OPX.packed_train_step(q.detach(), Cb, pos_idx, W, r, n_ctx, inv_T, grad_scale, 1.0 / Nq,
                                                                                     used, 0 if dc_dtype == _BF16 else 2)
                G = G if (G is not None and G.numel() > 0) else None  # (an undefined at::Tensor arrives as None)
                if part is not None and part.numel() > 0:
                    dQ = (dQ, part)
                ctx.opx_packed = True
            else:
                try:
                    row_loss, row_lse, loss_sum, G, dQ, dC_part = kn.train_step_packed_f32(q, Cb, Qb, W, r, n_ctx, pos_idx, inv_T, grad_scale,
                                                                                           1.0 / Nq, used, dc_dtype, defer_dq=True, want_G="auto")
                except Exception as e:  # a plan without a bf16 dC epilogue: fp32 partials, rounded to the wire format in backward
                    if dc_dtype == torch.float32 or "dc_kind" not in str(e):
                        raise
                    row_loss, row_lse, loss_sum, G, dQ, dC_part = kn.train_step_packed_f32(q, Cb, Qb, W, r, n_ctx, pos_idx, inv_T, grad_scale,
                                                                                           1.0 / Nq, used, torch.float32, defer_dq=True,
                                                                                           want_G="auto")
            eager, loss_is_mean = (dQ, dC_part), True
        elif multi and packed_step:  # (stand-in kernels of the CPU tests)
            row_loss, row_lse, loss_sum, G, dQ, dC_part = kn.inbatch_step_packed_f32(q, Cb, Qb, W, r, n_ctx, pos_idx, inv_T, grad_scale)
            eager = (dQ, dC_part)
        elif q_f32 and wants_grad and hasattr(kn, "train_step_f32"):
            # a backward will follow: forward and backward in ONE library call (two launches at the BASELINE shapes);
            # backward() only checks the grad_output it was given against the one the gradients were scaled by
            used = _ExpectedGradScale.get(q.device, (B, Nc, d))
            row_loss, row_lse, loss_sum, G, dQ, dC_part = kn.train_step_f32(q, c if c_direct else None, Qb, Cb, pos_idx, y_off, colmask,
                                                                            inv_T, grad_scale, 1.0 / Nq, used, defer_dq=True, want_G="auto")
            eager, loss_is_mean = (dQ, dC_part), True
        elif q_f32 and wants_grad and hasattr(kn, "inbatch_step_f32"):  # (stand-in kernels of the CPU tests)
            row_loss, row_lse, loss_sum, G, dQ, dC_part = kn.inbatch_step_f32(q, c if c_direct else None, Qb, Cb, pos_idx,
                                                                              y_off, colmask, inv_T, grad_scale)
            eager = (dQ, dC_part)
        elif q_f32:
            row_loss, row_lse, loss_sum, G, _ = kn.inbatch_fwd_f32(q, c if c_direct else None, Qb, Cb, pos_idx, y_off,
                                                                   colmask, inv_T, grad_scale)
        else:
            row_loss, row_lse, loss_sum, G, _ = kn.inbatch_fwd(Qb, Cb, pos_idx, y_off, colmask, inv_T, grad_scale)
        loss_sum = loss_sum[:1]
        if multi:
            D.all_reduce_sum(loss_sum, group)  # (of the means when loss_is_mean: the sum over ranks is the global mean)
        loss = (loss_sum if loss_is_mean else loss_sum / Nq).reshape(())

        ctx.kn, ctx.group = kn, group
        ctx.dims = (W, r, B, d, n_ctx, rows_c)
        ctx.site = (B, Nc, d)
        ctx.multi = multi
        ctx.direct_ctx_grad = gather is not None  # c came straight from defer_context_grad (the task's own flow)
        ctx.in_dtypes = (q.dtype, c.dtype)
        ctx.eager, ctx.used = eager, used
        ctx.spare = (Qb, Cb, G) if (eager is not None and used is not None) else None
        # (G is None where the step never materialised the dScores: a second backward through a retained graph recomputes them)
        ctx.regen = (pos_idx, y_off, None if (multi and packed_step) else colmask, inv_T, grad_scale) if ctx.spare is not None and G is None else None
        ctx.pending = pending if multi else None
        if eager is None:
            ctx.save_for_backward(Qb, Cb, G)
        ctx.row_lse = row_lse
        ctx.dbg = None if S_dbg is None else (S_dbg, pos_idx + y_off, grad_scale)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.fast is not None or getattr(ctx, "fast_done", False):
            need_dq, need_dc = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
            go = grad_out if (grad_out.dtype == torch.float32 and grad_out.is_contiguous()) else grad_out.detach().float().contiguous()
            if ctx.fast is not None:
                # computed in forward for grad_output = *used: ONE launch compares and only rescales when the scale really changed; the
                # tensors are handed to autograd and forgotten (see the general path below)
                dQ, dC, part, used, site = ctx.fast
                ctx.fast, ctx.fast_done = None, True
                _, nxt = _OPX.rescale(dQ, part, dC, go, used, need_dq, need_dc)
                _ExpectedGradScale.publish(go.device, nxt, site)
            else:
                # a second backward through a retained graph: the backward GEMMs run again on the operands the step left behind
                kn = default_kernels()
                Qb, Cb, G, pos_idx, mask, inv_T, grad_scale = ctx.regen
                if G is None:
                    G = kn.inbatch_fwd(Qb, Cb, pos_idx, 0, mask.view(torch.uint8) if mask.dtype == torch.bool else mask, inv_T, grad_scale)[3]
                    ctx.regen = (Qb, Cb, G, pos_idx, mask, inv_T, grad_scale)
                dQ, dC = kn.inbatch_bwd(G, Qb, Cb, 1.0, go.reshape(1), need_dq, need_dc)
            return (dQ if need_dq else None), (dC if need_dc else None), None, None, None, None, None, None, None
        kn, group = ctx.kn, ctx.group
        W, r, B, d, n_ctx, rows_c = ctx.dims
        need_dq, need_dc = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        go = grad_out.detach().reshape(1).float().contiguous()  # device scalar: AMP loss scale, no host sync
        mine_ready = None
        if ctx.eager is not None and ctx.used is not None:
            # computed in forward for grad_output = *ctx.used; ONE launch compares and only rescales when the scale really changed.
            # The tensors are handed to autograd and FORGOTTEN here: AccumulateGrad takes a gradient nobody else references as
            # .grad itself, and clones it otherwise (one copy launch per step, 50 MB of traffic at cfg3 per rank).
            dQ, dC_part = ctx.eager
            ctx.eager = None
            mine_ready = None
            if ctx.opx_packed and ctx.multi:
                # csrc/opx.cpp: the rescale launch, dC_part in the wire format (one cast launch where the step wrote fp32 for a bf16 wire)
                # and the reduce-scatter's receive buffer, in one C++ call
                part = dQ[1] if isinstance(dQ, tuple) else None
                nxt, dC_part, mine_ready = _OPX.packed_backward(dQ[0] if isinstance(dQ, tuple) else dQ, part, dC_part, go, ctx.used,
                                                                0 if _dc_wire_dtype(group) == _BF16 else 2, rows_c, need_dq, need_dc)
            else:
                out2 = kn.rescale_grads(dQ if need_dq else None, dC_part if need_dc else None, go, ctx.used)
                nxt = out2[1:2]
            if isinstance(dQ, tuple):
                dQ = dQ[0]  # (the slabs have been added up into it)
            ctx.used = None
            _ExpectedGradScale.publish(go.device, nxt, ctx.site)
            go = None
        elif ctx.spare is not None:
            # a second backward through a retained graph: the first one gave its gradient tensors away; the backward GEMMs run
            # again on the operands the step left behind (exact, rare)
            Qb, Cb, G = ctx.spare
            if G is None:
                pos_idx, y_off, colmask, inv_T, grad_scale = ctx.regen
                if colmask is None:
                    colmask = kn.empty((Cb.shape[0],), torch.uint8, Cb)
                    kn.unpack_mask(Cb, W, n_ctx, colmask)
                G = kn.inbatch_fwd(Qb, Cb, pos_idx, y_off, colmask, inv_T, grad_scale)[3]
                ctx.spare = (Qb, Cb, G)
            dQ, dC_part = kn.inbatch_bwd(G, Qb, Cb, 1.0, go, need_dq, need_dc)
            go = None
        elif ctx.eager is not None:
            dQ, dC_part = ctx.eager  # computed in forward for grad_output = 1 (stand-in kernels); the scale is applied below
        elif ctx.dbg is not None:
            # fp32-G debug mode: G = (exp(S - lse) - onehot) * scale recomputed in fp32, split into a bf16 pair, two GEMM passes
            Qb, Cb, _ = ctx.saved_tensors
            S, labels, gs = ctx.dbg
            G32 = torch.exp(S - ctx.row_lse[:, None])
            G32[torch.arange(S.shape[0], device=S.device), labels] -= 1.0
            G32 *= gs
            hi = G32.to(_BF16)
            lo = (G32 - hi.float()).to(_BF16)
            dQ1, dC1 = kn.inbatch_bwd(hi, Qb, Cb, 1.0, go, need_dq, need_dc)
            dQ2, dC2 = kn.inbatch_bwd(lo, Qb, Cb, 1.0, go, need_dq, need_dc)
            dQ = dQ1 + dQ2 if need_dq else None
            dC_part = dC1 + dC2 if need_dc else None
            go = None
        else:
            Qb, Cb, G = ctx.saved_tensors
            dQ, dC_part = kn.inbatch_bwd(G, Qb, Cb, 1.0, go, need_dq, need_dc)
            go = None
        dq = dc = None
        if need_dq:
            dq = (dQ if go is None else dQ * go).to(ctx.in_dtypes[0])
        if need_dc:
            if not ctx.multi:
                dc = dC_part[:n_ctx]
                dc = (dc if go is None else dc * go).to(ctx.in_dtypes[1])
            else:
                wire = _dc_wire_dtype(group)
                mine = mine_ready if mine_ready is not None else kn.empty((rows_c, d), wire, dC_part)
                if go is not None:
                    dC_part = dC_part * go  # scale before the collective: nothing is left to do after it
                if dC_part.dtype != wire:
                    dC_part = dC_part.to(wire)  # half the bytes on the links (and in RCCL's reduction)
                widen = wire != ctx.in_dtypes[1] and ctx.in_dtypes[1] == torch.float32 and hasattr(kn, "widen")
                if ctx.pending is not None and ctx.direct_ctx_grad and (wire == ctx.in_dtypes[1] or widen):
                    # reduce-scatter on RCCL's stream; whoever consumes dc (defer_context_grad, after the query-tower backward has
                    # been enqueued) waits for it -- and, on a half-width wire, widens the result into the fp32 gradient there.
                    # Only in the task's own flow (c IS the output of defer_context_grad: `gather` given), where autograd hands this
                    # very tensor to that node.  Should anything still sit in between (another consumer of c, a hook), the tensor that
                    # arrives there is a different one: the half-width wire therefore returns ZEROS here (whatever autograd adds to
                    # them stays right) and post() adds the widened rows to whatever arrives instead of overwriting it.
                    ctx.pending.work = D.reduce_scatter_rows(dC_part, mine, group, async_op=True)
                    if widen:
                        # nothing to return yet: a gradient of zeros that costs no launch (one cached zero, expanded).  post() hands
                        # the widened rows on in its place, or adds them to whatever autograd made of it (zeros + somebody else's term)
                        dc = _zeros_view((n_ctx, d), mine.device)

                        def post(g, kn=kn, mine=mine, zptr=dc.data_ptr(), shape=(n_ctx, d)):
                            if g.data_ptr() == zptr and g.stride() == (0, 0):
                                return kn.widen(mine, kn.empty(shape, torch.float32, mine))
                            g.add_(kn.widen(mine, kn.empty(shape, torch.float32, mine)).to(g.dtype))
                            return g

                        ctx.pending.post = post
                    else:
                        dc = mine[:n_ctx]
                else:
                    D.reduce_scatter_rows(dC_part, mine, group)  # sum over ranks of the partials of MY columns
                    if widen:
                        dc = kn.widen(mine, kn.empty((n_ctx, d), torch.float32, mine))
                    else:
                        dc = mine[:n_ctx].to(ctx.in_dtypes[1])
        return dq, dc, None, None, None, None, None, None, None


def _pad_hidden(q, c):
    """Hidden sizes that are not a multiple of 8 (16-byte bf16 rows) get zero columns appended: dot products are
    unchanged and autograd slices the gradients back (e.g. 30522-wide router vectors, citadel_task.py:249-262)."""
    pad = (-q.shape[1]) % 8
    if pad == 0:
        return q, c
    return torch.nn.functional.pad(q, (0, pad)), torch.nn.functional.pad(c, (0, pad))


def _opx_regen(Qb, Cb, G, pos_idx, mask, inv_T, grad_scale, go, need_dq, need_dc):
    """Second backward through a retained graph of the C++ node (csrc/opx.cpp: InBatchFn::backward): the backward GEMMs again, on the
    operands the step left behind; the dScores are recomputed first where the step never materialised them."""
    kn = default_kernels()
    if G is None:
        G = kn.inbatch_fwd(Qb, Cb, pos_idx, 0, mask.view(torch.uint8) if mask.dtype == torch.bool else mask, inv_T, grad_scale)[3]
    dQ, dC = kn.inbatch_bwd(G, Qb, Cb, 1.0, go.reshape(1), need_dq, need_dc)
    return dQ, dC, G


_OPX_NODE = _OPX is not None and hasattr(_OPX, "inbatch_loss") and os.environ.get("DPRHOT_OPX_NODE", "1") != "0"
if _OPX_NODE:
    _OPX.set_regen(_opx_regen)


def inbatch_contrastive_loss(q, c, pos_idx, ctx_mask, temperature=1.0, group=None, kernels=None, gather=None, pending=None):
    if gather is None:
        if (_OPX_NODE and kernels is None and q.is_cuda and (group is False or D.world(group)[0] == 1) and not _fp32_g_mode()
                and not (D.force_dist() and group is not False)):
            # the plain single-rank fp32 step: a C++ autograd node (csrc/opx.cpp), no Python frame in forward or backward
            # (DPRHOT_OPX_NODE=0: InBatchContrastive's own fast path, the same two library calls from Python)
            loss = _OPX.inbatch_loss(q, c, pos_idx, ctx_mask, 1.0 / float(temperature), default_kernels()._lib.options_epoch())
            if loss is not None:
                return loss
        q, c = _pad_hidden(q, c)
    return InBatchContrastive.apply(q, c, pos_idx, ctx_mask, temperature, group, kernels, gather, pending)


class WindowedContrastive(torch.autograd.Function):
    """in_batch_negatives=False (dpr_task.py:198-207): query i is scored only against its own K contexts
    [pos_i, pos_i + K) (dummy ones masked).  No gather in the reference on this branch either."""

    @staticmethod
    def forward(ctx, q, c, pos_idx, ctx_mask, temperature, kernels):
        kn = kernels if kernels is not None else default_kernels()
        B, d = q.shape
        n_ctx = c.shape[0]
        K = n_ctx // B
        Nc_pad = _pad_cols(n_ctx)
        Qb = kn.empty((B, d), _BF16, q)
        Cb = kn.empty((Nc_pad, d), _BF16, c)
        kn.prep(q, Qb, c, Cb[:n_ctx])
        m8 = torch.ones(Nc_pad, dtype=torch.uint8, device=c.device)
        m8[:n_ctx].copy_(ctx_mask.view(torch.uint8) if ctx_mask.dtype == torch.bool else ctx_mask)
        if Nc_pad != n_ctx:
            Cb[n_ctx:].zero_()
        inv_T = 1.0 / float(temperature)
        S = kn.sim(Qb, Cb, m8, inv_T)
        row_loss, _, G = kn.softmax_ce(S, pos_idx, 0, inv_T / B, want_G=True, row_win_start=pos_idx, win_len=K)
        loss = kn.reduce_sum(row_loss, 1.0 / B).reshape(())
        ctx.kn, ctx.n_ctx, ctx.in_dtypes = kn, n_ctx, (q.dtype, c.dtype)
        ctx.save_for_backward(Qb, Cb, G)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        Qb, Cb, G = ctx.saved_tensors
        go = grad_out.detach().reshape(1).float().contiguous()
        dQ, dC = ctx.kn.inbatch_bwd(G, Qb, Cb, 1.0, go, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        dq = dQ.to(ctx.in_dtypes[0]) if dQ is not None else None
        dc = dC[:ctx.n_ctx].to(ctx.in_dtypes[1]) if dC is not None else None
        return dq, dc, None, None, None, None


def windowed_contrastive_loss(q, c, pos_idx, ctx_mask, temperature=1.0, kernels=None):
    q, c = _pad_hidden(q, c)
    return WindowedContrastive.apply(q, c, pos_idx, ctx_mask, temperature, kernels)


# ---- scoring helpers used by the task's eval path and by subclasses that train through sim_score ----------------
def _sim_fwd(kn, q, c, colmask, inv_T):
    """(S, Qb, Cb, Nc) -- fp32 logits [Nq, Nc_pad] and the bf16 operand images (the backward of SimScore re-uses them)."""
    Nc = c.shape[0]
    Nc_pad = _pad_cols(Nc)
    Qb = kn.empty(tuple(q.shape), _BF16, q)
    kn.cast_bf16(q, Qb)
    Cb = kn.empty((Nc_pad, c.shape[1]), _BF16, c)
    kn.cast_bf16(c, Cb[:Nc])
    m8 = None
    if colmask is not None or Nc_pad != Nc:
        m8 = torch.zeros(Nc_pad, dtype=torch.uint8, device=c.device)
        if colmask is not None:
            m8[:Nc].copy_(colmask.view(torch.uint8) if colmask.dtype == torch.bool else colmask)
        if Nc_pad != Nc:
            Cb[Nc:].zero_()
            m8[Nc:] = 1
    return kn.sim(Qb, Cb, m8, inv_T), Qb, Cb, Nc


class SimScore(torch.autograd.Function):
    """scores = sim_score(q, c, colmask) * inv_T with gradients (dpr_task.py:98-105 is a differentiable torch.matmul):
    backward dq = inv_T * dS x C, dc = inv_T * dS^T x Q on the HIP backward GEMMs.  The incoming dS is rounded to bf16 for the
    MFMA operands, exactly like the dScores of the fused training step; masked columns receive no gradient (the reference's
    index-put leaves them out of the graph as well)."""

    @staticmethod
    def forward(ctx, q, c, colmask, inv_T, kernels):
        kn = kernels if kernels is not None else default_kernels()
        S, Qb, Cb, Nc = _sim_fwd(kn, q.detach(), c.detach(), colmask, inv_T)
        ctx.kn, ctx.Nc, ctx.inv_T, ctx.in_dtypes = kn, Nc, float(inv_T), (q.dtype, c.dtype)
        ctx.save_for_backward(Qb, Cb, colmask)
        return S if S.shape[1] == Nc else S[:, :Nc]

    @staticmethod
    def backward(ctx, dS):
        Qb, Cb, colmask = ctx.saved_tensors
        kn, Nc = ctx.kn, ctx.Nc
        G = kn.empty((dS.shape[0], Cb.shape[0]), _BF16, dS)
        if Cb.shape[0] != Nc:
            G[:, Nc:].zero_()
        g = dS.detach().float()
        if colmask is not None:
            g = g.masked_fill(colmask.bool()[None, :], 0.0)
        G[:, :Nc].copy_(g)
        dq = dc = None
        if ctx.needs_input_grad[0]:
            dq = kn.dq(G, Cb, ctx.inv_T).to(ctx.in_dtypes[0])
        if ctx.needs_input_grad[1]:
            dc = kn.dc(G, Qb, ctx.inv_T)[:Nc].to(ctx.in_dtypes[1])
        return dq, dc, None, None, None


def sim_score(q, c, colmask=None, inv_T=1.0, kernels=None):
    """dpr_task.py:98-105 on the device: fp32 logits [Nq, Nc] from fp32/bf16 embeddings; colmask is the [Nc]
    dummy-context mask (the row the reference broadcasts at :197).  Differentiable when an input requires grad."""
    kn = kernels if kernels is not None else default_kernels()
    q, c = _pad_hidden(q, c)
    if torch.is_grad_enabled() and (q.requires_grad or c.requires_grad):
        if not hasattr(kn, "dq"):
            raise RuntimeError(f"sim_score: an input requires grad but the kernels object {getattr(kn, 'name', kn)!r} has no backward "
                               "GEMMs (dq / dc); scores without a grad_fn would silently train nothing")
        return SimScore.apply(q, c, colmask, inv_T, kn)
    S, _, _, Nc = _sim_fwd(kn, q.detach(), c.detach(), colmask, inv_T)
    return S if S.shape[1] == Nc else S[:, :Nc]


class PairwiseScore(torch.autograd.Function):
    """citadel_task.py:137-146 `sim_score(q, c, mask, pairwise=True)`: scores[b][j] = <q[b], c[b*M + j]>, -inf at masked
    pairs; differentiable (HBM-streaming HIP kernels, fp32)."""

    @staticmethod
    def forward(ctx, q, c, mask, kernels):
        kn = kernels if kernels is not None else default_kernels()
        qf, cf = q.detach().float().contiguous(), c.detach().float().contiguous()
        m8 = None if mask is None else (mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)).contiguous().view(-1)
        ctx.kn, ctx.in_dtypes = kn, (q.dtype, c.dtype)
        ctx.save_for_backward(qf, cf, m8)
        return kn.pairwise_fwd(qf, cf, m8)

    @staticmethod
    def backward(ctx, dS):
        qf, cf, m8 = ctx.saved_tensors
        g = dS.detach().float()
        if m8 is not None:
            g = g.masked_fill(m8.view_as(g).bool(), 0.0)  # -inf entries were index-put: no gradient flows through them
        dq, dc = ctx.kn.pairwise_bwd(g.contiguous(), qf, cf, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return (dq.to(ctx.in_dtypes[0]) if dq is not None else None, dc.to(ctx.in_dtypes[1]) if dc is not None else None, None, None)


def pairwise_score(q, c, mask=None, kernels=None):
    """[B, M] scores of every query against its own M = c.shape[0] // B contexts (mask: [B*M] or [B, M] bool)."""
    return PairwiseScore.apply(q, c, mask, kernels)


def cross_entropy_mean(S, labels, kernels=None):
    """nn.CrossEntropyLoss()(S, labels) for existing fp32 logits (eval: dpr_task.py:224,299)."""
    kn = kernels if kernels is not None else default_kernels()
    S = S.contiguous()
    if S.shape[1] % 8 != 0:
        pad = _pad_cols(S.shape[1]) - S.shape[1]
        S = torch.nn.functional.pad(S, (0, pad), value=float("-inf"))
    row_loss, _, _ = kn.softmax_ce(S, labels)
    return kn.reduce_sum(row_loss, 1.0 / S.shape[0]).reshape(())


def rank_of_gold(S, labels, kernels=None):
    kn = kernels if kernels is not None else default_kernels()
    S = S.contiguous()
    if S.shape[1] % 4 != 0:
        S = torch.nn.functional.pad(S, (0, 4 - S.shape[1] % 4), value=float("-inf"))
    return kn.rank_of_gold(S, labels)


def rank_and_loss(q, c, labels, colmask=None, inv_T=1.0, kernels=None):
    """(ranks [Nq] int64, mean cross-entropy) of scores = q x c^T * inv_T with masked columns at -inf -- what
    compute_rank_metrics + self.loss (dpr_task.py:235-246,:299) derive from the score matrix, here without ever storing it
    when the problem is large (validation over a whole epoch's embeddings: 8192 x 65536 logits would be 2 GiB)."""
    kn = kernels if kernels is not None else default_kernels()
    q, c = _pad_hidden(q.detach(), c.detach())  # hidden sizes that are not a multiple of 8 (tiny encoders, projection heads)
    Nc = c.shape[0]
    Nc_pad = _pad_cols(Nc)
    Qb = kn.empty(tuple(q.shape), _BF16, q)
    kn.cast_bf16(q, Qb)
    Cb = kn.empty((Nc_pad, c.shape[1]), _BF16, c)
    kn.cast_bf16(c, Cb[:Nc])
    m8 = torch.zeros(Nc_pad, dtype=torch.uint8, device=c.device)
    if colmask is not None:
        m8[:Nc].copy_(colmask.view(torch.uint8) if colmask.dtype == torch.bool else colmask)
    if Nc_pad != Nc:
        Cb[Nc:].zero_()
        m8[Nc:] = 1
    y = torch.as_tensor(labels, dtype=torch.long, device=q.device)
    if hasattr(kn, "sim_rank_loss"):  # ONE pass of the Nq x Nc GEMM: count and softmax statistics in the same epilogue
        ranks, loss_sum = kn.sim_rank_loss(Qb, Cb, y, m8, inv_T)
    else:
        ranks = kn.sim_rank(Qb, Cb, y, m8, inv_T)
        _, _, loss_sum, _, _ = kn.inbatch_fwd(Qb, Cb, y, 0, m8, inv_T, 1.0, want_logits=False, want_G=False)
    return ranks, (loss_sum / q.shape[0]).reshape(())


def topk(S, k, kernels=None):
    kn = kernels if kernels is not None else default_kernels()
    return kn.topk(S.contiguous(), k)


class CorpusSearch:
    """search_index of run_retrieval_pytorch.py:141-166 plus its shard loop (:196-243) and re-merge (:272-277):

        s = CorpusSearch(query_embs, topk)      # [nq, d] fp32/bf16 on the GPU
        for shard, first_id in shards:          # each [n, d] corpus shard resident on the GPU, any n
            s.add(shard, first_id)
        scores, ids = s.result()                # [nq, topk] fp32 / int64, best first, ties by lower id

    The [nq, n] score matrix never exists: each `chunk` of passages is scored on the bf16 MFMA path and folded
    into the running top-k on the device (dprhot_search)."""

    KMAX = 1024   # largest k of dprhot_search (GEMM with the filter epilogue); run_retrieval_pytorch.py:149 accepts any --topk
    KWIDE = 4096  # largest k of the streaming top-k kernel (dprhot_topk_update: 8192 sort slots in 96 KB of LDS)

    def __init__(self, query_embs, k, chunk=None, kernels=None):
        self.kn = kernels if kernels is not None else default_kernels()
        nq, d = query_embs.shape
        self.Qb = self.kn.empty((nq, d), _BF16, query_embs)
        self.kn.cast_bf16(query_embs, self.Qb)
        self.k = int(k)
        self.wide_k = self.k > self.KMAX
        if chunk is None:  # filtered chunks of up to 262144 passages (2 GiB of candidate workspace at 1024 queries); the library starts
            chunk = max(1024, min(262144, (1 << 28) // max(nq, 1) // 8 * 8))  # an empty state from a 65536-passage head on its own
        self.chunk = int(chunk) // 8 * 8
        self.values = torch.full((nq, k), float("-inf"), dtype=torch.float32, device=query_embs.device)
        self.indices = torch.full((nq, k), -1, dtype=torch.int64, device=query_embs.device)
        self.ws = None if self.wide_k else self.kn.search_workspace(nq, self.chunk, query_embs)
        self.wide_ws = None
        self.first = True

    def _add_wide_native(self, Cb, first_id):
        """1024 < k <= 4096: the scores of a chunk from the MFMA path (dprhot_sim_fwd), folded into the running top-k by the streaming
        kernel's wide instantiation (dprhot_topk_update: radix select from an empty state, then a threshold filter + bitonic merges in
        LDS) -- no torch sort anywhere.  Chunks of 65536 passages: the score matrix of a chunk is nq x 256 KiB."""
        n, d = Cb.shape
        step = min(self.chunk, 65536)
        for j0 in range(0, n, step):
            cols = min(step, n - j0)
            pad = (-cols) % 8
            blk = Cb[j0:j0 + cols]
            if pad:
                blk = torch.cat([blk, torch.zeros((pad, d), dtype=_BF16, device=Cb.device)], 0)
            S = self.kn.sim(self.Qb, blk.contiguous(), None, 1.0)
            self.kn.topk_update(S, cols, first_id + j0, self.values, self.indices, self.first)
            self.first = False

    def _add_wide(self, Cb, first_id):
        """k beyond 4096: the scores come from the MFMA path chunk by chunk (dprhot_sim_fwd), the selection is the library's
        HBM-resident one (dprhot_topk_update_wide, csrc/wideselect.h: exact radix select over state + chunk, ties at the k-th score
        resolved by passage id, then only the k winners are sorted) -- no torch sort, no host sync per chunk.  _fold_rows_torch below is
        the path of the stand-in kernels only."""
        n, d = Cb.shape
        step = min(self.chunk, 65536)
        if self.wide_ws is None and hasattr(self.kn, "topk_update_wide"):
            self.wide_ws = self.kn.topk_wide_workspace(self.values.shape[0], self.k, Cb)
        for j0 in range(0, n, step):
            cols = min(step, n - j0)
            pad = (-cols) % 8
            blk = Cb[j0:j0 + cols]
            if pad:
                blk = torch.cat([blk, torch.zeros((pad, d), dtype=_BF16, device=Cb.device)], 0)
            S = self.kn.sim(self.Qb, blk.contiguous(), None, 1.0)
            if self.wide_ws is not None:
                self.kn.topk_update_wide(S, cols, first_id + j0, self.values, self.indices, self.first, self.wide_ws)
            else:
                if self.first:
                    self.values.fill_(float("-inf"))
                    self.indices.fill_(-1)
                self._fold_rows_torch(None, S[:, :cols], first_id + j0)
            self.first = False

    def _fold_rows_torch(self, rows, S, first_col):
        """Exact fold of one chunk into the state of `rows` (None: all) with torch's stable sorts, in the same total order (score desc,
        id asc): the library's escape for degenerate rows, and the whole path for stand-in kernels."""
        sel = slice(None) if rows is None else rows
        S = S[sel]
        vals, idx = self.values[sel], self.indices[sel]
        cols = S.shape[1]
        kk = min(self.k, cols)
        order = torch.sort(S, dim=1, descending=True, stable=True).indices[:, :kk]  # ties: lower column first
        v = torch.gather(S, 1, order)
        ids = order + first_col
        allv, alli = torch.cat([vals, v], 1), torch.cat([idx, ids], 1)
        alli_key = torch.where(alli < 0, torch.full_like(alli, torch.iinfo(torch.int64).max), alli)  # empty slots last
        o1 = torch.sort(alli_key, dim=1, stable=True).indices
        allv, alli = torch.gather(allv, 1, o1), torch.gather(alli, 1, o1)
        o2 = torch.sort(allv, dim=1, descending=True, stable=True).indices[:, :self.k]
        self.values[sel], self.indices[sel] = torch.gather(allv, 1, o2), torch.gather(alli, 1, o2)

    def add(self, corpus_embs, first_id=0):
        n, d = corpus_embs.shape
        if corpus_embs.dtype == _BF16 and corpus_embs.is_contiguous():
            Cb = corpus_embs
        else:
            Cb = self.kn.empty((n, d), _BF16, corpus_embs)
            self.kn.cast_bf16(corpus_embs, Cb)
        if self.wide_k:
            if self.k <= self.KWIDE and hasattr(self.kn, "topk_update"):
                return self._add_wide_native(Cb, first_id)
            return self._add_wide(Cb, first_id)
        n8 = n // 8 * 8
        if n8:
            self.kn.search(self.Qb, Cb[:n8], first_id, self.values, self.indices, self.first, self.chunk, self.ws)
            self.first = False
        if n8 != n:  # ragged tail: 8 padded rows, only the real columns are scanned
            tail = torch.zeros((8, d), dtype=_BF16, device=Cb.device)
            tail[: n - n8].copy_(Cb[n8:])
            S = self.kn.sim(self.Qb, tail, None, 1.0)
            self.kn.topk_update(S, n - n8, first_id + n8, self.values, self.indices, self.first)
            self.first = False

    def result(self):
        if self.wide_ws is not None and hasattr(self.kn, "topk_wide_errors"):
            err = self.kn.topk_wide_errors(self.wide_ws, self.values.shape[0])
            if bool(err.any()):  # (one sync, where the caller is about to read the result anyway; no condition sets the word today)
                raise RuntimeError("dprhot_topk_update_wide reported rows it could not update: " + str(torch.nonzero(err).flatten().tolist()[:8]))
        return self.values, self.indices
