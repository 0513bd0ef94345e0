"""TEST-ONLY stand-in for dpr_scale_amd.hotpath.HipKernels, built on the numpy oracle, so that the CPU suite
can run the *distributed orchestration* of InBatchContrastive on gloo (gather layout, label offsets,
reduce-scatter, loss all-reduce).  It lives under tests/ and is never importable from the product."""
import numpy as np
import torch

from oracle import inbatch_oracle as O


def _np(t):
    return t.detach().float().cpu().numpy()


class OracleKernels:
    name = "oracle-test-standin"

    def empty(self, shape, dtype, like):
        return torch.empty(shape, dtype=dtype, device=like.device)

    def cast_bf16(self, src, dst):
        dst.copy_(src.detach().to(torch.bfloat16))
        return dst

    def packed_rows(self, n_ctx, d):
        extra = -(-n_ctx // (2 * d))
        align = 64 if n_ctx >= 2048 else 8  # (as dprhot_packed_rows: whole 64-deep K steps over the gathered axis for large batches)
        return -(-(n_ctx + extra) // align) * align

    def pack_ctx(self, c, m8, send):
        n_ctx, d = c.shape
        send.zero_()
        send[:n_ctx].copy_(c.detach().to(torch.bfloat16))
        send.view(torch.uint8).view(-1)[n_ctx * d * 2:n_ctx * d * 2 + n_ctx] = (m8 != 0).to(torch.uint8)

    def unpack_mask(self, gathered, W, n_ctx, colmask):
        d = gathered.shape[1]
        rows_c = self.packed_rows(n_ctx, d)
        by = gathered.view(torch.uint8).view(W, rows_c * d * 2)
        cm = torch.ones(W, rows_c, dtype=torch.uint8)
        cm[:, :n_ctx] = by[:, n_ctx * d * 2:n_ctx * d * 2 + n_ctx]
        colmask.copy_(cm.view(-1))

    def prep(self, q, Qb, c, Cdst):
        self.cast_bf16(q, Qb)
        self.cast_bf16(c, Cdst)

    def inbatch_fwd(self, Qb, Cb, y, y_offset, colmask, inv_T, grad_scale, want_logits=False, want_G=True):
        S = O.sim_score(_np(Qb), _np(Cb), colmask.numpy().astype(bool)) * inv_T
        labels = y.numpy() + y_offset
        _, row_loss, lse = O.log_softmax_ce(S, labels)
        G = O.dscores(S, labels, lse, grad_scale)
        t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
        return (t(row_loss), t(lse), t(np.array([row_loss.sum()])), t(G, torch.bfloat16) if want_G else None,
                t(S) if want_logits else None)

    def inbatch_bwd(self, G, Qb, Cb, h_scale, d_scale, need_dq=True, need_dc=True):
        s = h_scale * (float(d_scale.item()) if d_scale is not None else 1.0)
        g = _np(G).astype(np.float64)
        dQ = torch.from_numpy((g @ _np(Cb).astype(np.float64) * s).astype(np.float32)) if need_dq else None
        dC = torch.from_numpy((g.T @ _np(Qb).astype(np.float64) * s).astype(np.float32)) if need_dc else None
        return dQ, dC

    # fp32-operand forms: the product rounds to bf16 while staging and leaves the bf16 images in Qb / Cb
    def dq(self, G, Cb, h_scale=1.0, d_scale=None):
        return self.inbatch_bwd(G, None, Cb, h_scale, d_scale, True, False)[0]

    def dc(self, G, Qb, h_scale=1.0, d_scale=None):
        return self.inbatch_bwd(G, Qb, None, h_scale, d_scale, False, True)[1]

    def pairwise_fwd(self, q, c, m8):
        B, d = q.shape
        S = (q.double()[:, None, :] * c.double().view(B, -1, d)).sum(-1).float()
        if m8 is not None:
            S = S.masked_fill(m8.view_as(S).bool(), float("-inf"))
        return S

    def pairwise_bwd(self, g, q, c, need_dq=True, need_dc=True):
        B, d = q.shape
        c3 = c.view(B, -1, d)
        dq = (g[:, :, None] * c3).sum(1) if need_dq else None
        dc = (g[:, :, None] * q[:, None, :]).reshape(c.shape) if need_dc else None
        return dq, dc

    def inbatch_fwd_f32(self, q, c, Qb, Cb, y, y_offset, colmask, inv_T, grad_scale, want_logits=False, want_G=True):
        self.cast_bf16(q, Qb)
        if c is not None:
            self.cast_bf16(c, Cb)
        return self.inbatch_fwd(Qb, Cb, y, y_offset, colmask, inv_T, grad_scale, want_logits, want_G)

    def inbatch_step_f32(self, q, c, Qb, Cb, y, y_offset, colmask, inv_T, grad_scale):
        row_loss, lse, loss_sum, G, _ = self.inbatch_fwd_f32(q, c, Qb, Cb, y, y_offset, colmask, inv_T, grad_scale)
        dQ, dC = self.inbatch_bwd(G, Qb, Cb, 1.0, None)
        return row_loss, lse, loss_sum, G, dQ, dC

    # ---- forward-only / piecewise ops (eval path, windowed branch) ----
    def sim(self, Qb, Cb, colmask=None, inv_T=1.0):
        m = None if colmask is None else colmask.numpy().astype(bool)
        return torch.from_numpy((O.sim_score(_np(Qb), _np(Cb), m) * inv_T).astype(np.float32))

    def softmax_ce(self, S, y, y_offset=0, grad_scale=1.0, want_G=False, row_win_start=None, win_len=0):
        s = S.detach().double().numpy().copy()
        labels = y.numpy() + y_offset
        if row_win_start is not None:
            cols = np.arange(s.shape[1])[None, :]
            lo = (row_win_start.numpy() + y_offset)[:, None]
            s[(cols < lo) | (cols >= lo + win_len)] = -np.inf
        _, row_loss, lse = O.log_softmax_ce(s, labels)
        G = torch.from_numpy(O.dscores(s, labels, lse, grad_scale)).to(torch.bfloat16) if want_G else None
        return torch.from_numpy(row_loss.astype(np.float32)), torch.from_numpy(lse.astype(np.float32)), G

    def reduce_sum(self, x, scale=1.0):
        return (x.double().sum() * scale).float().reshape(1)

    def rank_of_gold(self, S, y, y_offset=0):
        return torch.from_numpy(O.rank_of_gold(S.numpy(), y.numpy() + y_offset))

    def topk(self, S, k):
        v, i = O.topk_stable(S.numpy(), k)
        return torch.from_numpy(v), torch.from_numpy(i)

    def topk_update(self, S, cols, col_offset, values, indices, first):
        v, i = O.topk_merge(None if first else (values.numpy(), indices.numpy()), S.numpy()[:, :cols], col_offset,
                            values.shape[1])
        values.copy_(torch.from_numpy(v))
        indices.copy_(torch.from_numpy(i))

    def search_workspace(self, nq, chunk, like):
        return torch.empty(0, dtype=torch.uint8)

    def search(self, Qb, Cb, id_offset, values, indices, first, chunk, ws):
        for j0 in range(0, Cb.shape[0], chunk):
            S = self.sim(Qb, Cb[j0:j0 + chunk])
            self.topk_update(S, S.shape[1], id_offset + j0, values, indices, first and j0 == 0)
