"""Measurements for the two "next" rows of SURVEY.md section 8(f) that reuse the similarity kernel (extra to bench.py,
never instead of it):

  rank    validation scoring + rank metrics at scale (dpr_task.py:248-310, :235-246): Nq x Nc logits, rank of gold
  search  brute-force retrieval (run_retrieval_pytorch.py:141-166, :272-277): top-k passages per query over a corpus

Each is timed against the reference's own formulation written in torch ops on the same GPU (fp16 einsum + torch.topk;
matmul + full descending sort + per-row nonzero), with the results compared (ids bit-exact when the scores tie-break
identically, otherwise the overlap is reported).  One JSON line per measurement.
"""
import argparse
import json
import time

import torch


def _time(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters, out


def bench_search(nq, n, d, k, chunk, iters):
    from dpr_scale_amd.hotpath import CorpusSearch, default_kernels

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(nq, d, device=dev, generator=g)
    C = torch.randn(n, d, device=dev, generator=g).to(torch.bfloat16)
    kn = default_kernels()

    def ours():
        s = CorpusSearch(q, k, chunk=chunk or None, kernels=kn)
        s.add(C, 0)
        return s.result()

    C16 = C.to(torch.float16)

    def ref():  # run_retrieval_pytorch.py:148-150
        scores = torch.einsum("ik,jk->ij", q.to(torch.float16), C16)
        return torch.topk(scores, dim=-1, k=k)

    t_ours, (v, i) = _time(ours, iters)
    t_ref, (rv, ri) = _time(ref, iters)
    # agreement: fp16 scores of the reference tie/round differently; compare id sets
    inter = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(i[:64].cpu(), ri[:64].cpu())) / (64.0 * k)
    S = torch.sort(q[:8].to(torch.bfloat16).float() @ C.float().T, dim=1, descending=True, stable=True)
    exact = bool(torch.equal(S.indices[:, :10], i[:8, :10]))  # top-10 of fp32 torch scores (top-k deep ties may reorder)
    return {"what": "search", "workload": f"nq={nq} corpus={n} d={d} k={k} chunk={chunk or 'default (head 65536, then up to 262144)'} bf16 resident",
            "ms": round(t_ours * 1e3, 3), "queries_per_s": round(nq / t_ours, 1),
            "tflops_scoring": round(2.0 * nq * n * d / t_ours * 1e-12, 1),
            "torch_fp16_einsum_topk_ms": round(t_ref * 1e3, 3), "speedup_vs_torch": round(t_ref / t_ours, 2),
            "top10_equals_fp32_stable_sort": exact, "id_overlap_with_fp16_reference": round(inter, 4)}


def bench_rank(nq, nc, d, iters):
    from dpr_scale_amd.hotpath import default_kernels, rank_of_gold, sim_score

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn(nq, d, device=dev, generator=g)
    c = torch.randn(nc, d, device=dev, generator=g)
    y = torch.randint(0, nc, (nq,), device=dev, generator=g)
    kn = default_kernels()

    def ours():
        S = sim_score(q, c, kernels=kn)
        r = rank_of_gold(S, y, kernels=kn)
        return r.sum(), (1.0 / r.double()).sum(), (r <= 1).sum()  # three device scalars, no per-row host sync

    def ref():  # dpr_task.py:235-246 with the per-row python loop replaced by its vectorised equivalent
        S = torch.matmul(q, c.transpose(0, 1))
        _, indices = torch.sort(S, dim=1, descending=True)
        ranks = (indices == y[:, None]).nonzero()[:, 1] + 1
        return ranks.sum(), (1.0 / ranks.double()).sum(), (ranks <= 1).sum()

    def free():  # ranks AND mean loss without the score matrix (what _eval_epoch_end runs): count-greater GEMM + statistics GEMM
        from dpr_scale_amd.hotpath import rank_and_loss

        r, loss = rank_and_loss(q, c, y, None, 1.0, kn)
        return r.sum(), (1.0 / r.double()).sum(), (r <= 1).sum(), loss

    t_ours, o = _time(ours, iters)
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    t_free, f = _time(free, iters)
    peak_free = torch.cuda.max_memory_allocated() - base
    t_ref, r = _time(ref, iters)
    return {"what": "rank", "workload": f"Nq={nq} Nc={nc} d={d} fp32 embeddings in, fp32 logits",
            "ms": round(t_ours * 1e3, 3), "queries_per_s": round(nq / t_ours, 1),
            "score_free_ms_ranks_and_loss": round(t_free * 1e3, 3), "score_free_sum_rank": int(f[0]),
            "score_free_equals_score_matrix_ranks": bool(int(f[0]) == int(o[0]) and float(f[1]) == float(o[1])),
            "score_free_peak_extra_bytes": int(peak_free), "score_matrix_bytes": int(nq) * int(nc) * 4,
            "torch_matmul_sort_ms": round(t_ref * 1e3, 3), "speedup_vs_torch": round(t_ref / t_ours, 2),
            "sum_rank": int(o[0]), "torch_sum_rank": int(r[0]), "note": "torch scores in fp32/TF32-off matmul; ours bf16 inputs"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="both", choices=["search", "rank", "both"])
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--corpus", type=int, default=1 << 21)
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--chunk", type=int, default=0, help="0 = CorpusSearch default")
    ap.add_argument("--nc", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    if a.what in ("search", "both"):
        print(json.dumps(bench_search(a.nq, a.corpus, a.d, a.k, a.chunk, a.iters)), flush=True)
    if a.what in ("rank", "both"):
        print(json.dumps(bench_rank(8192, a.nc, a.d, a.iters)), flush=True)
