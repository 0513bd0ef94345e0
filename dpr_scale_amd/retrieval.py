"""Brute-force retrieval over pickled embedding shards -- the data path of dpr_scale/run_retrieval_pytorch.py
(build_index :177-189, search_index :141-175, the shard loop of main :196-243 and its re-merge :272-277) on the
streaming search of hotpath.CorpusSearch (dprhot_search): every shard is scored chunk by chunk on the bf16 MFMA path and
folded into ONE running top-k, so neither a [queries, passages] score matrix nor per-shard top-k lists exist.

On-disk format (what dpr_eval_task.py:40-49 writes and the reference reads): `reps_*` files under
`ctx_embeddings_dir`, each a pickled [n_i, d] float array; the query file is one pickled [nq, d] array.  Reading the
question / passage TSVs and writing the run file stay the reference's Python (merge_results and below) -- they take the
(scores, indexes) this returns.
"""
import glob
import os
import pickle

import torch

from . import hotpath


def _load(path):
    with open(path, "rb") as f:
        return torch.as_tensor(pickle.load(f))


def shard_files(ctx_embeddings_dir, shard):
    paths = sorted(glob.glob(os.path.join(ctx_embeddings_dir, "reps_*")))  # :203-205
    assert len(paths) > 0 and len(paths) % shard == 0, "Invalid Shard number"  # :207
    per = len(paths) // shard
    return [paths[s * per:(s + 1) * per] for s in range(shard)]


def build_index(paths, device):
    """The reference's build_index, including its sizing rule: the index has rows(first file) x len(paths) rows and files
    are copied in one after the other, so a short last file leaves zero vectors at the end -- which the reference
    searches (and counts in its id offsets) too."""
    first = _load(paths[0])
    index = torch.zeros((first.shape[0] * len(paths), first.shape[1]), dtype=torch.float32)
    n = 0
    for k, p in enumerate(paths):
        v = first if k == 0 else _load(p)
        index[n:n + v.shape[0]] = v.float()
        n += v.shape[0]
    return index.to(device)


def search_shards(query_embs, ctx_embeddings_dir, topk, shard=1, device=None, chunk=None, kernels=None):
    """(scores [nq, topk] fp32, indexes [nq, topk] int64) over all `reps_*` files, passage ids counted as the reference
    counts them (offset += len(index) per shard).  Order: score descending, ties by lower id."""
    q = _load(query_embs) if isinstance(query_embs, (str, os.PathLike)) else torch.as_tensor(query_embs)
    device = torch.device(device) if device is not None else torch.device("cuda", 0)
    search = hotpath.CorpusSearch(q.float().to(device), topk, chunk=chunk, kernels=kernels)
    offset = 0
    for paths in shard_files(ctx_embeddings_dir, shard):
        index = build_index(paths, device)
        search.add(index, offset)
        offset += index.shape[0]
        del index
    return search.result()
