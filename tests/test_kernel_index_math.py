"""Index arithmetic the kernels rely on, restated on the CPU (no library call): cheap to check exhaustively here, expensive to debug
on the device."""
import numpy as np
import pytest


def test_bitonic_stages_of_stride_up_to_64_stay_inside_a_wave_s_blocks():
    """rowwise.h: tk_bitonic -- 256 threads walk the pair index t = tid, tid + 256, ...; a wave holds 64 consecutive t.  For strides
    <= 64 every slot a wave touches in one pass of the t loop lies in [128 (t // 64), 128 (t // 64) + 128): consecutive stages of
    that kind need no workgroup barrier between them (only the wave's own LDS ordering).  For larger strides that is false."""
    for P in (64, 128, 256, 1024, 4096, 8192):
        t = np.arange(P // 2)
        for stride in (1, 2, 4, 8, 16, 32, 64):
            if stride > P // 2:
                continue
            lo = (t // stride) * 2 * stride + (t % stride)
            hi = lo + stride
            blk = t // 64
            assert np.all(lo // 128 == blk) and np.all(hi // 128 == blk), (P, stride)
        if P >= 512:
            stride = 128
            lo = (t // stride) * 2 * stride + (t % stride)
            assert not np.all((lo + stride) // 128 == t // 64)


def test_p_image_swizzle_serves_both_read_patterns_without_conflicts():
    """skinny.h: the [128 rows][128 contexts] bf16 image of a P tile (sk_dc_unit_f / sk_bwdp_kernel).  16-byte chunk c of row r sits at
    slot ((c >> 1) ^ mswz(r)) << 1 | (c & 1), mswz(r) = (r & 3) | ((r >> 3) & 1) << 2.  (a) ds_read_b128 of the dQ product: the lane
    groups {rows 0-3, 12-15 with chunk K} + {rows 4-11 with chunk K + 1} (K a multiple of 4) must hit 16 distinct slots; (b) the
    transposed 8-byte reads of the dC product: rows k..k+3 and k+8..k+11 (k a multiple of 8, or 8-aligned + 4) of one 32-byte group
    must hit 8 distinct 32-byte slots.  The map is an involution (the DMA fills with it, the readers invert it)."""
    def mswz(r):
        return (r & 3) | (((r >> 3) & 1) << 2)

    def slot(r, c):
        return (((c >> 1) ^ mswz(r)) << 1) | (c & 1)

    for r in range(128):
        assert sorted(slot(r, c) for c in range(16)) == list(range(16))
        assert all(slot(r, slot(r, c)) == c for c in range(16))
    for base in range(0, 128, 16):
        for K in (0, 4, 8, 12):
            rows_a = [base + i for i in (0, 1, 2, 3, 12, 13, 14, 15)]
            rows_b = [base + i for i in range(4, 12)]
            slots = [slot(r, K) for r in rows_a] + [slot(r, K + 1) for r in rows_b]
            assert len(set(slots)) == 16, (base, K)
    for k0 in range(0, 128, 8):
        for off in (0, 4):
            k = k0 + off
            rows = [k + i for i in range(4)] + [((k + 8) % 128) + i for i in range(4)] if off == 0 else None
            if rows is None:
                continue
            for g in range(8):
                groups = [g ^ mswz(r) for r in rows]
                assert len(set(groups)) == 8, (k, g)


def test_dq_slices_of_the_plan_without_the_dscores_launch_cover_every_step_once():
    """dprhot.hip: sk_fused_plan -- nslices slices of ksteps 64-context steps; a slice touches at most SK_FT = 8 statistics tiles
    ((ksteps + 1) / 2 + 1 <= 8), and where the few-rows plan's own slices are longer the form cuts 13-step slices of its own."""
    SK_FT = 8
    for nk in range(9, 258):
        for ns in (8, 10, 16):
            ks = -(-nk // ns)
            n = ns
            if (ks + 1) // 2 + 1 > SK_FT:
                ks = 2 * (SK_FT - 1) - 1
                n = -(-nk // ks)
            assert (ks + 1) // 2 + 1 <= SK_FT and n * ks >= nk  # (trailing slices may be empty: their units write zero slabs)
            covered = np.zeros(nk, int)
            for s in range(n):
                lo, hi = s * ks, min((s + 1) * ks, nk)
                if hi > lo:
                    covered[lo:hi] += 1
                    assert (hi - 1) // 2 - lo // 2 + 1 <= SK_FT
            assert np.all(covered == 1)


# ---------------------------------------------------------------------------------------------------------------------
# csrc/wideselect.h (top-k for any k, state in HBM): the selection rule restated on the CPU.  The kernel finds the key T of the
# k-th best by four radix passes over an order-preserving 32-bit key (most significant byte first, each pass counting only the
# entries that share the prefix found so far), then -- when only some of the entries AT T are taken -- the id of the last one
# taken by the same radix select over the ties' ids (smallest first: run_retrieval_pytorch.py's order is value descending, ties by
# lower id).  An entry is selected iff key > T, or key == T and id <= id_cut.
# ---------------------------------------------------------------------------------------------------------------------
def _wsel_key(v):
    u = np.asarray(v, dtype=np.float32).view(np.uint32).copy()
    u[u == 0x80000000] = 0  # -0 ties with +0
    neg = (u & 0x80000000) != 0
    return np.where(neg, ~u, u | np.uint32(0x80000000)).astype(np.uint32)


def _wsel_select(values, ids, k):
    keys = _wsel_key(values)
    need = min(len(keys), k)
    if need == 0:
        return np.zeros(len(keys), dtype=bool)
    prefix, ngt = np.uint32(0), 0
    for p in range(4):
        shift = 24 - 8 * p
        live = np.ones(len(keys), dtype=bool) if p == 0 else (keys >> np.uint32(shift + 8)) == (prefix >> np.uint32(shift + 8))
        hist = np.bincount(((keys[live] >> np.uint32(shift)) & np.uint32(255)).astype(np.int64), minlength=256)
        cum, b = 0, 255
        while b > 0 and cum + hist[b] < need:
            cum += hist[b]
            b -= 1
        ngt += cum
        need -= cum
        neq = hist[b]
        prefix = prefix | np.uint32(b << shift)
    T = prefix
    id_cut = np.iinfo(np.int64).max
    if need < neq:
        tie_ids = ids[keys == T].astype(np.uint64)
        top = 7
        while top > 0 and not np.any((tie_ids >> np.uint64(8 * top)) & np.uint64(255)):
            top -= 1
        idp, want = np.uint64(0), need
        for byte in range(top, -1, -1):
            sh = np.uint64(8 * byte)
            live = np.ones(len(tie_ids), dtype=bool) if byte == 7 else (tie_ids >> (sh + np.uint64(8))) == (idp >> (sh + np.uint64(8)))
            hist = np.bincount(((tie_ids[live] >> sh) & np.uint64(255)).astype(np.int64), minlength=256)
            cum, b = 0, 0
            while b < 255 and cum + hist[b] < want:
                cum += hist[b]
                b += 1
            want -= cum
            idp = idp | (np.uint64(b) << sh)
        id_cut = int(idp)
    sel = (keys > T) | ((keys == T) & (ids <= id_cut))
    assert int((keys > T).sum()) == ngt and int(sel.sum()) == min(len(keys), k)
    return sel


def test_wide_selection_key_is_order_preserving_and_ties_zeroes():
    v = np.array([-np.inf, -3.4e38, -1.0, -1e-45, -0.0, 0.0, 1e-45, 1.0, 3.4e38, np.inf], dtype=np.float32)
    k = _wsel_key(v)
    assert k[4] == k[5]  # -0 == +0
    assert np.all(np.diff(k.astype(np.int64)) >= 0) and np.all(np.diff(np.delete(k, 4).astype(np.int64)) > 0)
    rng = np.random.default_rng(0)
    x = rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 20000).astype(np.float32)
    order = np.argsort(x, kind="stable")
    assert np.all(np.diff(_wsel_key(x[order]).astype(np.int64)) >= 0)


@pytest.mark.parametrize("seed", range(6))
def test_wide_selection_rule_equals_the_stable_sort_prefix(seed):
    rng = np.random.default_rng(seed)
    for case in range(40):
        n = int(rng.integers(1, 4000))
        k = int(rng.integers(1, 5000))
        mode = case % 5
        if mode == 0:
            v = rng.standard_normal(n).astype(np.float32)
        elif mode == 1:
            v = rng.integers(0, int(rng.integers(1, 6)), n).astype(np.float32)  # most entries tie at the k-th value
        elif mode == 2:
            v = np.where(rng.random(n) < 0.5, -np.inf, rng.standard_normal(n)).astype(np.float32)
        elif mode == 3:
            v = np.where(rng.random(n) < 0.5, 0.0, -0.0).astype(np.float32)
        else:
            v = np.full(n, 7.0, dtype=np.float32)
        # ids as the kernel meets them: distinct, not in order, some beyond 2^32 (passage ids of a sharded corpus)
        ids = rng.permutation(n).astype(np.int64) * int(rng.integers(1, 5)) + int(rng.choice([0, 1 << 20, (1 << 33) + 5]))
        sel = _wsel_select(v, ids, k)
        order = np.lexsort((ids, -v.astype(np.float64)))  # value descending (+0 == -0), then id ascending
        want = np.zeros(n, dtype=bool)
        want[order[:k]] = True
        assert np.array_equal(sel, want), (seed, case, n, k, mode)
