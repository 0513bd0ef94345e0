"""Index arithmetic the kernels rely on, restated on the CPU (no library call): cheap to check exhaustively here, expensive to debug
on the device."""
import numpy as np


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
