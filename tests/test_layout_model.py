"""CPU model of the lane/register bookkeeping used by the HIP kernels (no GPU needed).

The kernels in dreamvla_amd/csrc rely on the v_mfma_f32_32x32x16_bf16 fragment layout documented in
/opt/skills/guides/cdna_hip_programming.md section 3:

  a operand : lane l holds A[i = l & 31][slot (g = l >> 5, j = 0..7)]
  b operand : lane l holds B[slot (g, j)][jcol = l & 31]
  c / d     : lane l, register r holds C[i = (r & 3) + 8 * (r >> 2) + 4 * g][jcol = l & 31]

and on the fact that the hardware contracts slot (g, j) of `a` with slot (g, j) of `b`, so ANY assignment of
k values to slots is valid as long as both operands use the same one.  This file re-implements, in numpy,
the index helpers of gemm.hip / attention.hip (stage_store, frag_load, frag_pi, acc_row, pack order) on top
of that model and checks that the data flow computes the intended matrix products.  It pins the *design*;
the `-m gpu` parity tests pin the compiled kernels.
"""
import numpy as np
import pytest

LANES = np.arange(64)
L31 = LANES & 31
G = LANES >> 5


def acc_row(r, g):
    return (r & 3) + 8 * (r >> 2) + 4 * g


def mfma_32x32x16(a, b, c):
    """a, b: [64 lanes][8 slots]; c: [64][16].  Returns d with the documented c/d layout."""
    A = np.zeros((32, 2, 8))
    B = np.zeros((2, 8, 32))
    for l in range(64):
        A[l & 31, l >> 5, :] = a[l]
        B[l >> 5, :, l & 31] = b[l]
    C = np.einsum("igj,gjn->in", A, B)
    d = c.copy()
    for l in range(64):
        for r in range(16):
            d[l, r] += C[acc_row(r, l >> 5), l & 31]
    return d


# ---------------------------------------------------------------------------------------------------
# GEMM tile model (gemm.hip): 128x128x64 tile, both operand layouts
# ---------------------------------------------------------------------------------------------------
BM, BK = 128, 64


def rm_off(row, octet):  # element offset of k-octet `octet` of row `row` in the swizzled row-major image (gemm.hip rm_off / 2)
    return row * 64 + ((octet ^ ((row >> 1) & 7)) << 3)


def gemm_stage_rm(tile):  # tile[row][k] (128 x 64)  ->  LDS element array (row-major, padded)
    lds = np.zeros(BM * BK)
    for t in range(256):
        r, kc = t >> 3, (t & 7) * 8
        for i in range(4):
            rr = r + 32 * i
            o = rm_off(rr, t & 7)
            lds[o:o + 8] = tile[rr, kc:kc + 8]
    return lds


def gemm_stage_dma(tile):  # LDS-DMA image: wave w, chunk c = 4w+i, lane -> LDS byte (c*1024 + lane*16), global octet swizzled
    lds = np.zeros(BM * BK)
    for wave in range(4):
        for i in range(4):
            c = wave * 4 + i
            for lane in range(64):
                rl = c * 8 + (lane >> 3)
                octet = (lane & 7) ^ ((rl >> 1) & 7)
                dst = (c * 1024 + lane * 16) // 2
                lds[dst:dst + 8] = tile[rl, octet * 8: octet * 8 + 8]
    return lds


def gemm_stage_quad(tile):  # same logical tile, memory is [k][row]; LDS 8-byte units [k/4][128] -> model as (unit, 4)
    lds = np.zeros((16 * BM, 4))
    for t in range(256):
        kq, r0 = t >> 4, (t & 15) * 8
        vec = [tile[r0:r0 + 8, 4 * kq + kk] for kk in range(4)]    # four 16-B vectors: 8 rows at k = 4kq+kk
        # register transpose exactly as stage_store<true>: dword w of a vector = rows (r0+2w, r0+2w+1)
        for w in range(4):
            lo = [vec[kk][2 * w] for kk in range(4)]
            hi = [vec[kk][2 * w + 1] for kk in range(4)]
            lds[kq * BM + r0 + 2 * w] = lo       # o[4w+0], o[4w+1] = {k0,k1},{k2,k3} of row r0+2w
            lds[kq * BM + r0 + 2 * w + 1] = hi   # o[4w+2], o[4w+3]
    return lds


def gemm_frag_rm(lds, row_base, ks):
    f = np.zeros((64, 8))
    for l in range(64):
        row, g = row_base + (l & 31), l >> 5
        off = rm_off(row, ks * 2 + g)
        f[l] = lds[off:off + 8]
    return f


def test_gemm_dma_image_equals_register_staged_image():
    rng = np.random.default_rng(3)
    tile = rng.integers(-9, 10, (BM, BK)).astype(np.float64)
    np.testing.assert_array_equal(gemm_stage_dma(tile), gemm_stage_rm(tile))


def test_gemm_rm_image_bank_conflict_free():
    # ds_read_b128 is served in 4 groups of 16 lanes; within a group every 4-dword access must hit distinct banks (64 banks)
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for base in (0, 32, 64, 96):
        for ks in range(4):
            for g in range(2):
                for grp in groups:
                    banks = set()
                    for l in grp:
                        dw = rm_off(base + l, ks * 2 + g) * 2 // 4
                        for d in range(4):
                            banks.add((dw + d) % 64)
                    assert len(banks) == 64


def gemm_frag_quad(lds, row_base, ks):
    f = np.zeros((64, 8))
    for l in range(64):
        row, g = row_base + (l & 31), l >> 5
        q0 = ks * 4 + g * 2
        f[l, 0:4] = lds[q0 * BM + row]
        f[l, 4:8] = lds[(q0 + 1) * BM + row]
    return f


@pytest.mark.parametrize("a_t,b_t", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_tile_dataflow(a_t, b_t):
    rng = np.random.default_rng(0)
    A = rng.integers(-4, 5, (BM, BK)).astype(np.float64)   # A[m][k]
    B = rng.integers(-4, 5, (BM, BK)).astype(np.float64)   # B[n][k]
    la = gemm_stage_quad(A) if a_t else gemm_stage_rm(A)
    lb = gemm_stage_quad(B) if b_t else gemm_stage_rm(B)
    fa = gemm_frag_quad if a_t else gemm_frag_rm
    fb = gemm_frag_quad if b_t else gemm_frag_rm
    C = np.zeros((BM, BM))
    for wave in range(4):
        wm, wn = wave & 1, wave >> 1
        acc = [[np.zeros((64, 16)) for _ in range(2)] for _ in range(2)]
        for ks in range(BK // 16):
            for i in range(2):
                for j in range(2):
                    acc[i][j] = mfma_32x32x16(fb(lb, wn * 64 + i * 32, ks), fa(la, wm * 64 + j * 32, ks), acc[i][j])
        for i in range(2):
            for j in range(2):
                for l in range(64):
                    for r in range(16):
                        m = wm * 64 + 32 * j + (l & 31)
                        n = wn * 64 + 32 * i + 8 * (r >> 2) + 4 * (l >> 5) + (r & 3)
                        C[m, n] = acc[i][j][l, r]
    np.testing.assert_array_equal(C, A @ B.T)


# ---------------------------------------------------------------------------------------------------
# attention tile model (attention.hip)
# ---------------------------------------------------------------------------------------------------
RM72 = 72


def at_stage_rm(tile):  # tile[32 rows][64]
    lds = np.zeros(32 * RM72)
    for u in range(128):
        pr, oc = u >> 3, u & 7
        for row in (2 * pr, 2 * pr + 1):
            lds[row * RM72 + oc * 8: row * RM72 + oc * 8 + 8] = tile[row, oc * 8: oc * 8 + 8]
    return lds


def at_stage_pi(tile):
    lds = np.zeros((16 * 64, 2))
    for u in range(128):
        pr, oc = u >> 3, u & 7
        for i in range(8):
            lds[pr * 64 + oc * 8 + i, 0] = tile[2 * pr, oc * 8 + i]
            lds[pr * 64 + oc * 8 + i, 1] = tile[2 * pr + 1, oc * 8 + i]
    return lds


def at_frag_rm(lds, s):
    f = np.zeros((64, 8))
    for l in range(64):
        off = (l & 31) * RM72 + s * 16 + (l >> 5) * 8
        f[l] = lds[off:off + 8]
    return f


def at_frag_pi(lds, db, mm):
    f = np.zeros((64, 8))
    for l in range(64):
        c, g = 32 * db + (l & 31), l >> 5
        base = (8 * mm + 2 * g) * 64 + c
        for w, off in enumerate((0, 64, 4 * 64, 5 * 64)):
            f[l, 2 * w], f[l, 2 * w + 1] = lds[base + off, 0], lds[base + off, 1]
    return f


def direct_frag(mat, s):  # register fragments loaded straight from global: row = l&31, cols 16s+8g..+7
    f = np.zeros((64, 8))
    for l in range(64):
        f[l] = mat[l & 31, 16 * s + 8 * (l >> 5): 16 * s + 8 * (l >> 5) + 8]
    return f


def test_attention_forward_tile_dataflow():
    rng = np.random.default_rng(1)
    Q = rng.integers(-2, 3, (32, 64)).astype(np.float64)
    K = rng.integers(-2, 3, (32, 64)).astype(np.float64)
    V = rng.integers(-2, 3, (32, 64)).astype(np.float64)
    k_rm, v_pi = at_stage_rm(K), at_stage_pi(V)
    sacc = np.zeros((64, 16))
    for s in range(4):
        sacc = mfma_32x32x16(at_frag_rm(k_rm, s), direct_frag(Q, s), sacc)
    # lane l: query l&31, register r: key acc_row(r, g)
    S = Q @ K.T
    for l in range(64):
        for r in range(16):
            assert sacc[l, r] == S[l & 31, acc_row(r, l >> 5)]
    P = sacc  # use raw scores as "probabilities" (integers -> exact)
    oacc = [np.zeros((64, 16)), np.zeros((64, 16))]
    for db in range(2):
        oacc[db] = mfma_32x32x16(at_frag_pi(v_pi, db, 0), P[:, 0:8], oacc[db])
        oacc[db] = mfma_32x32x16(at_frag_pi(v_pi, db, 1), P[:, 8:16], oacc[db])
    O = S @ V
    for l in range(64):
        for db in range(2):
            for r in range(16):
                assert oacc[db][l, r] == O[l & 31, 32 * db + acc_row(r, l >> 5)]


def test_attention_backward_tile_dataflow():
    rng = np.random.default_rng(2)
    Q = rng.integers(-2, 3, (32, 64)).astype(np.float64)
    K = rng.integers(-2, 3, (32, 64)).astype(np.float64)
    V = rng.integers(-2, 3, (32, 64)).astype(np.float64)
    dO = rng.integers(-2, 3, (32, 64)).astype(np.float64)
    # ---- dQ kernel orientation: lane = query
    k_rm, k_pi, v_rm = at_stage_rm(K), at_stage_pi(K), at_stage_rm(V)
    sacc, dpacc = np.zeros((64, 16)), np.zeros((64, 16))
    for s in range(4):
        sacc = mfma_32x32x16(at_frag_rm(k_rm, s), direct_frag(Q, s), sacc)
        dpacc = mfma_32x32x16(at_frag_rm(v_rm, s), direct_frag(dO, s), dpacc)
    S, dP = Q @ K.T, dO @ V.T
    for l in range(64):
        for r in range(16):
            assert dpacc[l, r] == dP[l & 31, acc_row(r, l >> 5)]
    dS = sacc * dpacc  # any elementwise combination keeps the layout
    dq = [np.zeros((64, 16)), np.zeros((64, 16))]
    for db in range(2):
        dq[db] = mfma_32x32x16(at_frag_pi(k_pi, db, 0), dS[:, 0:8], dq[db])
        dq[db] = mfma_32x32x16(at_frag_pi(k_pi, db, 1), dS[:, 8:16], dq[db])
    dQ = (S * dP) @ K
    for l in range(64):
        for db in range(2):
            for r in range(16):
                assert dq[db][l, r] == dQ[l & 31, 32 * db + acc_row(r, l >> 5)]
    # ---- dK/dV kernel orientation: lane = key, registers = queries
    q_rm, q_pi, do_rm, do_pi = at_stage_rm(Q), at_stage_pi(Q), at_stage_rm(dO), at_stage_pi(dO)
    sacc, dpacc = np.zeros((64, 16)), np.zeros((64, 16))
    for s in range(4):
        sacc = mfma_32x32x16(at_frag_rm(q_rm, s), direct_frag(K, s), sacc)
        dpacc = mfma_32x32x16(at_frag_rm(do_rm, s), direct_frag(V, s), dpacc)
    for l in range(64):
        for r in range(16):
            assert sacc[l, r] == S[acc_row(r, l >> 5), l & 31]
            assert dpacc[l, r] == dP[acc_row(r, l >> 5), l & 31]
    P, dSr = sacc, sacc * dpacc
    dv = [np.zeros((64, 16)), np.zeros((64, 16))]
    dk = [np.zeros((64, 16)), np.zeros((64, 16))]
    for db in range(2):
        for mm in range(2):
            dv[db] = mfma_32x32x16(at_frag_pi(do_pi, db, mm), P[:, 8 * mm: 8 * mm + 8], dv[db])
            dk[db] = mfma_32x32x16(at_frag_pi(q_pi, db, mm), dSr[:, 8 * mm: 8 * mm + 8], dk[db])
    dV, dK = S.T @ dO, (S * dP).T @ Q
    for l in range(64):
        for db in range(2):
            for r in range(16):
                assert dv[db][l, r] == dV[l & 31, 32 * db + acc_row(r, l >> 5)]
                assert dk[db][l, r] == dK[l & 31, 32 * db + acc_row(r, l >> 5)]


# ---------------------------------------------------------------------------------------------------
# attention ring kernels (attention.hip, attn_*_ring_kernel): ONE swizzled row-major LDS-DMA image per 32 x 64 tile
# serves row fragments (ds_read_b128) and transposed fragments (ds_read_b64_tr_b16)
# ---------------------------------------------------------------------------------------------------
def fa_sw(r):
    return (((r >> 1) & 1) << 2) | ((r >> 2) & 3)


def fa_dma_image(tile):
    """tile[32][64] -> LDS byte image as written by the 4 DMA pieces (wave w: rows 8w..8w+7, lane -> row 8w + lane/8,
    LDS slot lane % 8 holding global d-octet (lane % 8) ^ fa_sw(row)); modelled as elements (2 B each)."""
    lds = np.zeros(32 * 64)
    for w in range(4):
        for lane in range(64):
            rl = 8 * w + (lane >> 3)
            octet = (lane & 7) ^ fa_sw(rl)
            dst = (w * 1024 + lane * 16) // 2
            lds[dst:dst + 8] = tile[rl, octet * 8: octet * 8 + 8]
    return lds


def fa_frag_rm(lds, s):
    f = np.zeros((64, 8))
    for l in range(64):
        row, g = l & 31, l >> 5
        off = (row * 128 + (((2 * s + g) ^ fa_sw(row)) << 4)) // 2
        f[l] = lds[off:off + 8]
    return f


def tr_read_b64(lds, byte_addr):
    """ds_read_b64_tr_b16 as probed on hardware (tests/probes/tr_probe.hip): inside each 16-lane group lane i receives
    element (i & 3) of the four 8-byte chunks FETCHED by lanes 4 j + (i >> 2), j = 0..3."""
    out = np.zeros((64, 4))
    for l in range(64):
        base, i = l & ~15, l & 15
        for j in range(4):
            src = base + 4 * j + (i >> 2)
            out[l, j] = lds[byte_addr[src] // 2 + (i & 3)]
    return out


def fa_frag_tr(lds, db, mm):
    f = np.zeros((64, 8))
    for h in range(2):
        addr = np.zeros(64, dtype=np.int64)
        for l in range(64):
            gi, c = l >> 4, l & 15
            d = 32 * db + 16 * (gi & 1) + 4 * (c & 3)
            row = 16 * mm + 8 * h + 4 * (gi >> 1) + (c >> 2)
            addr[l] = row * 128 + (((d >> 3) ^ fa_sw(row)) << 4) + (d & 7) * 2
        f[:, 4 * h: 4 * h + 4] = tr_read_b64(lds, addr)
    return f


def test_attention_ring_tile_dataflow():
    rng = np.random.default_rng(11)
    Q = rng.integers(-2, 3, (32, 64)).astype(np.float64)
    K = rng.integers(-2, 3, (32, 64)).astype(np.float64)
    V = rng.integers(-2, 3, (32, 64)).astype(np.float64)
    k_img, v_img = fa_dma_image(K), fa_dma_image(V)
    sacc = np.zeros((64, 16))
    for s in range(4):
        sacc = mfma_32x32x16(fa_frag_rm(k_img, s), direct_frag(Q, s), sacc)
    S = Q @ K.T
    for l in range(64):
        for r in range(16):
            assert sacc[l, r] == S[l & 31, acc_row(r, l >> 5)]
    oacc = [np.zeros((64, 16)), np.zeros((64, 16))]
    for db in range(2):
        oacc[db] = mfma_32x32x16(fa_frag_tr(v_img, db, 0), sacc[:, 0:8], oacc[db])
        oacc[db] = mfma_32x32x16(fa_frag_tr(v_img, db, 1), sacc[:, 8:16], oacc[db])
    O = S @ V
    for l in range(64):
        for db in range(2):
            for r in range(16):
                assert oacc[db][l, r] == O[l & 31, 32 * db + acc_row(r, l >> 5)]
    # dQ^T = K^T . dS^T uses the transposed fragments of the SAME K image
    dq = [np.zeros((64, 16)), np.zeros((64, 16))]
    for db in range(2):
        dq[db] = mfma_32x32x16(fa_frag_tr(k_img, db, 0), sacc[:, 0:8], dq[db])
        dq[db] = mfma_32x32x16(fa_frag_tr(k_img, db, 1), sacc[:, 8:16], dq[db])
    DQ = S @ K
    for l in range(64):
        for db in range(2):
            for r in range(16):
                assert dq[db][l, r] == DQ[l & 31, 32 * db + acc_row(r, l >> 5)]


def test_attention_ring_image_bank_conflict_free():
    # row fragments: ds_read_b128, 4 groups of 16 lanes, 64 banks of 4 B
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for s in range(4):
        for g in range(2):
            for grp in groups:
                banks = set()
                for row in grp:
                    dw = (row * 128 + (((2 * s + g) ^ fa_sw(row)) << 4)) // 4
                    banks.update((dw + d) % 64 for d in range(4))
                assert len(banks) == 64
    # transposed fragments: ds_read_b64_tr_b16, 2 groups of 32 lanes, 8 B per lane
    for db in range(2):
        for mm in range(2):
            for h in range(2):
                for half in range(2):
                    banks = set()
                    for l in range(32 * half, 32 * half + 32):
                        gi, c = l >> 4, l & 15
                        d = 32 * db + 16 * (gi & 1) + 4 * (c & 3)
                        row = 16 * mm + 8 * h + 4 * (gi >> 1) + (c >> 2)
                        dw = (row * 128 + (((d >> 3) ^ fa_sw(row)) << 4) + (d & 7) * 2) // 4
                        banks.update((dw + x) % 64 for x in range(2))
                    assert len(banks) == 64
