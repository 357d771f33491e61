"""CPU check of the algebra behind csrc/srt_nn4.hip: the 5x5 stride-2 transposed convolution of the decoder
(Executable/spleeter.c:239-294 via conv_transpose2d) equals, per output parity class (py, px), Winograd minimal filtering over
2x2 input blocks - F(2,3) along an axis with 3 taps (k = 4, 2, 0), F(2,2) along one with 2 taps (k = 3, 1) - with the transform
matrices, tap order, patch origin (rows a0-1..a0+2, columns b0-1..b0+2) and point count (16 + 12 + 12 + 9 = 49) the kernel uses.
Pure numpy, float64, tiny sizes; the GPU tests hold the kernel itself against the oracle."""
import numpy as np

B3 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)      # 4 points from d0..d3
B2 = np.array([[1, -1, 0, 0], [0, 1, 0, 0], [0, 1, -1, 0]], float)                      # 3 points from d0..d2 (d3 unused)
G3 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], float)               # taps (g[-1], g[0], g[+1])
G2 = np.array([[1, 0], [1, 1], [0, 1]], float)                                          # taps (g[-1], g[0])
A3 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)
A2 = np.array([[1, 1, 0], [0, 1, -1]], float)
BT, GT, AT = {1: B3, 0: B2}, {1: G3, 0: G2}, {1: A3, 0: A2}


def taps(p):                        # (input shift d, kernel index k = p + 1 - 2 d)
    return [(-1, 4), (0, 2), (1, 0)] if p == 1 else [(-1, 3), (0, 1)]


def direct(x, w):
    cin, cout, H, W = x.shape[0], w.shape[1], x.shape[1], x.shape[2]
    y = np.zeros((cout, 2 * H, 2 * W))
    for h in range(H):
        for ww in range(W):
            for ky in range(5):
                for kx in range(5):
                    Y, X = 2 * h + ky - 1, 2 * ww + kx - 1
                    if 0 <= Y < 2 * H and 0 <= X < 2 * W:
                        y[:, Y, X] += w[:, :, ky, kx].T @ x[:, h, ww]
    return y


def winograd(x, w):
    cin, cout, H, W = x.shape[0], w.shape[1], x.shape[1], x.shape[2]
    xp = np.zeros((cin, H + 3, W + 3))
    xp[:, 1:H + 1, 1:W + 1] = x
    y = np.zeros((cout, 2 * H, 2 * W))
    npts = 0
    for py in (1, 0):
        for px in (1, 0):
            g = np.stack([np.stack([w[:, :, ky, kx] for (_, kx) in taps(px)], -1) for (_, ky) in taps(py)], -2)
            U = np.einsum('ik,cokl,jl->coij', GT[py], g, GT[px])
            npts += U.shape[2] * U.shape[3]
            for a0 in range(0, H, 2):
                for b0 in range(0, W, 2):
                    V = np.einsum('ik,ckl,jl->cij', BT[py], xp[:, a0:a0 + 4, b0:b0 + 4], BT[px])
                    Yb = np.einsum('ik,okl,jl->oij', AT[py], np.einsum('coij,cij->oij', U, V), AT[px])
                    for da in range(2):
                        for db in range(2):
                            y[:, 2 * (a0 + da) + py, 2 * (b0 + db) + px] = Yb[:, da, db]
    return y, npts


def test_transposed_conv_equals_winograd_classes():
    rng = np.random.default_rng(7)
    for (cin, cout, H, W) in ((3, 2, 6, 8), (1, 1, 2, 4), (2, 3, 4, 4)):
        x = rng.standard_normal((cin, H, W))
        w = rng.standard_normal((cin, cout, 5, 5))
        y, npts = winograd(x, w)
        assert npts == 49
        assert np.abs(y - direct(x, w)).max() < 1e-12


# ------------------------------------------------------------------------------------------- encoder (csrc/srt_nn5.hip)
# The stride-2 5x5 TF-SAME convolution (spleeter.c:182-238 via conv2d, pad 1 before / 2 after) seen from the input's parity planes:
# output oy reads the ODD rows 2(oy-1)+1, 2oy+1, 2(oy+1)+1 with taps ky = 0, 2, 4 and the EVEN rows 2oy, 2(oy+1) with taps ky = 1, 3.
# Over blocks of 2x2 OUTPUT pixels that is F(2,3) on the odd plane and F(2,2) on the even plane per axis: 16 + 12 + 12 + 9 = 49 products,
# and the four classes ADD into the same 2x2 outputs.  Same B / G / A matrices as the decoder; the taps are taken in ascending order.
def enc_taps(p):                    # kernel indices of the class, in the order the G matrices expect
    return [0, 2, 4] if p == 1 else [1, 3]


def enc_direct(x, w):
    cin, H, W = x.shape
    cout = w.shape[0]
    y = np.zeros((cout, H // 2, W // 2))
    for oy in range(H // 2):
        for ox in range(W // 2):
            for ky in range(5):
                for kx in range(5):
                    Y, X = 2 * oy + ky - 1, 2 * ox + kx - 1
                    if 0 <= Y < H and 0 <= X < W:
                        y[:, oy, ox] += w[:, :, ky, kx] @ x[:, Y, X]
    return y


def enc_winograd(x, w):
    cin, H, W = x.shape
    cout = w.shape[0]
    xp = np.zeros((cin, H + 8, W + 8))                      # input row r at xp row r + 1 (row -1 = the pad-before row)
    xp[:, 1:H + 1, 1:W + 1] = x
    y = np.zeros((cout, H // 2, W // 2))
    npts = 0
    for py in (1, 0):                                       # plane parity along y: 1 = odd input rows (3 taps), 0 = even (2 taps)
        for px in (1, 0):
            g = np.stack([np.stack([w[:, :, ky, kx] for kx in enc_taps(px)], -1) for ky in enc_taps(py)], -2)     # [co][ci][ny][nx]
            U = np.einsum('ik,ockl,jl->ocij', GT[py], g, GT[px])
            npts += U.shape[2] * U.shape[3]
            for ya in range(H // 4):                        # block = outputs 2ya..2ya+1
                for xb in range(W // 4):
                    # patch of the block: input rows 4ya-1 .. 4ya+5 (xp rows 4ya .. 4ya+6); the odd plane is every second row from the
                    # first, the even plane every second row from the second
                    r0, c0 = 4 * ya, 4 * xb
                    rows = [r0 + 2 * i for i in range(4)] if py else [r0 + 1 + 2 * i for i in range(4)]
                    cols = [c0 + 2 * j for j in range(4)] if px else [c0 + 1 + 2 * j for j in range(4)]
                    d = xp[:, rows][:, :, cols]
                    V = np.einsum('ik,ckl,jl->cij', BT[py], d, BT[px])
                    Yb = np.einsum('ik,okl,jl->oij', AT[py], np.einsum('ocij,cij->oij', U, V), AT[px])
                    y[:, 2 * ya:2 * ya + 2, 2 * xb:2 * xb + 2] += Yb
    return y, npts


def test_strided_conv_equals_winograd_over_parity_planes():
    rng = np.random.default_rng(11)
    for (cin, cout, H, W) in ((3, 2, 8, 12), (1, 1, 4, 4), (2, 3, 4, 8)):
        x = rng.standard_normal((cin, H, W))
        w = rng.standard_normal((cout, cin, 5, 5))
        y, npts = enc_winograd(x, w)
        assert npts == 49
        assert np.abs(y - enc_direct(x, w)).max() < 1e-12


def _wino_unit_xy(sp, tilesX, tilesY, tpw, walk=True):
    """csrc/srt_nn4.hip wino_sp_xy + the kernels' choice of `colrun`: which spatial tile the unit `sp` of an instance is"""
    colrun = 1 if not walk else (tpw if tilesY % tpw == 0 else (tilesY if tpw % tilesY == 0 else 1))
    if colrun > 1:
        run, step = divmod(sp, colrun)
        return run % tilesX, (run // tilesX) * colrun + step
    return sp % tilesX, sp // tilesX


def test_column_walk_of_the_winograd_units_is_a_bijection():
    """The Winograd kernels hand a workgroup `tpw` consecutive units; with the column walk they are tiles of one tile column (so that x neighbours
    are read by neighbouring workgroups at the same time).  Whatever the geometry, every spatial tile must be visited exactly once, the units
    of a workgroup must stay inside one instance, and with the walk on they must form a vertical run."""
    for tilesX in range(1, 10):
        for tilesY in range(1, 18):
            nsp = tilesX * tilesY
            for tpw in (1, 2, 4, 8):
                if nsp % tpw:                                   # the launcher only picks a tpw that divides the units of an instance group
                    continue
                seen = {_wino_unit_xy(sp, tilesX, tilesY, tpw) for sp in range(nsp)}
                assert seen == {(x, y) for x in range(tilesX) for y in range(tilesY)}, (tilesX, tilesY, tpw)
                if tilesY % tpw == 0 and tpw > 1:
                    for wg in range(nsp // tpw):
                        cells = [_wino_unit_xy(wg * tpw + i, tilesX, tilesY, tpw) for i in range(tpw)]
                        assert len({x for x, _ in cells}) == 1 and [y for _, y in cells] == list(range(cells[0][1], cells[0][1] + tpw)), (tilesX, tilesY, tpw, cells)
                        if wg + 1 < nsp // tpw and (wg + 1) % tilesX:      # the next workgroup walks the column to the right, same rows
                            nxt = _wino_unit_xy((wg + 1) * tpw, tilesX, tilesY, tpw)
                            assert nxt == (cells[0][0] + 1, cells[0][1]), (tilesX, tilesY, tpw)


def test_up6_stream_ring_schedule():
    """csrc/srt_nn.hip srt_up6_stream_kernel: in interval i the MFMA waves write the tap rows of chunk i (image rows CR i ..) into a ring of 2 CR + 2
    rows while the gather waves read the rows g - 1 .. g + 1 of the input rows g = CR (i - 1) - 1 + a0 - with ONE barrier per interval.  Replays the
    schedule on row numbers: what the gather reads must be the right rows, written in EARLIER intervals, and never a slot this interval's MFMA writes."""
    CR, RR = 2, 6
    for H in (2, 4, 6, 32, 33, 64, 130):
        nchunks = (H + CR - 1) // CR + 1
        ring = [None] * RR                                   # None = zero-initialised, i.e. "a row outside the image"
        slot0 = 0
        emitted = []
        for i in range(nchunks + 1):
            writes = {}
            if i < nchunks:
                for row in range(CR):
                    writes[(slot0 + row) % RR] = CR * i + row
            for a0 in range(CR):
                g = CR * (i - 1) - 1 + a0
                if 0 <= g < H:
                    sm = (slot0 - CR - 2 + a0) % RR
                    for d, want in zip((sm, (sm + 1) % RR, (sm + 2) % RR), (g - 1, g, g + 1)):
                        assert d not in writes, (H, i, g)
                        have = ring[d]
                        if want < 0:
                            assert have is None, (H, i, g, have)      # above the image: the zero-initialised slot, not yet reused
                        else:
                            assert have == want, (H, i, g, have, want)   # (rows >= H were written too: chunks of zeros from the DMA's range check)
                    emitted.append(g)
            for s, r in writes.items():
                ring[s] = r
            slot0 = (slot0 + CR) % RR
        assert emitted == list(range(H)), (H, emitted[:8])
