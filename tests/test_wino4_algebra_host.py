"""CPU checks of the Winograd F(4x4,3x3) algebra the round-3 kernels implement (csrc/conv_wino4.hip, conv_wino4_wgrad.hip):
the transform matrices, the gradient-direction identity, the frequency-column pairing of the input transform, and the
image-pair argument for 16x16 maps.  The kernels themselves are checked on the GPU (tests/kernel_checks.py: wino4*,
wino4_wgrad*, wino4_pair*)."""
import numpy as np

BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
               [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
              [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def conv_same(x, w):
    """x [Ci][H][W], w [Co][Ci][3][3] -> y [Co][H][W] (stride 1, zero padding 1; cross-correlation as nn.Conv2d)"""
    Ci, H, W = x.shape
    xp = np.zeros((Ci, H + 2, W + 2))
    xp[:, 1:-1, 1:-1] = x
    y = np.zeros((w.shape[0], H, W))
    for r in range(3):
        for c in range(3):
            y += np.einsum("oc,chw->ohw", w[:, :, r, c], xp[:, r:r + H, c:c + W])
    return y


def tiles(xp, H, W):
    """6x6 patches of the zero-padded input, one per 4x4 output tile: [th][tw][Ci][6][6]"""
    return np.array([[xp[:, 4 * i:4 * i + 6, 4 * j:4 * j + 6] for j in range(W // 4)] for i in range(H // 4)])


def test_forward_identity():
    rng = np.random.default_rng(0)
    Ci, Co, H, W = 3, 4, 8, 12
    x, w = rng.standard_normal((Ci, H, W)), rng.standard_normal((Co, Ci, 3, 3))
    xp = np.zeros((Ci, H + 2, W + 2))
    xp[:, 1:-1, 1:-1] = x
    U = np.einsum("ir,ocrs,js->ocij", G, w, G)                 # U = G g G^T
    V = np.einsum("ir,abcrs,js->abcij", BT, tiles(xp, H, W), BT)  # V = B^T d B
    M = np.einsum("ocij,abcij->aboij", U, V)
    Y = np.einsum("pi,aboij,qj->abopq", AT, M, AT)             # Y = A^T M A
    y = Y.transpose(2, 0, 3, 1, 4).reshape(Co, H, W)
    assert np.abs(y - conv_same(x, w)).max() < 1e-10


def test_weight_gradient_identity():
    """dg = G^T [ sum over tiles (A dY A^T) . (B^T d B) ] G   (conv_wino4_wgrad.hip header)"""
    rng = np.random.default_rng(1)
    Ci, Co, H, W = 2, 3, 8, 16
    x, dy = rng.standard_normal((Ci, H, W)), rng.standard_normal((Co, H, W))
    xp = np.zeros((Ci, H + 2, W + 2))
    xp[:, 1:-1, 1:-1] = x
    # direct: dw[o][c][r][s] = sum_hw dy[o][h][w] * xp[c][h + r][w + s]
    dw = np.array([[[[(dy[o] * xp[c, r:r + H, s:s + W]).sum() for s in range(3)] for r in range(3)] for c in range(Ci)]
                   for o in range(Co)])
    A = AT.T
    dyt = np.array([[dy[:, 4 * i:4 * i + 4, 4 * j:4 * j + 4] for j in range(W // 4)] for i in range(H // 4)])
    Mg = np.einsum("ip,abopq,jq->aboij", A, dyt, A)            # A dY A^T (4x4 -> 6x6)
    V = np.einsum("ir,abcrs,js->abcij", BT, tiles(xp, H, W), BT)
    dU = np.einsum("aboij,abcij->ocij", Mg, V)
    dg = np.einsum("ir,ocij,js->ocrs", G, dU, G)               # G^T dU G
    assert np.abs(dg - dw).max() < 1e-9


def test_frequency_column_pairs_share_partial_sums():
    """the input transform by column pairs (conv_wino4.hip / conv_wino4_wgrad.hip): (1,2) = (d4 - 4 d2) +- (d3 - 4 d1),
    (3,4) = (d4 - d2) +- 2 (d3 - d1), (0,5) from columns 0..4 / 1..5; pairs (1,2) and (3,4) never read patch columns 0, 5"""
    rng = np.random.default_rng(2)
    d = rng.standard_normal(6)
    t = BT @ d
    a, b = d[4] - 4 * d[2], d[3] - 4 * d[1]
    assert np.allclose([t[1], t[2]], [a + b, a - b])
    a, b = d[4] - d[2], d[3] - d[1]
    assert np.allclose([t[3], t[4]], [a + 2 * b, a - 2 * b])
    assert np.allclose([t[0], t[5]], [4 * d[0] - 5 * d[2] + d[4], 4 * d[1] - 5 * d[3] + d[5]])
    assert not BT[1:5, [0, 5]].any()
    # output / gradient transform A (4 -> 6): {d0, e + o, e - o, e' + 2 o', e' - 2 o', d3}
    v = rng.standard_normal(4)
    e, o, e2, o2 = v[0] + v[2], v[1] + v[3], v[0] + 4 * v[2], v[1] + 4 * v[3]
    assert np.allclose(AT.T @ v, [v[0], e + o, e - o, e2 + 2 * o2, e2 - 2 * o2, v[3]])


def test_image_pair_seam():
    """16x16 maps run as image pairs: two images side by side in one 16 x 32 block.  Only the (0,5) frequency columns read
    across the seam (patch column 5 of tile column 3, patch column 0 of tile column 4); zeroing those two reads gives
    exactly the two separate zero-padded convolutions."""
    rng = np.random.default_rng(3)
    Ci, Co, H = 2, 3, 16
    xa, xb, w = rng.standard_normal((Ci, H, H)), rng.standard_normal((Ci, H, H)), rng.standard_normal((Co, Ci, 3, 3))
    xp = np.zeros((Ci, H + 2, 2 * H + 2))
    xp[:, 1:-1, 1:H + 1], xp[:, 1:-1, H + 1:2 * H + 1] = xa, xb
    P = tiles(xp, H, 2 * H).copy()                             # [4][8][Ci][6][6]
    P[:, 3, :, :, 5] = 0.0                                     # tile column 3: patch column 5 is image b's column 0
    P[:, 4, :, :, 0] = 0.0                                     # tile column 4: patch column 0 is image a's column 15
    U = np.einsum("ir,ocrs,js->ocij", G, w, G)
    V = np.einsum("ir,abcrs,js->abcij", BT, P, BT)
    Y = np.einsum("pi,aboij,qj->abopq", AT, np.einsum("ocij,abcij->aboij", U, V), AT)
    y = Y.transpose(2, 0, 3, 1, 4).reshape(Co, H, 2 * H)
    assert np.abs(y[:, :, :H] - conv_same(xa, w)).max() < 1e-9
    assert np.abs(y[:, :, H:] - conv_same(xb, w)).max() < 1e-9
