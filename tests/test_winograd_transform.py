"""The Winograd F(2x2, 3x3) algebra that ``csrc/conv_wino_bf16x6_kernel.h`` (conv2's forward) is built on, restated in
numpy and checked against the direct 3x3 cross-correlation ``nn.Conv2d`` computes (models/models.py:15-17, padding 1):

* the transform matrices and their orientation (``V = B^T d B``, ``U = G g G^T``, ``Y = A^T (U . V) A``);
* the kernel's bookkeeping: column ``j = 3`` of ``U`` is stored negated so that it accumulates in place, the column inverse
  transform is ``W[i][0] = M[i][0] + M[i][1] + M[i][2]``, ``W[i][1] = M[i][1] - M[i][2] + M'[i][3]``, and the two frequency
  halves (transform rows 0, 1 and 2, 3) contribute ``Y[0] = W0 + W1 | W2`` and ``Y[1] = W1 | -W2 - W3``;
* the staging split: a thread of half ``fh`` loads patch rows ``fh .. fh + 2`` only.

The GPU parity tests (tests/test_model_gpu.py, tests/test_train_gpu.py) exercise the kernel itself; this file pins the
convention, on the CPU.
"""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def direct_tile(d, g):
    """2x2 outputs of the 3x3 cross-correlation over a 4x4 patch: y[a][b] = sum_kl d[a+k][b+l] g[k][l]."""
    y = np.zeros((2, 2))
    for a in range(2):
        for b in range(2):
            y[a, b] = np.sum(d[a:a + 3, b:b + 3] * g)
    return y


def test_transforms_reproduce_the_correlation():
    rng = np.random.default_rng(0)
    for _ in range(20):
        d, g = rng.standard_normal((4, 4)), rng.standard_normal((3, 3))
        y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
        np.testing.assert_allclose(y, direct_tile(d, g), rtol=0, atol=1e-12)


def test_kernel_bookkeeping_over_channels():
    """Sum over input channels per frequency, negated j = 3 weights, column transform on the accumulators, row transform
    per frequency half, halves added -- the order of operations of the kernel."""
    rng = np.random.default_rng(1)
    cin = 5
    d = rng.standard_normal((cin, 4, 4))
    g = rng.standard_normal((cin, 3, 3))
    want = sum(direct_tile(d[c], g[c]) for c in range(cin))

    u = np.stack([G @ g[c] @ G.T for c in range(cin)])
    u[:, :, 3] *= -1.0                                       # prep_conv_w_wino_bf16x3_elem: if (j == 3) u = -u
    # staging: half fh loads patch rows fh .. fh + 2 and produces transform rows 2 fh, 2 fh + 1
    v = np.zeros((cin, 4, 4))
    for c in range(cin):
        for fh in range(2):
            l0, l1, l2 = d[c, fh], d[c, fh + 1], d[c, fh + 2]
            r = (l0 - l2, l1 + l2) if fh == 0 else (l1 - l0, l0 - l2)
            for ii in range(2):
                rr = r[ii]
                v[c, 2 * fh + ii] = [rr[0] - rr[2], rr[1] + rr[2], rr[2] - rr[1], rr[1] - rr[3]]
    np.testing.assert_allclose(v, np.stack([BT @ d[c] @ BT.T for c in range(cin)]), atol=1e-12)

    m = np.einsum("cij,cij->ij", u, v)                       # per-frequency GEMMs (here: dot products over the channels)
    w = np.zeros((4, 2))
    for i in range(4):
        w[i, 0] = m[i, 0] + m[i, 1] + m[i, 2]                # j = 0 in place, j = 1, 2 through the scratch accumulator
        w[i, 1] = m[i, 1] - m[i, 2] + m[i, 3]                # j = 3 accumulates in place: its weights carry the minus sign
    y_half0 = np.stack([w[0] + w[1], w[1]])                  # wf = 0: rows 0, 1
    y_half1 = np.stack([w[2], -w[2] - w[3]])                 # wf = 1: rows 2, 3
    np.testing.assert_allclose(y_half0 + y_half1, want, atol=1e-12)


def test_block_geometry_covers_every_output_once():
    """16 x 2 tiles per workgroup, grid ceil(ceil(W / 2) / 2): every output pixel of an H x W map belongs to exactly one
    (block, tile, a, b); odd widths leave the last tile half outside (the kernel's gx < W guard)."""
    for h, w in ((32, 100), (32, 47), (32, 6), (32, 1)):
        seen = np.zeros((h, w), dtype=int)
        nbx = ((w + 1) // 2 + 1) // 2
        for by in range((h + 31) // 32):
            for bx in range(nbx):
                for tm in range(32):
                    ty, tx = 16 * by + (tm >> 1), 2 * bx + (tm & 1)
                    for a in range(2):
                        for b in range(2):
                            gy, gx = 2 * ty + a, 2 * tx + b
                            if gy < h and gx < w:
                                seen[gy, gx] += 1
        assert (seen == 1).all(), (h, w)
