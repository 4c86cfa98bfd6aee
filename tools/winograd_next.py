#!/usr/bin/env python
"""Groundwork for the two convolution families that still run direct sums: CPU checks of the algebra
and of the tile geometry BEFORE any kernel is written.  Nothing here is on the product path.

  python tools/winograd_next.py f32      F(3,2) for the two taps per phase of the polyphase ConvTranspose1d
                                          (voicefixer/vocoder/model/modules.py:449-459,519): matrices, exactness in
                                          float64, fp32 rounding error against the direct sum
  python tools/winograd_next.py tile     F(4,3) along the DILATED axis inside one LDS tile of the fused C = 64 layer
                                          (modules.py:592-609): block geometry per dilation, MFMA column use, exactness
"""
import sys

import numpy as np


# ---- F(3,2): y_i = g0 d_i + g1 d_{i+1}, i = 0..2, from d_0..d_3 with 4 products (points 0, 1, -1, inf) ----------
BT32 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, -1, 0, 1]], dtype=np.float64)
G32 = np.array([[1, 0], [0.5, 0.5], [0.5, -0.5], [0, 1]], dtype=np.float64)
AT32 = np.array([[1, 1, 1, 0], [0, 1, -1, 0], [0, 1, 1, 1]], dtype=np.float64)


def f32_exact():
    """Bilinear identity on the basis: A^T[(G g) * (B^T d)] == correlation for every unit g, d."""
    worst = 0.0
    for a in range(2):
        for b in range(4):
            g = np.zeros(2); g[a] = 1
            d = np.zeros(4); d[b] = 1
            y = AT32 @ ((G32 @ g) * (BT32 @ d))
            ref = np.array([g[0] * d[i] + g[1] * d[i + 1] for i in range(3)])
            worst = max(worst, np.abs(y - ref).max())
    return worst


def f32_error(cin=512, cout=256, n=3 * 1024, seed=0):
    """One phase of a transposed convolution = a 2-tap correlation per (co, ci): fp32 direct vs fp32 F(3,2) vs float64."""
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((cout, cin, 2)) * (2 * cin) ** -0.5)
    x = rng.standard_normal((cin, n + 1))
    x = x + np.sin(x)                                   # the stage's pre-activation (modules.py:516)
    ref = np.einsum("oc,cn->on", w[:, :, 0], x[:, :-1]) + np.einsum("oc,cn->on", w[:, :, 1], x[:, 1:])
    w32, x32 = w.astype(np.float32), x.astype(np.float32)
    direct = (w32[:, :, 0] @ x32[:, :-1] + w32[:, :, 1] @ x32[:, 1:]).astype(np.float64)
    # F(3,2): triples of outputs 3t .. 3t+2 share x[3t .. 3t+3]
    U = np.einsum("kj,ocj->koc", G32, w).astype(np.float32)             # transformed on the host in float64, stored fp32
    nt = n // 3
    d = np.stack([x32[:, j:j + 3 * nt:3] for j in range(4)], 0)         # (4, cin, nt)
    V = np.einsum("kj,jcn->kcn", BT32.astype(np.float32), d).astype(np.float32)
    M = np.stack([U[k] @ V[k] for k in range(4)], 0)                    # fp32 GEMMs = the MFMA accumulations
    Y = np.einsum("ik,kon->oni", AT32.astype(np.float32), M).reshape(cout, 3 * nt).astype(np.float64)
    rms = np.sqrt(np.mean(ref[:, :3 * nt] ** 2))
    e_d = np.sqrt(np.mean((direct[:, :3 * nt] - ref[:, :3 * nt]) ** 2)) / rms
    e_w = np.sqrt(np.mean((Y - ref[:, :3 * nt]) ** 2)) / rms
    return e_d, e_w


# ---- F(4,3) along the dilated axis inside one tile ------------------------------------------------------------------
BT43 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                 [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G43 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
AT43 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def tile_geometry(d, width=256):
    """A tile whose first-phase outputs are whole blocks of 4d positions: block b covers [4d b, 4d (b+1)), quad
    (b, r) = outputs 4d b + r + i d (i < 4) from inputs 4d b + r + (j - 1) d (j < 6).  Returns (blocks, quads,
    first-phase columns, second-phase outputs = columns - 2, MFMA column use of the 32-wide quad blocks)."""
    nblk = max(1, width // (4 * d))
    cols = 4 * d * nblk
    quads = d * nblk
    slots = -(-quads // 32) * 32
    return nblk, quads, cols, cols - 2, quads / slots


def tile_exact(d, cin=16, cout=8, seed=1):
    """One tile, float64: dilated conv by in-tile F(4,3) -> Y in natural column order -> dilation-1 conv by F(4,3) on
    the quads of the tile that are whole, against the two direct convolutions."""
    rng = np.random.default_rng(seed)
    nblk, quads, cols, outs, _ = tile_geometry(d)
    w1 = rng.standard_normal((cout, cin, 3)); w2 = rng.standard_normal((cout, cout, 3))
    x = rng.standard_normal((cin, cols + 2 * d))                       # staged: columns [-d, cols + d) of the tile
    lre = lambda v: np.where(v > 0, v, 0.01 * v)
    xa = lre(x)
    ref1 = sum(np.einsum("oc,cn->on", w1[:, :, t], xa[:, t * d:t * d + cols]) for t in range(3))
    ya = lre(ref1)
    ref2 = sum(np.einsum("oc,cn->on", w2[:, :, t], ya[:, t:t + cols - 2]) for t in range(3))
    # phase 1: quads (b, r); column of input j of quad (b, r) in the staged tile = 4d b + r + j d
    U1 = np.einsum("kj,ocj->koc", G43, w1)
    Y = np.zeros((cout, cols))
    for b in range(nblk):
        for r in range(d):
            dv = np.stack([xa[:, 4 * d * b + r + j * d] for j in range(6)], 0)        # (6, cin)
            V = BT43 @ dv
            m = np.stack([U1[k] @ V[k] for k in range(6)], 0)                          # (6, cout)
            y = AT43 @ m                                                               # (4, cout)
            for i in range(4):
                Y[:, 4 * d * b + r + i * d] = y[i]
    e1 = np.abs(Y - ref1).max()
    # phase 2 on lrelu(Y): output quad 4Q .. 4Q+3 from columns 4Q .. 4Q+5
    U2 = np.einsum("kj,ocj->koc", G43, w2)
    ya2 = lre(Y)
    nq2 = (cols - 2) // 4
    out = np.zeros((cout, 4 * nq2))
    for Q in range(nq2):
        V = BT43 @ np.stack([ya2[:, 4 * Q + j] for j in range(6)], 0)
        out[:, 4 * Q:4 * Q + 4] = (AT43 @ np.stack([U2[k] @ V[k] for k in range(6)], 0)).T
    e2 = np.abs(out - ref2[:, :4 * nq2]).max()
    return e1, e2, 4 * nq2


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("f32", "all"):
        print("F(3,2), points 0, 1, -1, inf: 4 products per 3 outputs of a 2-tap correlation (direct: 6)")
        print("  B^T =", BT32.tolist()); print("  G   =", G32.tolist()); print("  A^T =", AT32.tolist())
        print("  bilinear identity on the basis, max |error| (float64): %.1e" % f32_exact())
        for cin, cout in ((1024, 512), (512, 256), (256, 128), (128, 64)):
            e_d, e_w = f32_error(cin, cout)
            print("  up stage %4d -> %3d channels, fp32 vs float64 (rms error / rms y): direct %.2e   F(3,2) %.2e" % (cin, cout, e_d, e_w))
    if what in ("tile", "all"):
        print("F(4,3) along the dilated axis inside one fused-layer tile (first phase = whole blocks of 4d columns)")
        print("  d   blocks  quads  columns  outputs(2nd phase, whole quads)  MFMA column use   exactness 1st / 2nd (float64)")
        for d in (1, 3, 9, 27, 81):
            nblk, quads, cols, outs, use = tile_geometry(d, 256 if d < 81 else 324)
            e1, e2, n2 = tile_exact(d) if d < 81 else (float("nan"), float("nan"), 0)
            print("  %-3d %-7d %-6d %-8d %-32d %-16.2f %.1e / %.1e" % (d, nblk, quads, cols, n2 if n2 else (cols - 2) // 4 * 4, use, e1, e2))
        print("  (d >= 81: 4d exceeds the tile -- one block per tile needs 324 columns at d = 81; the two-tile form of DESIGN 8.1)")


if __name__ == "__main__":
    main()
