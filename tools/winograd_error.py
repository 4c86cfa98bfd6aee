#!/usr/bin/env python
"""Rounding error of the three evaluations of a k = 3 convolution in fp32 -- direct sum, Winograd F(2,3), Winograd
F(4,3) -- against a float64 reference, on one ResStack-like layer (CPU, torch; the arithmetic of convwg_kernel /
convwg4_kernel restated with the same transform constants, weights transformed in float64 and rounded once like
packing.pack_wino / pack_wino4).  Backs the figures quoted in DESIGN.md 3.0b.

    python tools/winograd_error.py [channels] [positions]
"""
import sys

import torch


def main():
    c = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    g = torch.Generator().manual_seed(0)
    w = torch.randn((c, c, 3), generator=g) * (3 * c) ** -0.5          # [co][ci][tap]
    x = torch.nn.functional.leaky_relu(torch.randn((c, n + 5), generator=g), 0.01)
    w64, x64 = w.double(), x.double()
    taps = lambda t, i: t[:, i:i + n]                                   # x[q + i] for q = 0..n-1 (dilation 1 without loss of generality)
    ref = sum(w64[:, :, k] @ taps(x64, k) for k in range(3))            # y[q] = sum_k w_k x[q + k]
    scale = ref.abs().max().item()

    direct = sum(w[:, :, k] @ taps(x, k) for k in range(3))

    # F(2,3): pairs (q, q+1), inputs d0..d3 = x[q..q+3]
    G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    U2 = torch.einsum("pk,oik->poi", G2, w64).float()                   # [4][co][ci]
    xe = x[:, 0:n + 3]
    d = [xe[:, i:i + n:2] for i in range(4)]                            # pairs start at even q
    V2 = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]
    m = [U2[k] @ V2[k] for k in range(4)]
    f23 = torch.empty((c, n))
    f23[:, 0::2] = (m[0] + m[1]) + m[2]
    f23[:, 1::2] = (m[1] - m[2]) - m[3]

    # F(4,3): quads (q..q+3), inputs d0..d5 = x[q..q+5]
    G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                       [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
    U4 = torch.einsum("pk,oik->poi", G4, w64).float()
    e = [x[:, i:i + n:4] for i in range(6)]
    s1, s2 = e[4] - 4 * e[2], e[3] - 4 * e[1]
    s3, s4 = e[4] - e[2], 2 * (e[3] - e[1])
    V4 = [4 * e[0] - 5 * e[2] + e[4], s1 + s2, s1 - s2, s3 + s4, s3 - s4, 4 * e[1] - 5 * e[3] + e[5]]
    m = [U4[k] @ V4[k] for k in range(6)]
    p12, m12, p34, m34 = m[1] + m[2], m[1] - m[2], m[3] + m[4], m[3] - m[4]
    f43 = torch.empty((c, n))
    f43[:, 0::4] = (m[0] + p12) + p34
    f43[:, 1::4] = 2 * m34 + m12
    f43[:, 2::4] = 4 * p34 + p12
    f43[:, 3::4] = 8 * m34 + m12 + m[5]

    print("k = 3 convolution, %d -> %d channels, %d positions, fp32 against float64 (max |error| / max |y|, rms error / rms y)" % (c, c, n))
    for name, y in (("direct sum", direct), ("Winograd F(2,3)", f23), ("Winograd F(4,3)", f43)):
        err = (y.double() - ref)
        print("  %-16s %.2e   %.2e" % (name, err.abs().max().item() / scale, (err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()))


if __name__ == "__main__":
    main()
