#!/usr/bin/env python
"""Rounding error of the three evaluations of a k = 3 convolution in fp32 -- direct sum, Winograd F(2,3), Winograd
F(4,3) -- against a float64 reference (CPU, torch; the arithmetic of convwg_kernel / convwg4_kernel restated with the same
transform constants and the same operation order, weights transformed in float64 and rounded once like packing.pack_wino /
pack_wino4).  Backs the figures quoted in DESIGN.md 3.0b.

    python tools/winograd_error.py [channels] [positions]          # the Gaussian layer of round 2
    python tools/winograd_error.py --sweep [channels] [positions]  # adversarial operand statistics (round 3)

The sweep answers the round-2 review: the 8.2e-7 figure was measured on N(0, sigma) weights only, while real checkpoints
are weight-normed (w = g v / |v| with heavy-tailed per-row gains g) and real activations are neither centred nor of one
scale.  Cases: log-normal row gains (sigma 1, 2 -- a per-row factor cannot change a row's RELATIVE error, shown for
completeness), Student-t(2) weight entries, smooth / alternating filters (taps nearly equal or nearly cancelling: the
cases where G w loses digits), activations with per-channel log-normal scale (sigma 2), with a DC offset 30x their spread
(post-leaky-ReLU maps are mostly positive: B^T d cancels), and slowly varying inputs (neighbouring taps nearly equal).
Error measures per case: rms error / rms y, and max over outputs of |error| / (|w| * |x|)(q) -- the latter is the
quantity a forward error bound controls (error <= c eps sum |w_k||x_k|) and does not reward a small |y|."""
import sys

import torch


def evaluate(w, x, n):
    """w [co][ci][3], x [ci][n + 5] (already activated) -> dict name -> (y, ) for positions 0..n-1 (y[q] = sum_k w_k x[q+k])."""
    c_out, c_in = w.shape[0], w.shape[1]
    w64, x64 = w.double(), x.double()
    taps = lambda t, i: t[:, i:i + n]
    ref = sum(w64[:, :, k] @ taps(x64, k) for k in range(3))
    mag = sum(w64[:, :, k].abs() @ taps(x64, k).abs() for k in range(3))    # sum |w||x| per output
    direct = sum(w[:, :, k] @ taps(x, k) for k in range(3))

    G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    U2 = torch.einsum("pk,oik->poi", G2, w64).float()
    xe = x[:, 0:n + 3]
    d = [xe[:, i:i + n:2] for i in range(4)]
    V2 = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]
    m = [U2[k] @ V2[k] for k in range(4)]
    f23 = torch.empty((c_out, n))
    f23[:, 0::2] = (m[0] + m[1]) + m[2]
    f23[:, 1::2] = (m[1] - m[2]) - m[3]

    G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                       [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
    U4 = torch.einsum("pk,oik->poi", G4, w64).float()
    e = [x[:, i:i + n:4] for i in range(6)]
    s1, s2 = e[4] - 4 * e[2], e[3] - 4 * e[1]
    s3, s4 = e[4] - e[2], 2 * (e[3] - e[1])
    V4 = [4 * e[0] - 5 * e[2] + e[4], s1 + s2, s1 - s2, s3 + s4, s3 - s4, 4 * e[1] - 5 * e[3] + e[5]]
    m = [U4[k] @ V4[k] for k in range(6)]
    p12, m12, p34, m34 = m[1] + m[2], m[1] - m[2], m[3] + m[4], m[3] - m[4]
    f43 = torch.empty((c_out, n))
    f43[:, 0::4] = (m[0] + p12) + p34
    f43[:, 1::4] = 2 * m34 + m12
    f43[:, 2::4] = 4 * p34 + p12
    f43[:, 3::4] = 8 * m34 + m12 + m[5]
    out = {}
    for name, y in (("direct", direct), ("F(2,3)", f23), ("F(4,3)", f43)):
        err = y.double() - ref
        row_rel = (err.pow(2).mean(1).sqrt() / ref.pow(2).mean(1).sqrt().clamp_min(1e-300))   # per output row
        out[name] = {"rms_rel": (err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(),
                     "max_rel": err.abs().max().item() / ref.abs().max().item(),
                     "worst_row_rms_rel": row_rel.max().item(),
                     "max_over_mag": (err.abs() / mag.clamp_min(1e-300)).max().item()}
    return out


def cases(c, n, g):
    randn = lambda *s: torch.randn(s, generator=g)
    base_w = randn(c, c, 3) * (3 * c) ** -0.5
    base_x = torch.nn.functional.leaky_relu(randn(c, n + 5), 0.01)
    yield "gaussian w, lrelu(gaussian) x  [round 2]", base_w, base_x
    for sig in (1.0, 2.0):
        v = randn(c, c, 3)
        gain = torch.exp(sig * randn(c, 1, 1))
        yield "weight-norm, log-normal row gains sigma=%g" % sig, gain * v / v.reshape(c, -1).norm(dim=1).reshape(c, 1, 1), base_x
    t2 = randn(c, c, 3) / (randn(c, c, 3).pow(2) + randn(c, c, 3).pow(2)).div(2).sqrt().clamp_min(1e-3)
    yield "Student-t(2) weight entries", t2 * (3 * c) ** -0.5, base_x
    smooth = randn(c, c, 1) * (3 * c) ** -0.5 * (1 + 1e-3 * randn(c, c, 3))
    yield "smooth filters (taps equal to 1e-3)", smooth, base_x
    alt = smooth * torch.tensor([1.0, -2.0, 1.0])
    yield "second-difference filters (1,-2,1)(1 +- 1e-3)", alt, base_x
    yield "per-channel log-normal activation scale sigma=2", base_w, base_x * torch.exp(2.0 * randn(c, 1))
    yield "DC offset 30x the spread (all-positive maps)", base_w, torch.nn.functional.leaky_relu(30.0 + randn(c, n + 5), 0.01)
    slow = torch.cumsum(randn(c, n + 5), 1) * 0.05 + randn(c, 1) * 5
    yield "slowly varying inputs (random walk + offset)", base_w, torch.nn.functional.leaky_relu(slow, 0.01)
    yield "second-difference filters on slowly varying inputs", alt, torch.nn.functional.leaky_relu(slow, 0.01)
    spikes = base_x.clone()
    spikes[:, ::97] *= 1e4
    yield "activation outliers 1e4x every 97th position", base_w, spikes


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    c = int(args[0]) if len(args) > 0 else 256
    n = int(args[1]) if len(args) > 1 else 4096
    g = torch.Generator().manual_seed(0)
    if "--sweep" not in sys.argv:
        w = torch.randn((c, c, 3), generator=g) * (3 * c) ** -0.5
        x = torch.nn.functional.leaky_relu(torch.randn((c, n + 5), generator=g), 0.01)
        r = evaluate(w, x, n)
        print("k = 3 convolution, %d -> %d channels, %d positions, fp32 against float64 (max |error| / max |y|, rms error / rms y)" % (c, c, n))
        for name, label in (("direct", "direct sum"), ("F(2,3)", "Winograd F(2,3)"), ("F(4,3)", "Winograd F(4,3)")):
            print("  %-16s %.2e   %.2e" % (label, r[name]["max_rel"], r[name]["rms_rel"]))
        return
    print("k = 3 convolution, %d -> %d channels, %d positions, fp32 against float64" % (c, c, n))
    print("columns per algorithm: rms error / rms y | worst output ROW's rms error / rms y | max |error| / sum|w||x|")
    worst = {"direct": 0.0, "F(2,3)": 0.0, "F(4,3)": 0.0}
    for label, w, x in cases(c, n, g):
        r = evaluate(w.float(), x.float(), n)
        print("%-52s" % label + "".join("  %s %.1e %.1e %.1e" % (k, r[k]["rms_rel"], r[k]["worst_row_rms_rel"], r[k]["max_over_mag"])
                                        for k in ("direct", "F(2,3)", "F(4,3)")))
        for k in worst:
            worst[k] = max(worst[k], r[k]["max_over_mag"])
    print("worst max |error| / sum|w||x| over all cases: " + ", ".join("%s %.2e" % kv for kv in worst.items()))
    print("(fp32 eps = 6.0e-8; a K-term dot product accumulated in fp32 is bounded by ~K eps and behaves like sqrt(K) eps)")


if __name__ == "__main__":
    main()
