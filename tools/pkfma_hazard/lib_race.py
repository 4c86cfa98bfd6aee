import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voicefixer_amd import ops, packing, _lib
dev = "cuda"
g = torch.Generator().manual_seed(5)
act1 = ops.Act(pre=_lib.PRE_LRELU, pre_slope=0.01, post=_lib.POST_LRELU, post_slope=0.01)
def mk(B, Cin, Cout, L, dil, seed, x3=True, res=False):
    gg = torch.Generator().manual_seed(seed)
    x = ops.guarded(B, Cin, L, 2187 + 264, dev); x._vfx_base.zero_(); x[:, :, :L] = torch.randn((B, Cin, L), generator=gg).to(dev)
    wp = packing.pack_conv1d(torch.randn((Cout, Cin, 3), generator=gg) * (Cin * 3) ** -0.5)
    w, w3, b = wp.to(dev), packing.pack_x3(wp).to(dev), torch.zeros(Cout, device=dev)
    y = torch.zeros((B, Cout, (L + 3) // 4 * 4), device=dev)
    def fn():
        ops.conv1d(x, w, b, y, L, 3, dil, 0, act1, y if res else None, w3=w3 if x3 else None)
    return fn
def victim(Cin, Cout, H, lp):
    P = 1 << lp
    x = ops.guarded(1, Cin, H * P, P + 1 + 264, dev); x._vfx_base.zero_()
    xv = torch.randn((1, Cin, H, P), generator=g); xv[..., P - 1] = 0
    x[:, :, :H * P] = xv.reshape(1, Cin, H * P).to(dev)
    wp = packing.pack_conv2d(torch.randn((Cout, Cin, 3, 3), generator=g) * (9 * Cin) ** -0.5).to(dev)
    sc = (0.8 + 0.4 * torch.rand(Cin, generator=g)).to(dev); sh = (0.3 * torch.randn(Cin, generator=g)).to(dev)
    act = ops.Act(pre=_lib.PRE_AFFINE_LRELU, pre_slope=0.01, scale=sc, shift=sh, post=_lib.POST_LRELU, post_slope=0.01)
    def fn():
        y = torch.empty((1, Cout, H * P), device=dev)
        ops.conv2d(x, wp, None, y, H, lp, 3, act, None)
        return y
    return fn
vic = victim(32, 32, 832, 7)
ref = vic(); torch.cuda.synchronize()
bgs = {"x3 tapsplit 64x256 C64 d729": mk(1, 64, 64, 311346, 729, 8), "x3 mode0 64x256 C64 d1 L311k": mk(1, 64, 64, 311346, 1, 6),
       "x3 mode0 128x128 C128 d1 res": mk(1, 128, 128, 103782, 1, 9, True, True)}
sb = torch.cuda.Stream()
for name, bg in bgs.items():
    bg(); torch.cuda.synchronize()
    nbad = n = 0
    for rep in range(8):
        with torch.cuda.stream(sb):
            for _ in range(40): bg()
        outs = [vic() for _ in range(40)]
        torch.cuda.synchronize()
        nbad += sum(int(not torch.equal(o, ref)) for o in outs); n += len(outs)
    print("%-40s: victim corrupted %d / %d" % (name, nbad, n))
