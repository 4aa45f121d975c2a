"""dec4.conv2 (16 -> 16 @512^2, B = 32) on the lean fp32 kernel: plain store + statistics vs the fused BatchNorm-backward
sums, with y read from the SAME tensor as the input (cache-warm) or from its own — where do the extra 260 us come from?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deadtrees_amd import ops

B, H, W, Cc = 32, 512, 512, int(os.environ.get("C", "16"))
g = torch.Generator().manual_seed(0)
x = torch.randn((B, H, W, Cc), generator=g).cuda()
y = torch.randn((B, H, W, Cc), generator=g).cuda()
w = (torch.randn((3, 3, Cc, Cc), generator=g) * 0.05).cuda()
mu, istd, sc, sh = [torch.randn(Cc, generator=g).cuda() for _ in range(4)]


def clock(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = torch.empty_like(x)
print("plain + stats      %.1f us" % clock(lambda: ops.conv2d(x, w, 3, 1, 1, want_stats=True, out0=out)))
print("plain, no stats    %.1f us" % clock(lambda: ops.conv2d(x, w, 3, 1, 1, out0=out)))
print("fused sums, y own  %.1f us" % clock(lambda: ops.conv2d_bn_bwd(x, w, y, mu, istd, act_scale=sc, act_shift=sh)))
print("fused sums, y = x  %.1f us" % clock(lambda: ops.conv2d_bn_bwd(x, w, x, mu, istd, act_scale=sc, act_shift=sh)))
