"""Per-layer micro-benchmark of dt_conv2d / dt_conv2d_wgrad on the U-Net's layer shapes (B=32, 512x512 tile)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deadtrees_amd import ops

B = int(os.environ.get("B", "32"))
# (name, H, W, C0, C1, mode0, Cout, k, s, p)  — H,W = stored size of source 0
SHAPES = [
    ("stem 3->64 7x7/2", 512, 512, 3, 0, 0, 64, 7, 2, 3),
    ("layer1 64->64 @128", 128, 128, 64, 0, 0, 64, 3, 1, 1),
    ("layer2.0.c1 64->128 s2", 128, 128, 64, 0, 0, 128, 3, 2, 1),
    ("layer2 128->128 @64", 64, 64, 128, 0, 0, 128, 3, 1, 1),
    ("layer3 256->256 @32", 32, 32, 256, 0, 0, 256, 3, 1, 1),
    ("layer4 512->512 @16", 16, 16, 512, 0, 0, 512, 3, 1, 1),
    ("dec0.c1 up512+256->256 @32", 16, 16, 512, 256, 1, 256, 3, 1, 1),
    ("dec1.c1 up256+128->128 @64", 32, 32, 256, 128, 1, 128, 3, 1, 1),
    ("dec2.c1 up128+64->64 @128", 64, 64, 128, 64, 1, 64, 3, 1, 1),
    ("dec3.c1 up64+64->32 @256", 128, 128, 64, 64, 1, 32, 3, 1, 1),
    ("dec3.c2 32->32 @256", 256, 256, 32, 0, 0, 32, 3, 1, 1),
    ("dec4.c1 up32->16 @512", 256, 256, 32, 0, 1, 16, 3, 1, 1),
    ("dec4.c2 16->16 @512", 512, 512, 16, 0, 0, 16, 3, 1, 1),
    ("dec3.c1 dgrad 32->128 @256", 256, 256, 32, 0, 0, 128, 3, 1, 1),
]
only = os.environ.get("ONLY")
which = os.environ.get("WHICH", "fwd,wgrad").split(",")
reps = int(os.environ.get("REPS", "5"))
for name, H, W, C0, C1, mode0, Cout, k, s, p in SHAPES:
    if only and only not in name:
        continue
    g = torch.Generator().manual_seed(0)
    src0 = torch.randn((B, H, W, C0), generator=g).cuda()
    Hin, Win = (2 * H, 2 * W) if mode0 else (H, W)
    src1 = torch.randn((B, Hin, Win, C1), generator=g).cuda() if C1 else None
    w = (torch.randn((k, k, C0 + C1, Cout), generator=g) * 0.05).cuda()
    Ho, Wo = (Hin + 2 * p - k) // s + 1, (Win + 2 * p - k) // s + 1
    flops = 2.0 * k * k * (C0 + C1) * Cout * Ho * Wo * B
    res = {"layer": name}
    if "fwd" in which:
        out0, _, st = ops.conv2d(src0, w, k, s, p, src1=src1, mode0=mode0, want_stats=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.conv2d(src0, w, k, s, p, src1=src1, mode0=mode0, want_stats=True, out0=out0)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e-3
        res["fwd_us"] = round(t * 1e6, 1)
        res["fwd_TF"] = round(flops / t / 1e12, 1)
    if "wgrad" in which:
        dy = torch.randn((B, Ho, Wo, Cout), generator=g).cuda()
        ops.conv2d_wgrad(src0, dy, k, s, p, src1=src1, mode0=mode0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.conv2d_wgrad(src0, dy, k, s, p, src1=src1, mode0=mode0)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e-3
        res["wgrad_us"] = round(t * 1e6, 1)
        res["wgrad_TF"] = round(flops / t / 1e12, 1)
    if "bf16" in which:
        bf = torch.bfloat16
        s0, s1 = src0.to(bf), (src1.to(bf) if src1 is not None else None)
        wp = ops.pack_weights_bf16(w)
        if k in (1, 3) and C0 % 8 == 0:
            from deadtrees_amd import _lib
            lib = _lib.load()
            out0, _, _ = ops.conv2d_bf16(s0, wp, k, s, p, Cout, src1=s1, mode0=mode0, want_stats=True)
            torch.cuda.synchronize()
            # A/B in one process, interleaved rounds: register-staged kernels (0) vs the LDS-DMA kernel where it applies (2)
            times = {0: [], 2: []}
            for rnd in range(3):
                for mode in (0, 2):
                    lib.dt_set_option(b"bf16_dma", mode)
                    ops.conv2d_bf16(s0, wp, k, s, p, Cout, src1=s1, mode0=mode0, want_stats=True, out0=out0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        ops.conv2d_bf16(s0, wp, k, s, p, Cout, src1=s1, mode0=mode0, want_stats=True, out0=out0)
                    e1.record()
                    torch.cuda.synchronize()
                    times[mode].append(e0.elapsed_time(e1) / reps * 1e-3)
            lib.dt_set_option(b"bf16_dma", 1)
            t = min(times[0])
            res["bf16_fwd_us"] = round(t * 1e6, 1)
            res["bf16_fwd_TF"] = round(flops / t / 1e12, 1)
            t2 = min(times[2])
            res["bf16_dma_us"] = round(t2 * 1e6, 1)
            res["bf16_dma_TF"] = round(flops / t2 / 1e12, 1)
            dyb = torch.randn((B, Ho, Wo, Cout), generator=g).to(bf).cuda()
            ops.conv2d_wgrad_bf16(s0, dyb, k, s, p, src1=s1, mode0=mode0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.conv2d_wgrad_bf16(s0, dyb, k, s, p, src1=s1, mode0=mode0)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / reps * 1e-3
            res["bf16_wgrad_us"] = round(t * 1e6, 1)
            res["bf16_wgrad_TF"] = round(flops / t / 1e12, 1)
    print(json.dumps(res), flush=True)
