"""Per-layer A/B of the Winograd kernel against the direct fp32 kernel on the U-Net's 3x3 stride-1 shapes (B=32)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deadtrees_amd import ops

B = int(os.environ.get("B", "32"))
SHAPES = [  # name, H, W (stored, source 0), C0, C1, mode0, Cout
    ("layer1 64->64 @128", 128, 128, 64, 0, 0, 64),
    ("layer2 128->128 @64", 64, 64, 128, 0, 0, 128),
    ("layer3 256->256 @32", 32, 32, 256, 0, 0, 256),
    ("layer4 512->512 @16", 16, 16, 512, 0, 0, 512),
    ("dec0.c1 up512+256->256 @32", 16, 16, 512, 256, 1, 256),
    ("dec1.c1 up256+128->128 @64", 32, 32, 256, 128, 1, 128),
    ("dec2.c1 up128+64->64 @128", 64, 64, 128, 64, 1, 64),
    ("dec3.c1 dgrad 32->128 @256", 256, 256, 32, 0, 0, 128),
]
reps = int(os.environ.get("REPS", "5"))
only = os.environ.get("ONLY")
for name, H, W, C0, C1, mode0, Cout in SHAPES:
    if only and only not in name:
        continue
    g = torch.Generator().manual_seed(0)
    src0 = torch.randn((B, H, W, C0), generator=g).cuda()
    Hin, Win = (2 * H, 2 * W) if mode0 else (H, W)
    src1 = torch.randn((B, Hin, Win, C1), generator=g).cuda() if C1 else None
    w = (torch.randn((3, 3, C0 + C1, Cout), generator=g) * 0.05).cuda()
    u = ops.winograd_weights(w)
    flops = 2.0 * 9 * (C0 + C1) * Cout * Hin * Win * B
    res = {"layer": name}
    for tf in (False, True):
        sc = torch.ones(C0).cuda() if tf else None
        sh = torch.zeros(C0).cuda() if tf else None
        out = {}
        for kind in ("direct", "wino"):
            def run(out0=None):
                if kind == "direct":
                    return ops.conv2d(src0, w, 3, 1, 1, src1=src1, mode0=mode0, want_stats=True, out0=out0,
                                      in_scale=sc, in_shift=sh)
                return ops.conv2d_winograd(src0, u, src1=src1, mode0=mode0, want_stats=True, out0=out0, in_scale=sc,
                                           in_shift=sh)
            o0 = run()[0]
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run(o0)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / reps * 1e-3)
            t = min(ts)
            out[kind] = o0
            res[f"{kind}{'_tf' if tf else ''}_us"] = round(t * 1e6, 1)
            res[f"{kind}{'_tf' if tf else ''}_TF"] = round(flops / t / 1e12, 1)
        res[f"maxdiff{'_tf' if tf else ''}"] = float((out["direct"] - out["wino"]).abs().max() / out["direct"].abs().max())
    if C0 % 64 == 0 and C1 % 64 == 0 and Cout % 64 == 0:
        dy = torch.randn((B, Hin, Win, Cout), generator=g).cuda()
        for kind in ("direct", "wino"):
            def runw():
                if kind == "direct":
                    return ops.conv2d_wgrad(src0, dy, 3, 1, 1, src1=src1, mode0=mode0)
                return ops.conv2d_wgrad_winograd(src0, dy, src1=src1, mode0=mode0)
            out[kind] = runw()
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    runw()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / reps * 1e-3)
            res[f"wgrad_{kind}_us"] = round(min(ts) * 1e6, 1)
            res[f"wgrad_{kind}_TF"] = round(flops / min(ts) / 1e12, 1)
        res["wgrad_maxdiff"] = float((out["direct"] - out["wino"]).abs().max() / out["direct"].abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.winograd_weights(w)
    e1.record()
    torch.cuda.synchronize()
    res["wtransform_us"] = round(e0.elapsed_time(e1) / reps * 1e3, 1)
    print(json.dumps(res), flush=True)
