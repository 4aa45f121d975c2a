#!/usr/bin/env python3
"""bench.py — 512x512 RGB tiles/s of the U-Net training step (fwd + loss + bwd + clip + Adam) on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU, RCCL).
Rank 0 prints ONE JSON line.  Workload = BASELINE.json configs[1]: reference U-Net (smp Unet/resnet34
topology), fp32, batch 32 per GPU, 512x512x3 synthetic tiles, losses GDICE+FOCAL, clip 0.5, Adam 3e-4.

`roofline`: the dominant kernel (most GPU time among the convolution launches) timed live with HIP events on the launch
stream.  `achieved` / `frac` are fractions of a ROOF, <= 1 by construction: for the Winograd F(2x2,3x3) kernels (every
3x3 stride-1 layer with wide enough channels since round 2) `achieved` is the rate of the multiplies the kernel ISSUES on
the matrix cores (16 of every 36 algorithmic ones) and `frac` = that / 157.3 TFLOP/s (fp32 MFMA peak,
MI355X_MICROARCH.md); the direct-convolution-equivalent rate (algorithmic FLOPs of SURVEY 8d / duration, which exceeds
the peak) is kept beside it as `direct_equiv_TFLOPs`.  `roofline.kernels` lists every kernel family of the step (time
per step, share, binding roof, fraction of it), taken from two serial eager steps after the timed region.  The whole
fp32 network is compute-bound (176 FLOP/B), so the binding roof is "mfma"; the HBM fraction of the step is reported next
to it in `whole_net.hbm_frac_step`.
`cpu_baseline`: the oracle port of the reference's CPU path (oracle/train_ref.py) timed on this box's host
cores on a bounded sample (B=2, 512x512, a few steps) — rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import deadtrees_amd  # noqa: E402,F401  (first: its import sets the HIP hardware-queue count before the runtime initialises)

FLOP_PER_TILE_TRAIN = 186.53e9      # SURVEY §8d / BASELINE.md §2 (conv FLOPs fwd+bwd, 512x512x3, K=2); scaled by
BYTES_PER_TILE_TRAIN = 924.3e6      # (size/512)^2 for other tile sizes.  ideal-fused fp32 algorithmic HBM bytes
PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="tiles per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the whole training step as one HIP graph (auto: single-GPU bf16, where the step "
                         "is shorter than the Python launch path; multi-GPU runs only with --graph on)")
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--no-legs", action="store_true", help="skip the extra configs[2] (bf16, B=64) leg of the default run")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32",
                    help="fp32 = BASELINE configs[1] (headline); bf16 = configs[2] (bf16 storage, fp32 accumulate)")
    ap.add_argument("--mode", choices=["train", "infer"], default="train",
                    help="train = BASELINE configs[1] (the headline metric); infer = tiled-inference forward "
                         "(uint8 tiles -> uint8 class maps, BASELINE configs[4] per-GPU leg)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as typed: this process becomes a launcher — it starts the N ranks as a CHILD
        # torch.distributed.run (never exec) before anything here has touched the GPU, relays their output and rank 0's
        # JSON line, and exits with the child's code
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                         "(python bench.py --gpus N does that by itself when no launcher is involved)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback on the product path")
    ndev = torch.cuda.device_count()
    # DT_DIST_BACKEND=gloo lets several ranks share one GPU to rehearse the N>1 path on a 1-GPU box
    backend = os.environ.get("DT_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # DT_FORCE_DIST=1 runs the RCCL process group even at world size 1 (exercises init / broadcast / bucketed
    # all-reduce / barrier on a 1-GPU box under torch.distributed.run)
    distributed = world > 1 or os.environ.get("DT_FORCE_DIST") == "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from deadtrees_amd.network.unet import UNetHIP

    if args.mode == "infer":
        model = UNetHIP(in_channels=3, classes=2)
        model.reset_parameters(seed=0)
        model.to(dev)
        return infer_bench(args, model, dev, world, rank, distributed)

    ctx = dict(dev=dev, world=world, rank=rank, distributed=distributed)
    out = train_leg(args, ctx, args.precision, args.batch, headline=True)
    # configs[2] travels in the SAME JSON line as a leg (the headline stays configs[1]): bf16 storage / fp32 accumulate,
    # batch 64 per GPU, its own dominant-kernel roofline.  A failure of the leg must not cost the headline.
    if args.precision == "fp32" and not args.no_legs:
        try:
            leg = train_leg(args, ctx, "bf16", 64, headline=False)
            out["legs"] = {"bf16_b64": {k: leg[k] for k in ("value", "unit", "ms_per_step", "dtype", "config", "loss",
                                                            "hip_graph", "whole_net", "roofline")}}
        except Exception as e:  # noqa: BLE001
            out["legs"] = {"bf16_b64": {"error": f"{type(e).__name__}: {e}"}}
    # SURVEY 8d's secondary metric in the same line: forward-only tiles/s and km^2/h of the tiled inference path
    # (BASELINE configs[4]: 256x256 sub-tiles at batch 64, uint8 in / uint8 class map out; tiler-inclusive figure too)
    if args.precision == "fp32" and not args.no_legs and world == 1:
        try:
            import copy
            from deadtrees_amd.network.unet import UNetHIP
            ia = copy.copy(args)
            # 6.3 ms batches: enough of them that the leg is timed at steady clocks (5 + 2 right after the idle gap of the
            # model set-up read 8.8 k sub-tiles/s where `--mode infer` alone reads 10.1 k)
            ia.size, ia.batch, ia.steps, ia.warmup, ia.graph = 256, 64, 40, 10, "auto"
            im = UNetHIP(in_channels=3, classes=2)
            im.reset_parameters(seed=0)
            leg = infer_bench(ia, im.to(dev), dev, world, rank, distributed, as_leg=True)
            out.setdefault("legs", {})["infer_fp32_256"] = {k: leg[k] for k in ("metric", "value", "unit", "ms_per_step", "km2_per_hour",
                                                                                  "tiler_inclusive", "config", "whole_net", "roofline")}
            del im
        except Exception as e:  # noqa: BLE001
            out.setdefault("legs", {})["infer_fp32_256"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


def spawn_ranks(n: int) -> int:
    """launcher half of `python bench.py --gpus N`: N ranks under torch.distributed.run on 127.0.0.1 as a child process"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    line_json = None
    for line in proc.stdout:
        st = line.strip()
        if st.startswith("{") and '"metric"' in st:
            line_json = st                  # rank 0's result: printed last, alone
        else:
            sys.stderr.write(line)          # everything else the ranks said (warnings, tracebacks)
    rc = proc.wait()
    if line_json is not None:
        print(line_json, flush=True)
    elif rc == 0:
        rc = 1
    return rc


def cpu_baseline(args):
    """the oracle port of the reference's CPU training step (oracle/train_ref.py) on this box's host cores: all
    cores of the box's CPU share (>= 5 timed steps, median) and one thread (2 timed steps) — a reported baseline"""
    import torch
    from deadtrees_amd.data.synthetic import synth_batch
    from oracle.train_ref import time_cpu_baseline
    S = args.size
    cimg, cmask = synth_batch(2, S, S, 3, 2, seed=1234)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)   # a 1-GPU box's CPU share is 16 cores; more threads only oversubscribe
    steps = max(args.cpu_steps, 5)
    res = time_cpu_baseline(cimg, cmask, steps=steps, warmup=1, threads=cores)
    one = time_cpu_baseline(cimg, cmask, steps=2, warmup=1, threads=1)
    return {"value": round(res["tiles_per_s"], 3), "unit": "tiles/s", "cores": res["threads"], "kind": "port",
            "single_thread_value": round(one["tiles_per_s"], 3),
            "sample": f"oracle port of the reference CPU path (torch {torch.__version__} CPU, B=2, {S}x{S}, "
                      f"{steps} timed steps after 1 warm-up, median; single thread: 2 timed steps)"}


WINO_ISSUED = 16.0 / 36.0     # Winograd F(2x2,3x3): multiplies issued on the matrix cores per algorithmic multiply


def _hold_gpu(torch, seconds: float):
    """keep the current stream busy for about `seconds` (device-side spin, torch.cuda._sleep counts GPU clock cycles)"""
    try:
        torch.cuda._sleep(int(seconds * 2.0e9))
    except Exception:      # no such private helper in another torch: the profile pass is then as noisy as before
        pass


def kernel_family(name: str):
    """profiled launch name -> (family label, binding roof, issued-multiply share of the algorithmic FLOPs, peak, unit)"""
    bf = "bf16" in name
    mf = 2500.0 if bf else PEAK_FP32_TFLOPS
    if name.startswith("conv3x3_wino_wgrad"):
        return ("Winograd F(2x2,3x3) weight gradient + its split-K reductions", "mfma", WINO_ISSUED, mf, "TFLOP/s")
    if name.startswith("conv3x3_wino_kernel"):
        return ("Winograd F(2x2,3x3) forward / data gradient", "mfma", WINO_ISSUED, mf, "TFLOP/s")
    if name.startswith("conv3x3_bf16_dma"):
        return ("bf16 LDS-DMA convolution forward / data gradient", "mfma", 1.0, mf, "TFLOP/s")
    if name.startswith("conv_wgrad_bf16"):
        return ("bf16 weight gradient + split-K final", "mfma", 1.0, mf, "TFLOP/s")
    if name.startswith("conv3x3_bf16_narrow"):   # HBM-bound by a wide margin (64-96 B and 4.6-18 kFLOP per pixel)
        return ("bf16 narrow-layer convolution forward / data gradient (Cin, Cout <= 32 at full resolution)", "hbm", 0.0,
                PEAK_HBM_GBS, "GB/s")
    if name.startswith("conv_fwd_bf16"):
        return ("bf16 register-staged convolution (Cout <= 32, stride 2, 1x1, stem)", "mfma", 1.0, mf, "TFLOP/s")
    if name.startswith("conv3x3_f32_upc"):   # 16 of the 36 direct multiplies per source pixel are issued
        return ("fp32 sub-pixel convolution of the up-sampled narrow layer (forward: 4 combined taps per output parity; "
                "data gradient: one 4x4 stride-2 kernel)", "mfma", 16.0 / 36.0, mf, "TFLOP/s")
    if name.startswith("conv3x3_f32_narrow"):
        return ("fp32 narrow-layer convolution forward / data gradient (Cin, Cout <= 32 at full resolution)", "mfma", 1.0, mf,
                "TFLOP/s")
    if name.startswith("conv_wgrad"):
        return ("direct weight gradient + split-K reductions (Cout <= 32, stride 2, 1x1, stem)", "mfma", 1.0, mf, "TFLOP/s")
    if name.startswith("conv_fwd"):
        return ("direct convolution forward / data gradient (Cout <= 32, stride 2, 1x1, stem)", "mfma", 1.0, mf, "TFLOP/s")
    if name.startswith("bn_bwd_apply"):
        return ("bn_bwd_apply (BatchNorm + ReLU backward, elementwise)", "hbm", 0.0, PEAK_HBM_GBS, "GB/s")
    if name.startswith("bn_bwd_reduce"):
        return ("bn_bwd_reduce (BatchNorm backward sums not fused into a producer)", "hbm", 0.0, PEAK_HBM_GBS, "GB/s")
    if name.startswith("bn_act"):
        return ("bn_act (BatchNorm apply + ReLU + residual, materialised activations)", "hbm", 0.0, PEAK_HBM_GBS, "GB/s")
    if name.startswith("head_"):
        return ("segmentation head forward / backward", "hbm", 0.0, PEAK_HBM_GBS, "GB/s")
    return (name, "hbm", 0.0, PEAK_HBM_GBS, "GB/s")


def roofline_tables(prof, prof_steps, share_s):
    """per-launch records (name, algorithmic FLOPs, start, end, algorithmic bytes) of `prof_steps` serial steps ->
    (per-kernel-name aggregate, family table sorted by time, issued matrix-core FLOPs per step, direct-equivalent FLOPs
    per step).  Every `frac` is a fraction of the family's binding roof: issued multiplies / MFMA peak, or algorithmic
    bytes / HBM peak."""
    by_name, fam = {}, {}
    for name, flops, a, b, nbytes in prof:
        t = a.elapsed_time(b) * 1e-3
        r = by_name.setdefault(name, [0.0, 0.0, 0, 0.0])
        r[0] += t; r[1] += flops; r[2] += 1; r[3] += nbytes
        label, bound, issued, peak, unit = kernel_family(name)
        f = fam.setdefault(label, {"t": 0.0, "fl": 0.0, "n": 0, "nb": 0.0, "bound": bound, "issued": issued, "peak": peak, "unit": unit})
        f["t"] += t; f["fl"] += flops; f["n"] += 1; f["nb"] += nbytes
    table, issued_step, direct_step = [], 0.0, 0.0
    for label, f in sorted(fam.items(), key=lambda kv: -kv[1]["t"]):
        if f["t"] <= 0.0:
            continue
        row = {"family": label, "bound": f["bound"], "launches_per_step": round(f["n"] / prof_steps, 1),
               "ms_per_step": round(1e3 * f["t"] / prof_steps, 3), "avg_launch_us": round(1e6 * f["t"] / f["n"], 1),
               "share_of_step": round((f["t"] / prof_steps) / share_s, 4), "peak": f["peak"], "unit": f["unit"]}
        if f["bound"] == "mfma":
            direct = f["fl"] / f["t"] / 1e12
            row["achieved"] = round(direct * f["issued"], 2)
            if f["issued"] != 1.0:
                row["direct_equiv_TFLOPs"] = round(direct, 2)
            issued_step += f["fl"] * f["issued"] / prof_steps
            direct_step += f["fl"] / prof_steps
            row["hbm_GBps_algorithmic"] = round(f["nb"] / f["t"] / 1e9, 1)
        else:
            row["achieved"] = round(f["nb"] / f["t"] / 1e9, 1)
        row["frac"] = round(row["achieved"] / f["peak"], 4)
        table.append(row)
    return by_name, table, issued_step, direct_step


def train_leg(args, ctx, precision, B, headline):
    """one timed training configuration -> result dict (the bench.py JSON contract keys + roofline block)"""
    import torch
    import torch.distributed as dist
    from deadtrees_amd.data.synthetic import synth_batch
    from deadtrees_amd.network.unet import UNetHIP
    from deadtrees_amd.trainer import HipTrainer
    dev, world, rank, distributed = ctx["dev"], ctx["world"], ctx["rank"], ctx["distributed"]
    S = args.size
    model = UNetHIP(in_channels=3, classes=2)
    model.reset_parameters(seed=0)
    model.to(dev)
    # auto: graph replay only where the step is launch-bound (single-GPU bf16); fp32 and every multi-GPU run stay
    # eager by default so that the 1/2/4/8-GPU numbers come from one code path ("on" forces it, RCCL included)
    use_graph = args.graph == "on" or (args.graph == "auto" and precision == "bf16" and not distributed)
    tr = HipTrainer(model, lr=3e-4, clip=0.5, losses=("GDICE", "FOCAL"), distributed=distributed,
                    precision=precision, graph=use_graph)
    tr.broadcast_parameters(0)
    nparams = model.flat_params.numel()
    img, mask = synth_batch(B, S, S, 3, 2, seed=1234 + rank)
    img, mask = img.to(dev), mask.to(dev)   # inputs resident in HBM before the timed region

    def sync():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    if use_graph:            # set-up, not warm-up: two eager steps, then the capture (first graph call)
        for _ in range(3):
            tr.step(img, mask)
        sb = tr.static_batch()   # the resident batch lives in the buffers the captured step reads (no staging copy)
        if sb is not None:
            sb[0].copy_(img)
            sb[1].copy_(mask)
            img, mask = sb[0], sb[1]
    for _ in range(args.warmup):
        tr.step(img, mask)
    sync()
    prof = []
    # fp32 default: the weight gradients run on a second HIP stream (UNetEngine.overlap_wgrad), so kernel lifetimes
    # overlap and a per-kernel duration is no longer that kernel's own time.  Like the graph-replay case the per-kernel
    # events therefore come from two SERIAL eager steps after the timed region (same kernels, same shapes, one stream).
    overlapped = bool(getattr(model.engine, "overlap_wgrad", False)) and precision == "fp32"
    model.engine.profile = None   # the timed steps run without per-launch events; the roofline tables come from below
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        loss = tr.step(img, mask)
    e1.record()
    sync()
    wall = time.perf_counter() - t0
    model.engine.profile = None
    dt = torch.tensor([wall], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    wall_rank = wall
    wall = float(dt)
    serial_step_s = None
    # per-kernel events for the roofline block come from two EAGER, single-stream steps after the timed region (same
    # kernels, same shapes; the timed steps above were graph replays / had the weight gradients on a second stream)
    tr.use_graph = False
    model.engine.overlap_wgrad = False
    tr.step(img, mask)
    torch.cuda.synchronize()
    model.engine.profile = prof
    p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # ~900 events per step make the host the slower side of these two steps; an event pair then also times the host's gap
    # between record() and the launch (observed: the dominant kernel's average anywhere from 0.22 to 0.42 ms).  A device-side
    # spin ahead of the steps lets the host enqueue both of them while the GPU is held: the events then run back to back
    _hold_gpu(torch, 2 * 0.09)
    p0.record()
    for _ in range(2):
        tr.step(img, mask)
    p1.record()
    torch.cuda.synchronize()
    serial_step_s = float(p0.elapsed_time(p1)) * 1e-3 / 2
    model.engine.profile = None
    model.engine.overlap_wgrad = overlapped
    # PCIe-inclusive rate (never `value`): the batch arrives in pinned host memory every step (fp32 image + int64
    # mask, what the reference's loader hands to Lightning) on a copy stream, overlapped with the previous step
    pcie = None
    if rank == 0 and not distributed and headline:
        tr.use_graph = use_graph
        himg, hmask = img.cpu().pin_memory(), mask.cpu().pin_memory()
        bufs = [(torch.empty_like(img), torch.empty_like(mask)) for _ in range(2)]
        copy, main_s = torch.cuda.Stream(), torch.cuda.current_stream()
        ev_in = [torch.cuda.Event() for _ in range(2)]
        ev_free = [torch.cuda.Event() for _ in range(2)]
        n = 6
        for timed in (False, True):      # first pass untimed: freshly pinned pages copy slowly the first time
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(n if timed else 2):
                k = i & 1
                with torch.cuda.stream(copy):
                    if i >= 2:
                        copy.wait_event(ev_free[k])
                    bufs[k][0].copy_(himg, non_blocking=True)
                    bufs[k][1].copy_(hmask, non_blocking=True)
                    ev_in[k].record(copy)
                main_s.wait_event(ev_in[k])
                tr.step(bufs[k][0], bufs[k][1])
                ev_free[k].record(main_s)
            torch.cuda.synchronize()
            pcie = B * n / (time.perf_counter() - t1)
    tiles = B * world * args.steps
    value = tiles / wall
    ms_per_step = 1e3 * wall / args.steps

    # ---- roofline of the dominant conv kernel (rank 0's launches)
    prof_steps = 2
    share_s = serial_step_s            # the (serial) step the profiled launches belong to
    agg, kernels, issued_step, direct_step = roofline_tables(prof, prof_steps, share_s)
    roof = None
    conv_names = {k: v for k, v in agg.items() if kernel_family(k)[1] == "mfma" and "(+" not in k}   # single-kernel brackets
    conv_time = sum(r[0] for k, r in agg.items() if k.startswith(("conv_fwd", "conv3x3_wino_kernel", "conv3x3_bf16_dma")))
    if conv_names:
        name, (t, fl, n, nb) = max(conv_names.items(), key=lambda kv: kv[1][0])
        _, _, issued, kpeak, _ = kernel_family(name)
        direct = fl / t / 1e12
        ach = direct * issued           # multiplies the kernel issues on the matrix cores per second
        roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": kpeak,
                "unit": "TFLOP/s", "frac": round(ach / kpeak, 4), "traffic": None,
                "launches": n, "avg_launch_ms": round(1e3 * t / n, 4),
                "algorithmic_bytes_per_launch": round(nb / n), "algorithmic_flops_per_launch": round(fl / n),
                "hbm_GBps_algorithmic": round(nb / t / 1e9, 1),
                "share_of_step": round((t / prof_steps) / share_s, 4)}
        if issued != 1.0:
            roof["algorithm"] = ("Winograd F(2x2,3x3): 16 MFMA multiplies per 36 algorithmic ones; achieved / frac count the "
                                 "ISSUED multiplies (a fraction of the matrix peak), direct_equiv_TFLOPs the algorithmic ones")
            roof["direct_equiv_TFLOPs"] = round(direct, 2)
            roof["issued_flops_per_launch"] = round(fl * issued / n)
        if overlapped or use_graph:
            roof["measured_on"] = ("2 serial eager steps after the timed region (one stream, "
                                   f"{1e3 * serial_step_s:.2f} ms per step): the timed steps " +
                                   ("run the weight gradients on a second stream, which overlaps kernel lifetimes" if overlapped
                                    else "are HIP-graph replays (no per-launch events)") +
                                   "; profiles/ holds rocprofv3 --kernel-trace --stats of the same serial steps")
        # HBM bytes per launch from the PMC passes (profiles/traffic.json: separate rocprofv3 --pmc runs of this command,
        # per-access-shape corrections from scripts/ubench/pmc_calib — see profiles/README.md)
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                # rocprofv3 spells every template argument (conv3x3_wino_kernel<false, 0, false>); the engine's name for
                # the launch leaves defaulted ones out: exact key first, then the key this name is a prefix of
                ent = tj.get(name) or next((v for k, v in sorted(tj.items()) if k.startswith(name[:-1] + ",") and isinstance(v, dict)), {})
                roof["traffic"] = ent.get("hbm_bytes_per_launch")
                if ent:
                    roof["traffic_detail"] = {k: ent[k] for k in ("fetch_bytes_per_launch", "write_bytes_per_launch", "read_shape",
                                                                   "read_factor", "write_shape", "write_factor") if k in ent}
            except Exception:
                pass
        roof["kernels"] = kernels
    per_gpu_tiles_s = B * args.steps / wall * (S / 512.0) ** 2   # 512x512-equivalent tiles for the FLOP/byte model
    step_wall_s = wall / args.steps
    peak_tf = PEAK_FP32_TFLOPS if precision == "fp32" else 2500.0   # dense bf16 MFMA peak
    bytes_per_tile = BYTES_PER_TILE_TRAIN if precision == "fp32" else 462.2e6
    out = {
        "metric": f"{S}x{S} RGB tiles/sec (train fwd+bwd)", "value": round(value, 2), "unit": "tiles/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if precision == "fp32" else "bf16 (fp32 accumulate, fp32 master weights)", "data": "synthetic",
        "config": {"workload": ("configs[1]" if precision == "fp32" else "configs[2]") +
                               f": reference U-Net (smp Unet/resnet34 topology) {precision} train step, "
                               f"batch {B}/GPU, {S}x{S}x3 tiles, GDICE+FOCAL, clip 0.5, Adam 3e-4",
                   "global_batch": B * world, "tile": S, "parallelism": f"dp{world}"},
        "loss": round(float(loss), 6), "hip_graph": use_graph,
        "pcie_inclusive_tiles_per_s": None if pcie is None else round(pcie, 2),
        # whole step: the multiplies all convolution kernels ISSUE per step (profiled launches: Winograd layers count 16/36
        # of their algorithmic FLOPs) / the timed step = the fraction of the matrix peak the step really uses; the
        # direct-convolution-equivalent rate (SURVEY 8d's 186.53 GFLOP per tile) beside it is NOT a roofline fraction
        "whole_net": {"mfma_issued_TFLOPs": round(issued_step / step_wall_s / 1e12, 2),
                      "mfma_issued_frac": round(issued_step / step_wall_s / 1e12 / peak_tf, 4),
                      "direct_equiv_TFLOPs": round(per_gpu_tiles_s * FLOP_PER_TILE_TRAIN / 1e12, 2),
                      "direct_equiv_over_mfma_peak": round(per_gpu_tiles_s * FLOP_PER_TILE_TRAIN / 1e12 / peak_tf, 4),
                      "profiled_conv_flops_per_step_over_model": round(direct_step / (B * (S / 512.0) ** 2 * FLOP_PER_TILE_TRAIN), 4),
                      "hbm_frac_step": round(per_gpu_tiles_s * bytes_per_tile / 1e9 / PEAK_HBM_GBS, 4),
                      "conv_fwd_dgrad_share_of_step": round((conv_time / prof_steps) / share_s, 4)},
        "roofline": roof,
    }
    if distributed:
        # N > 1 evidence: the collective really spans `world` ranks (an all-reduce of ones must return N), the spread of
        # the per-rank step times, and the gradient bytes each rank hands to the bucketed all-reduce per step
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        lo = torch.tensor([wall_rank], dtype=torch.float64, device=dev)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        out["rccl_ranks"] = int(round(float(ones)))
        out["dist_backend"] = dist.get_backend()
        out["ms_per_step_rank_min"] = round(1e3 * float(lo) / args.steps, 3)
        out["ms_per_step_rank_max"] = round(1e3 * float(hi) / args.steps, 3)
        out["allreduce_bytes_per_step"] = int(4 * nparams) if world > 1 else 0
        out["allreduce_buckets"] = 5
    # free this leg's buffers (saved activations, graph pool) before the next leg allocates its own
    del tr, model, img, mask
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def infer_bench(args, model, dev, world, rank, distributed, as_leg=False):
    """forward-only leg of the tiled inference path: uint8 RGBN sub-tiles resident in HBM -> normalise (fused
    kernel) -> U-Net forward (eval BN) -> uint8 class map from the head kernel.  One step = one batch."""
    import torch
    import torch.distributed as dist
    from deadtrees_amd import ops
    from deadtrees_amd.data.synthetic import MEAN, STD, synth_u8_batch
    B, S = args.batch, args.size
    u8 = synth_u8_batch(B, S, S, seed=99 + rank).to(dev)
    model.eval()

    use_graph = args.graph == "on"      # measured: the eager inference leg is already GPU-bound (6.9 k vs 6.7 k tiles/s)
    if use_graph:
        from deadtrees_amd.deployment.inference import GraphedTilePredictor
        graphed = GraphedTilePredictor(model, 3, args.precision)
        graphed(u8)                                  # set-up (eager warm-up + capture), not a timed or warm-up step

    def step(src=None):
        src = u8 if src is None else src
        if use_graph:
            return graphed(src)
        x = ops.normalize_u8(src, MEAN, STD, 3)          # NHWC fp32: the kernels' layout, no NCHW round trip
        return model.predict_classes(x, dtype="uint8", precision=args.precision, nhwc=True)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    wall = time.perf_counter() - t0
    dt = torch.tensor([wall], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    wall = float(dt)
    # roofline of the forward pass: per-launch events of two more (eager, single-stream) batches
    roof = None
    if not use_graph and rank == 0:
        prof = []
        step()
        torch.cuda.synchronize()
        model.engine.profile = prof
        q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _hold_gpu(torch, 0.04)     # see train_leg: the host enqueues both batches while the device spins
        q0.record()
        for _ in range(2):
            step()
        q1.record()
        torch.cuda.synchronize()
        model.engine.profile = None
        ser_s = float(q0.elapsed_time(q1)) * 1e-3 / 2
        agg, kernels, issued_step, direct_step = roofline_tables(prof, 2, ser_s)
        convs = {k: v for k, v in agg.items() if kernel_family(k)[1] == "mfma"}
        if convs:
            name, (t, fl, n, nb) = max(convs.items(), key=lambda kv: kv[1][0])
            _, _, issued, kpeak, _ = kernel_family(name)
            direct = fl / t / 1e12
            roof = {"bound": "mfma", "kernel": name, "achieved": round(direct * issued, 2), "peak": kpeak, "unit": "TFLOP/s",
                    "frac": round(direct * issued / kpeak, 4), "traffic": None, "launches": n,
                    "avg_launch_ms": round(1e3 * t / n, 4), "share_of_step": round((t / 2) / ser_s, 4),
                    "algorithmic_flops_per_launch": round(fl / n), "algorithmic_bytes_per_launch": round(nb / n),
                    "measured_on": f"2 eager batches with per-launch events after the timed region ({1e3 * ser_s:.2f} ms per batch)",
                    "whole_pass_mfma_issued_frac": round(issued_step / (wall / args.steps) / 1e12 / PEAK_FP32_TFLOPS, 4)
                    if args.precision == "fp32" else None,
                    "kernels": kernels}
            if issued != 1.0:
                roof["direct_equiv_TFLOPs"] = round(direct, 2)
            # HBM bytes per launch from the PMC passes of this command (profiles/traffic_infer.json, scripts/profile_round.sh)
            tfile = os.path.join(ROOT, "profiles", "traffic_infer.json")
            if os.path.exists(tfile) and (B, S) == (64, 256):
                try:
                    tj = json.load(open(tfile))
                    ent = tj.get(name) or next((v for k, v in sorted(tj.items())
                                                if k.startswith(name[:-1] + ",") and isinstance(v, dict)), {})
                    roof["traffic"] = ent.get("hbm_bytes_per_launch")
                    if ent:
                        roof["traffic_detail"] = {k: ent[k] for k in ("fetch_bytes_per_launch", "write_bytes_per_launch",
                                                                       "read_shape", "write_shape") if k in ent}
                except Exception:
                    pass
    # PCIe-inclusive rate (never `value`): uint8 tiles from pinned host memory, uint8 class maps back, double-buffered
    # on a copy stream so transfers overlap the previous batch's kernels
    pcie = None
    if rank == 0:
        host_in = [u8.cpu().pin_memory() for _ in range(2)]
        host_out = [torch.empty((B, S, S), dtype=torch.uint8).pin_memory() for _ in range(2)]
        dev_in = [torch.empty_like(u8) for _ in range(2)]
        copy, main_s = torch.cuda.Stream(), torch.cuda.current_stream()
        ev_in = [torch.cuda.Event() for _ in range(2)]
        ev_free = [torch.cuda.Event() for _ in range(2)]
        n = max(6, args.steps)
        for timed in (False, True):      # first pass untimed: freshly pinned pages copy slowly the first time
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(n if timed else 4):
                k = i & 1
                with torch.cuda.stream(copy):
                    if i >= 2:
                        copy.wait_event(ev_free[k])            # batch i-2 has consumed this buffer
                    dev_in[k].copy_(host_in[k], non_blocking=True)
                    ev_in[k].record(copy)
                main_s.wait_event(ev_in[k])
                o = step(dev_in[k])
                ev_free[k].record(main_s)
                host_out[k].copy_(o, non_blocking=True)
            torch.cuda.synchronize()
            pcie = B * n / (time.perf_counter() - t1)
    # BASELINE configs[4] end to end (scripts/inference.py:80-115 per ortho tile): a synthetic (4, 2048, 2048) uint8
    # raster -> block split (64 sub-tiles of 256x256 or 16 of 512x512) -> uint8 H2D -> normalise + forward + argmax on the
    # device -> uint8 D2H -> block merge.  Host work (split / merge, pageable copies) included; rank r takes ortho
    # tiles r, r + N, ... of the queue (no collective)
    tiler = None
    if S in (256, 512):
        import numpy as np
        from deadtrees_amd.deployment.tiler import infer_tile

        class _U8:
            in_channels = 3

            def run_u8(self, tiles_u8, device=None):
                x = ops.normalize_u8(tiles_u8.to(dev, non_blocking=True), MEAN, STD, 3)
                return model.predict_classes(x, dtype="uint8", precision=args.precision, nhwc=True)

            def run_blocks(self, raster, d, first, count):      # PyTorchInference.run_blocks on the bench's model
                x = ops.split_normalize_u8(raster, d, first, count, MEAN, STD, 3)
                return model.predict_classes(x, dtype="uint8", precision=args.precision, nhwc=True)

        ortho = np.random.default_rng(7 + rank).integers(0, 256, (4, 2048, 2048), dtype=np.uint8)
        for _ in range(3):                                                             # warm-up (clocks: the raster above
            infer_tile(_U8(), ortho, subtile=S, batch_size=64, device=str(dev))          # took ~50 ms of host time to draw)
        n_ortho = 8
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        t2 = time.perf_counter()
        for _ in range(n_ortho):
            merged = infer_tile(_U8(), ortho, subtile=S, batch_size=64, device=str(dev))
        torch.cuda.synchronize()
        dt2 = torch.tensor([time.perf_counter() - t2], dtype=torch.float64, device=dev)
        if distributed:
            dist.all_reduce(dt2, op=dist.ReduceOp.MAX)
        per_ortho = float(dt2) / n_ortho
        km2_ortho = (2048 * 0.20002 / 1000.0) ** 2       # 0.1678 km^2 per 2048^2 source tile
        tiler = {"ms_per_2048_ortho_tile": round(1e3 * per_ortho, 2),
                 "subtiles_per_s": round(world * (2048 // S) ** 2 / per_ortho, 1),
                 "km2_per_hour": round(world * km2_ortho / per_ortho * 3600.0, 1),
                 "what": "uint8 raster H2D + block split + normalise/forward/argmax + block merge on the device + uint8 map D2H, pageable host arrays",
                 "foreground_fraction": round(float(merged.mean()), 4)}
    tiles_s = B * world * args.steps / wall
    km2_per_tile = (S * 0.20002 / 1000.0) ** 2          # pixel 0.20002 m (scripts/computestats_inference.py:57-59)
    fwd_flop = 62.59e9 * (S / 512.0) ** 2
    res = {"metric": f"{S}x{S} RGB tiles/sec (inference fwd + argmax)", "value": round(tiles_s, 1), "unit": "tiles/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * wall / args.steps, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32" if args.precision == "fp32" else "bf16 (fp32 accumulate)", "data": "synthetic",
           "config": {"workload": f"tiled inference leg: uint8 {S}x{S}x4 sub-tiles, batch {B}/GPU, fused normalise + "
                                  "forward + uint8 argmax", "global_batch": B * world, "parallelism": f"dp{world}"},
           "km2_per_hour": round(tiles_s * km2_per_tile * 3600.0, 1),
           "whole_net": {"direct_equiv_TFLOPs": round(tiles_s / world * fwd_flop / 1e12, 2),
                         "direct_equiv_over_mfma_peak": round(tiles_s / world * fwd_flop / 1e12 / PEAK_FP32_TFLOPS, 4)},
           "roofline": roof,
           "tiler_inclusive": tiler,
           "foreground_pixels": int(out.sum()), "hip_graph": use_graph,
           "pcie_inclusive_tiles_per_s_per_gpu": None if pcie is None else round(pcie, 1)}
    if as_leg:
        return res
    if rank == 0:
        print(json.dumps(res), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
