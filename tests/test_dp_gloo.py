"""world_size-2 CPU (gloo) tests of the data-parallel pieces: bucketed gradient all-reduce over ranges of
the flat buffer, and tile-queue sharding of the tiled inference driver."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deadtrees_amd.network.spec import build_spec
        from deadtrees_amd.trainer import GradReducer
        spec = build_spec(3, 2)
        n = spec.n_params
        g = torch.Generator().manual_seed(100 + rank)
        grads = torch.randn(n, generator=g)
        mine = grads.clone()
        red = GradReducer()
        red.attach(grads)
        for name, lo, hi in spec.buckets:        # the order backward produces them
            red.hook(name, lo, hi)
        red.wait()
        other = torch.randn(n, generator=torch.Generator().manual_seed(100 + (1 - rank)))
        ok = torch.allclose(grads, mine + other, rtol=0, atol=0)

        # tile-queue sharding: fake inference = threshold of channel 0
        from deadtrees_amd.deployment.tiler import infer_tile

        class Fake:
            def run_u8(self, u8, device="cpu"):
                return (u8[..., 0] > 127).to(torch.uint8)
        rng = np.random.default_rng(0)
        arr = rng.integers(0, 256, (4, 512, 512), dtype=np.uint8)
        out = infer_tile(Fake(), arr, subtile=128, batch_size=4, rank=rank, world=world, device="cpu",
                         tile_shape=(512, 512))
        ok2 = np.array_equal(out, (arr[0] > 127).astype(np.uint8))
        q.put((rank, bool(ok), bool(ok2)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_allreduce_and_tile_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), "bucketed all-reduce mismatch"
    assert all(r[2] for r in res), "sharded tiled inference mismatch"


def _gpu_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deadtrees_amd.data.synthetic import synth_batch
        from deadtrees_amd.network.unet import UNetHIP
        from deadtrees_amd.trainer import HipTrainer
        dev = "cuda:0"
        torch.cuda.set_device(0)
        img, mask = synth_batch(2, 64, 64, 3, 2, seed=50 + rank)
        img, mask = img.to(dev), mask.to(dev)
        # local (un-reduced) gradient of this rank's batch
        solo = UNetHIP()
        solo.reset_parameters(seed=3)
        solo.to(dev)
        st = HipTrainer(solo)
        st.step(img, mask)
        g_local = solo._grad_buffer().clone()
        p_solo = solo.flat_params.detach().clone()
        # data-parallel step
        m = UNetHIP()
        m.reset_parameters(seed=3 + rank)     # different init per rank: broadcast must fix it
        m.to(dev)
        tr = HipTrainer(m, distributed=True)
        tr.broadcast_parameters(0)
        tr.step(img, mask)
        g_sum = m._grad_buffer().clone()
        gathered = [torch.empty_like(g_local) for _ in range(world)]
        dist.all_gather(gathered, g_local)
        ok_sum = torch.equal(g_sum, gathered[0] + gathered[1])     # sum of the independent replicas' gradients
        ps = [torch.empty_like(m.flat_params.data) for _ in range(world)]
        dist.all_gather(ps, m.flat_params.data)
        ok_sync = torch.equal(ps[0], ps[1])                         # replicas stay identical after the update
        moved = not torch.equal(ps[0], p_solo)                      # and differ from the single-replica update
        q.put(("fp32", rank, bool(ok_sum), bool(ok_sync), bool(moved)))
        _bf16_dp_case(rank, world, q)     # the same check under bf16 in the same pair of processes (a spawn costs ~10 s)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_data_parallel_step_equals_sum_of_replica_gradients():
    """2 ranks sharing one GPU (gloo): after the bucketed all-reduce the flat gradient buffer is bit-equal to the
    sum of the two replicas' independent gradients (per-replica BatchNorm / GDICE) and the replicas
    hold identical parameters after the fused clip+Adam step with the 1/N folded into the clip coefficient."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(4)]
    for p in procs:
        p.join(60)
    f32 = [r for r in res if r[0] == "fp32"]
    b16 = [r for r in res if r[0] == "bf16"]
    assert len(f32) == 2 and len(b16) == 2
    assert all(r[2] for r in f32), "all-reduced gradient != sum of replica gradients"
    assert all(r[3] for r in f32), "replicas diverged"
    assert all(r[4] for r in f32)
    # HipTrainer(distributed=True, precision="bf16") (BASELINE configs[2] x configs[3])
    assert all(r[2] for r in b16), "bf16: all-reduced gradient != sum of replica gradients"
    assert all(r[3] for r in b16), "bf16: replicas diverged"


def _nan_worker(rank, world, port, q, precisions):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for precision in precisions:      # both precisions in ONE pair of processes (a spawn costs ~10 s of imports)
            _nan_case(rank, world, q, precision)
    finally:
        dist.destroy_process_group()


def _nan_case(rank, world, q, precision):
    from deadtrees_amd.data.synthetic import synth_batch
    from deadtrees_amd.network.unet import UNetHIP
    from deadtrees_amd.trainer import HipTrainer
    dev = "cuda:0"
    torch.cuda.set_device(0)
    img, mask = synth_batch(2, 64, 64, 3, 2, seed=70 + rank)
    img, mask = img.to(dev), mask.to(dev)
    m = UNetHIP()
    m.reset_parameters(seed=5)
    m.to(dev)
    tr = HipTrainer(m, distributed=True, precision=precision)
    tr.broadcast_parameters(0)
    tr.step(img, mask)                                   # a clean step first: Adam state exists, t = 1
    p1 = m.flat_params.detach().clone()
    m1, v1 = tr.opt.m.clone(), tr.opt.v.clone()
    bad = img.clone()
    if rank == 1:
        bad[0, 0, 0, 0] = float("nan")                   # ONE rank sees a non-finite loss
    tr.step(bad, mask)
    local_skip = int(tr.last["skipped"])                 # the flag after the MAX all-reduce: global
    same = torch.equal(m.flat_params.detach(), p1) and torch.equal(tr.opt.m, m1) and torch.equal(tr.opt.v, v1)
    steps = tr.opt.steps_applied()
    tr.step(img, mask)                                   # training goes on, replicas identical
    ps = [torch.empty_like(p1) for _ in range(world)]
    dist.all_gather(ps, m.flat_params.data)
    q.put((precision, rank, local_skip, bool(same), steps, bool(torch.equal(ps[0], ps[1])), tr.opt.steps_applied(),
           bool(torch.isfinite(m.flat_params).all())))


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_nonfinite_loss_on_one_rank_skips_the_step_on_all_ranks():
    """The gradient buckets are summed over the replicas BEFORE the skip decision, so the decision must be global:
    with a NaN batch on rank 1 only, both ranks skip (parameters and Adam moments bit-identical to before, the step
    count does not advance — Lightning does not call optimizer.step when training_step returns None), and the next
    clean step keeps the replicas identical and finite."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    precisions = ("fp32", "bf16")
    procs = [ctx.Process(target=_nan_worker, args=(r, 2, port, q, precisions)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(2 * len(precisions))]
    for p in procs:
        p.join(60)
    assert sorted((r[0], r[1]) for r in res) == [("bf16", 0), ("bf16", 1), ("fp32", 0), ("fp32", 1)]
    for precision, rank, skip, same, steps, synced, steps_after, finite in res:
        assert skip == 1, f"{precision} rank {rank} did not skip"
        assert same, f"{precision} rank {rank}: parameters / Adam state moved in the skipped step"
        assert steps == 1 and steps_after == 2, (precision, rank, steps, steps_after)
        assert synced and finite


def _bf16_dp_case(rank, world, q):
    from deadtrees_amd.data.synthetic import synth_batch
    from deadtrees_amd.network.unet import UNetHIP
    from deadtrees_amd.trainer import HipTrainer
    dev = "cuda:0"
    img, mask = synth_batch(2, 64, 64, 3, 2, seed=90 + rank)
    img, mask = img.to(dev), mask.to(dev)
    solo = UNetHIP()
    solo.reset_parameters(seed=3)
    solo.to(dev)
    HipTrainer(solo, precision="bf16").step(img, mask)
    g_local = solo._grad_buffer().clone()
    m = UNetHIP()
    m.reset_parameters(seed=3)
    m.to(dev)
    tr = HipTrainer(m, distributed=True, precision="bf16")
    tr.broadcast_parameters(0)
    tr.step(img, mask)
    gathered = [torch.empty_like(g_local) for _ in range(world)]
    dist.all_gather(gathered, g_local)
    ok_sum = torch.equal(m._grad_buffer(), gathered[0] + gathered[1])
    ps = [torch.empty_like(m.flat_params.data) for _ in range(world)]
    dist.all_gather(ps, m.flat_params.data)
    q.put(("bf16", rank, bool(ok_sum), bool(torch.equal(ps[0], ps[1]))))


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_gpus_2_runs_as_typed():
    """`python bench.py --gpus 2 ...` with no launcher around it: the parent spawns the two ranks as
    a child torch.distributed.run before touching the GPU, relays rank 0's JSON line and the exit code.  DT_DIST_BACKEND
    =gloo lets both ranks share this box's one GPU; on an 8-GPU node the same command runs RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DT_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--batch", "2", "--size", "64", "--no-cpu-baseline", "--no-legs"],
                       env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["config"]["global_batch"] == 4
    assert out["scaling"] == "weak" and out["value"] > 0 and out["allreduce_bytes_per_step"] == 4 * 24436516
    assert out["ms_per_step_rank_min"] <= out["ms_per_step_rank_max"]
    # (a 64-pixel tile on two ranks sharing the GPU is launch-latency bound: the fraction can round to 0.0000)
    assert 0 <= out["roofline"]["frac"] <= 1.0 and out["roofline"]["kernels"] and out["roofline"]["launches"] > 0
