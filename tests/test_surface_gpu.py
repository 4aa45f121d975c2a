"""SemSegment / trainer / inference surface on the HIP path, against the oracle where numbers are involved."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cfg(**kw):
    from deadtrees_amd.utils.config import default_network, default_training
    return default_network(**kw), default_training()


@pytest.mark.parametrize("winograd", [False, True])
def test_trainer_loss_trajectory_matches_oracle(winograd):
    """4 optimiser steps (fwd, GDICE+FOCAL, bwd, clip 0.5, Adam 3e-4) on the same batch: loss curve vs oracle.
    Adam's first steps move every weight by +-lr, so a gradient within rounding of zero can go either way: the curves
    separate slowly.  Direct kernels (exact fp32 fma chains): 2e-3 after the first step; the default engine (3x3
    stride-1 layers as Winograd F(2x2,3x3), ~1e-6 relative per layer): 5e-3 (measured 2.2e-3 at step 3)."""
    from deadtrees_amd.data.synthetic import synth_batch
    from deadtrees_amd.network.unet import UNetHIP
    from deadtrees_amd.trainer import HipTrainer
    from oracle.train_ref import RefTrainer
    from oracle.unet_ref import make_oracle
    ref = make_oracle(3, 2, seed=0)
    m = UNetHIP()
    m.load_state_dict(ref.state_dict())
    m.to(DEV)
    m.engine.winograd = winograd
    later = 5e-3 if winograd else 2e-3
    img, mask = synth_batch(2, 128, 128, 3, 2, seed=7)
    rt = RefTrainer(ref)
    ht = HipTrainer(m)
    for step in range(4):
        lr_, gn_ = rt.step(img, mask)
        lh = float(ht.step(img.to(DEV), mask.to(DEV)))
        gn = float(ht.last["grad_norm"])
        assert lh == pytest.approx(lr_, rel=later if step else 2e-5), (step, lh, lr_)
        assert gn == pytest.approx(gn_, rel=5e-2), (step, gn, gn_)
        assert int(ht.last["skipped"]) == 0
    # parameters moved the same way (Adam's first steps are +-lr: compare the bulk)
    sd, sr = m.state_dict(), ref.state_dict()
    w, wr = sd["decoder.blocks.2.conv1.0.weight"].cpu(), sr["decoder.blocks.2.conv1.0.weight"]
    assert float((w - wr).abs().mean()) < 0.1 * 4 * 3e-4


def test_nonfinite_loss_skips_update():
    from deadtrees_amd.data.synthetic import synth_batch
    from deadtrees_amd.network.unet import UNetHIP
    from deadtrees_amd.trainer import HipTrainer
    m = UNetHIP().to(DEV)
    img, mask = synth_batch(2, 64, 64)
    ht = HipTrainer(m)
    before = m.flat_params.detach().clone()
    bad = img.clone()
    bad[0, 0, 0, 0] = float("nan")
    ht.step(bad.to(DEV), mask.to(DEV))
    assert int(ht.last["skipped"]) == 1
    assert torch.equal(m.flat_params.detach(), before)       # reference: training_step returns None -> no step


@pytest.mark.parametrize("decoder", ["resunet", "unetplusplus", "unet"])
def test_nan_in_a_raw_conv_output_skips_the_update(decoder, monkeypatch):
    """a NaN that enters a raw convolution output of a block whose BatchNorm + ReLU is applied while
    the next Winograd kernel stages it (resunet / unet++ conv2, unet with DT_MATERIALIZE_Z1=0) used to be zeroed by the
    v_max ReLU: the loss stayed finite, the step was NOT skipped and NaN gradients reached Adam.  The NaN is injected into
    ONE weight of a decoder conv1 (so the encoder and the input are clean): the step must be skipped, the parameters must
    stay bit-identical, and the next clean step must train."""
    from deadtrees_amd.data.synthetic import synth_batch
    from deadtrees_amd.network.unet import UNetHIP
    from deadtrees_amd.trainer import HipTrainer
    if decoder == "unet":
        monkeypatch.setenv("DT_MATERIALIZE_Z1", "0")
        monkeypatch.setenv("DT_MATERIALIZE_Z2", "0")
    m = UNetHIP(decoder=decoder).to(DEV)
    img, mask = synth_batch(2, 64, 64)
    ht = HipTrainer(m)
    ht.step(img.to(DEV), mask.to(DEV))
    assert int(ht.last["skipped"]) == 0
    conv1 = m.spec.decoder[1].conv1
    clean = m.flat_params.detach().clone()
    with torch.no_grad():
        m.flat_params[conv1.w_off + 5] = float("nan")
    m.engine.mark_weights_changed()
    poisoned = m.flat_params.detach().clone()
    ht.step(img.to(DEV), mask.to(DEV))
    assert int(ht.last["skipped"]) == 1
    same = (m.flat_params.detach() == poisoned) | (torch.isnan(m.flat_params.detach()) & torch.isnan(poisoned))
    assert bool(same.all())                                   # no update, and no NaN spread through the flat buffer
    assert int(torch.isnan(m.flat_params.detach()).sum()) == 1
    with torch.no_grad():
        m.flat_params.copy_(clean)
    m.engine.mark_weights_changed()
    ht.step(img.to(DEV), mask.to(DEV))
    assert int(ht.last["skipped"]) == 0 and not torch.equal(m.flat_params.detach(), clean)


def test_semsegment_steps():
    from deadtrees_amd.data.deadtreedata import DeadtreesDataModule
    from deadtrees_amd.network.segmodel import SemSegment
    net, tr = _cfg(losses=["DICE", "FOCAL", "BOUNDARY-RAMPED"])
    model = SemSegment(net, tr).to(DEV)
    dm = DeadtreesDataModule(train_dataloader_conf={"batch_size": 2}, val_dataloader_conf={"batch_size": 2},
                             test_dataloader_conf={"batch_size": 2}, synthetic_batches=1, tile_size=64, device=DEV)
    dm.setup(in_channels=3, classes=2)
    model.train()
    loss = model.training_step(next(iter(dm.train_dataloader())), 0)
    assert loss.dim() == 0 and torch.isfinite(loss)
    loss.backward()
    assert model.model.flat_params.grad is not None and float(model.model.flat_params.grad.abs().sum()) > 0
    means = model.epoch_means()
    for k in ("train/dice_loss", "train/boundary_loss", "train/focal_loss", "train/total_loss", "train/dice",
              "train/dice_with_bg"):
        assert k in means, k
    assert len(model.stats["train"]) == 2
    model.eval()
    with torch.no_grad():
        out = model.validation_step(next(iter(dm.val_dataloader())), 0)
        assert set(out) == {"val_loss", "target", "prediction", "lu"}
        assert out["prediction"].shape == (2, 64, 64) and out["prediction"].dtype == torch.int64
        out = model.test_step(next(iter(dm.test_dataloader())), 0)
        assert set(out) == {"target", "prediction", "lu"}
    assert int(model.label_error) == 0
    cms = model.test_epoch_end()
    assert set(cms) == {"cm_px", "cm_norm", "cm_px_masked", "cm_norm_masked"}
    assert int(cms["cm_px"].sum()) == 2 * 64 * 64
    assert set(model.validation_epoch_end()) == set(cms)


def test_confusion_matrix_kernel():
    from deadtrees_amd import ops
    g = torch.Generator().manual_seed(2)
    K = 3
    pred = torch.randint(0, K, (4, 96, 96), generator=g)
    tgt = torch.randint(0, K, (4, 96, 96), generator=g)
    lu = torch.randint(0, 3, (4, 96, 96), generator=g)
    counts, err = ops.confusion_matrix(pred.to(DEV), tgt.to(DEV), lu.to(DEV), K=K)
    counts, err = ops.confusion_matrix(pred.to(torch.uint8).to(DEV), tgt.to(DEV), lu.to(DEV), K=K, counts=counts)
    want = torch.zeros((K, K), dtype=torch.int64)
    wantm = torch.zeros((K, K), dtype=torch.int64)
    idx = (tgt * K + pred).flatten()
    want += torch.bincount(idx, minlength=K * K).view(K, K)
    wantm += torch.bincount(idx[lu.flatten() == 1], minlength=K * K).view(K, K)
    assert int(err) == 0
    assert torch.equal(counts[0].cpu(), 2 * want) and torch.equal(counts[1].cpu(), 2 * wantm)


def test_inference_checkpoint_fused_argmax_and_tiles(tmp_path):
    from deadtrees_amd.data.deadtreedata import val_transform
    from deadtrees_amd.deployment.inference import PyTorchInference
    from deadtrees_amd.deployment.tiler import infer_tile, make_blocks_vectorized
    from deadtrees_amd.network.segmodel import SemSegment
    from oracle.unet_ref import make_oracle
    net, tr = _cfg()
    model = SemSegment(net, tr)
    ref = make_oracle(3, 2, seed=1)
    model.model.load_state_dict(ref.state_dict())
    p = tmp_path / "model.ckpt"
    model.save_checkpoint(p)
    inf = PyTorchInference(p)
    rng = np.random.default_rng(1)
    arr = rng.integers(0, 256, (4, 512, 512), dtype=np.uint8)            # RGBN "ortho" tile
    # reference-style path: per-subtile val_transform on the host, f32 batch, run() (RGB slice inside)
    subs = make_blocks_vectorized(arr, 256)
    batch = torch.stack([val_transform(image=s.transpose(1, 2, 0))["image"] for s in subs])
    am = inf.run(batch, device=DEV).cpu()
    assert am.dtype == torch.int64 and tuple(am.shape) == (4, 256, 256)
    ref.eval()
    with torch.no_grad():
        want = ref(batch[:, :3].double() if False else batch[:, :3]).argmax(dim=1)
    assert float((am != want).float().mean()) < 1e-3                     # near-tie pixels only
    single = inf.run(batch[0], device=DEV)
    assert tuple(single.shape) == (256, 256)                             # tests/test_inference.py:87-93 shape contract
    assert torch.equal(single.cpu(), am[0])
    # MI355X path: uint8 in, uint8 out, normalisation on the device; block merge bit-identical
    merged = infer_tile(inf, arr, subtile=256, batch_size=2, device=DEV, tile_shape=(512, 512))
    assert merged.dtype == np.uint8 and merged.shape == (512, 512)
    from deadtrees_amd.deployment.tiler import unmake_blocks_vectorized
    want_merged = unmake_blocks_vectorized([am.numpy().astype(np.uint8)], 256, 512, 512)
    assert float((merged != want_merged).mean()) < 1e-4                  # fp32 normalise on device vs host


def test_fit_loop_with_cosine_schedule_and_boundary_loss():
    from deadtrees_amd.data.deadtreedata import DeadtreesDataModule
    from deadtrees_amd.data.distmap import distmaps_for_batch
    from deadtrees_amd.network.unet import UNetHIP
    from deadtrees_amd.trainer import HipTrainer, fit

    class WithDist:
        def __init__(self, loader):
            self.loader = loader

        def __iter__(self):
            for b in self.loader:
                img, mask, _, lu, stats = b["main"]
                yield {"main": (img, mask, distmaps_for_batch(mask, 2), lu, stats)}

    dm = DeadtreesDataModule(train_dataloader_conf={"batch_size": 2}, synthetic_batches=2, tile_size=64)
    dm.setup(in_channels=3, classes=2)
    m = UNetHIP().to(DEV)
    tr = HipTrainer(m, losses=("GDICE", "BOUNDARY-RAMPED", "FOCAL"))
    hist = fit(tr, WithDist(dm.train_dataloader()), epochs=3, base_lr=3e-4, t_max=10, to_device=DEV)
    assert [round(h["lr"] / 3e-4, 4) for h in hist] == [1.0, 0.9755, 0.9045]
    assert all(np.isfinite(h["train/total_loss"]) for h in hist)
    assert hist[-1]["train/total_loss"] < hist[0]["train/total_loss"]      # it learns on the repeated batches


def test_ensemble_vote_matches_torch_mode(tmp_path):
    """dt_ensemble_vote against torch.mode on the CPU (the call of deployment/inference.py:116), ties included; then
    the PyTorchEnsembleInference mirror with three checkpoints against stack + mode of the single-model maps."""
    from deadtrees_amd import ops
    from deadtrees_amd.deployment.inference import PyTorchEnsembleInference, PyTorchInference
    from deadtrees_amd.network.segmodel import SemSegment
    from oracle.unet_ref import make_oracle
    g = torch.Generator().manual_seed(0)
    for M, K in ((3, 2), (5, 3), (7, 3), (1, 2)):
        maps = torch.randint(0, K, (M, 2, 36, 52), generator=g, dtype=torch.uint8)
        want = torch.mode(maps.long(), dim=0)[0]
        got, err = ops.ensemble_vote(maps.to(DEV), K, dtype="int64")
        assert int(err) == 0 and torch.equal(got.cpu(), want)
        got8, _ = ops.ensemble_vote(maps.to(DEV), K, dtype="uint8")
        assert torch.equal(got8.cpu().long(), want)
    bad = torch.zeros((3, 1, 8, 8), dtype=torch.uint8)
    bad[1, 0, 2, 2] = 5
    assert int(ops.ensemble_vote(bad.to(DEV), 3)[1]) == 1
    net, tr = _cfg()
    files = []
    for seed in (1, 2, 3):
        model = SemSegment(net, tr)
        model.model.load_state_dict(make_oracle(3, 2, seed=seed).state_dict())
        files.append(tmp_path / f"m{seed}.ckpt")
        model.save_checkpoint(files[-1])
    with pytest.raises(ValueError):
        PyTorchEnsembleInference(files[0], files[1])
    x = torch.randn((2, 4, 128, 128), generator=g)
    ensemble = PyTorchEnsembleInference(*files)
    ens = ensemble.run(x, device=DEV).cpu()
    # the members' own maps: one checkpoint through PyTorchInference (the loader path), the other two straight from the
    # ensemble's loaded models (each load + model build costs seconds of host time)
    xs = x[:, :3].contiguous().to(DEV)
    first = PyTorchInference(files[0]).run(x, device=DEV).cpu()
    assert torch.equal(first, ensemble._models[0].predict_classes(xs).cpu())
    singles = torch.stack([first] + [m.predict_classes(xs).cpu() for m in ensemble._models[1:]], dim=1)
    assert torch.equal(ens, torch.mode(singles, dim=1)[0])          # the reference's stack(dim=1) + mode(axis=1)


def test_device_train_transform_matches_numpy_restatement():
    """dt_augment_normalize_u8 / dt_augment_labels against oracle/augment_ref.py for every flip x rot90 combination
    (exact: index maps) and brightness/contrast draws (exact LUT incl. clipping; albumentations itself is absent:
    parity unpinned for that step, see the oracle's header)."""
    from deadtrees_amd import ops
    from deadtrees_amd.data.deadtreedata import draw_train_params, train_transform_device
    from deadtrees_amd.data.synthetic import MEAN, STD
    from oracle import augment_ref as A
    rng = np.random.default_rng(3)
    combos = [(f, r) for f in (0, 1, 2) for r in (0, 1, 2, 3)]
    B, S = len(combos), 64
    tiles = rng.integers(0, 256, (B, S, S, 4), dtype=np.uint8)
    tiles[1] = np.clip(tiles[1].astype(np.int32) + 150, 0, 255)          # drive the LUT into the upper clip
    mask = rng.integers(0, 3, (B, S, S)).astype(np.int64)
    geo = torch.tensor(combos, dtype=torch.int32)
    bc = torch.tensor([[1.0, 0.0] if i % 3 == 0 else [1 + rng.uniform(-.15, .15), rng.uniform(-.2, .2)]
                       for i in range(B)], dtype=torch.float32)
    got = ops.augment_normalize_u8(torch.from_numpy(tiles).to(DEV), geo.to(DEV), bc.to(DEV), MEAN, STD, 3).cpu().numpy()
    gm = ops.augment_labels(torch.from_numpy(mask).to(DEV), geo.to(DEV)).cpu().numpy()
    for b, (f, r) in enumerate(combos):
        want = A.train_transform(tiles[b], f, r, float(bc[b, 0]), float(bc[b, 1]), MEAN, STD, 3)
        if float(bc[b, 0]) == 1.0 and float(bc[b, 1]) == 0.0:
            np.testing.assert_array_equal(got[b], want)
        else:   # LUT entries that land within float rounding of an integer may truncate differently: <= 1 grey level
            diff = np.abs(got[b] - want) * (np.asarray(STD[:3], np.float32) * 255.0)
            assert float(diff.max()) <= 1.0 + 1e-3 and float((diff > 1e-3).mean()) < 1e-2
        np.testing.assert_array_equal(gm[b], A.geometric(mask[b], f, r))
    # non-square tiles: flips and 180-degree turns only
    t2 = rng.integers(0, 256, (2, 32, 48, 4), dtype=np.uint8)
    g2 = torch.tensor([[1, 2], [2, 0]], dtype=torch.int32)
    b2 = torch.tensor([[1.0, 0.0], [1.0, 0.0]])
    got2 = ops.augment_normalize_u8(torch.from_numpy(t2).to(DEV), g2.to(DEV), b2.to(DEV), MEAN, STD, 4).cpu().numpy()
    for b in range(2):
        np.testing.assert_array_equal(got2[b], A.train_transform(t2[b], int(g2[b, 0]), int(g2[b, 1]), 1.0, 0.0, MEAN, STD, 4))
    with pytest.raises(RuntimeError):
        ops.augment_normalize_u8(torch.from_numpy(t2).to(DEV), torch.tensor([[0, 1], [0, 0]], dtype=torch.int32).to(DEV),
                                 b2.to(DEV), MEAN, STD, 4)
    # batch-level wrapper: shapes / dtypes of the reference's transform() output, distribution of the draws
    img, m, lu = train_transform_device(torch.from_numpy(tiles).to(DEV), torch.from_numpy(mask).to(DEV), None,
                                        np.random.default_rng(0), in_channels=3)
    assert tuple(img.shape) == (B, 3, S, S) and img.dtype == torch.float32 and m.dtype == torch.int64 and lu is None
    geo_d, bc_d = draw_train_params(4000, np.random.default_rng(1))
    assert abs(float((geo_d[:, 0] == 0).float().mean()) - 0.5) < 0.03
    assert abs(float((geo_d[:, 0] == 1).float().mean()) - 0.25) < 0.03
    assert abs(float((geo_d[:, 1] == 0).float().mean()) - 0.625) < 0.03      # p=0.5 none + 1/4 of the applied draws
    assert abs(float((bc_d[:, 0] != 1).float().mean()) - 0.5) < 0.03


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_graph_replayed_training_equals_eager(precision):
    """HipTrainer(graph=True): two eager steps, capture, replays.  The captured step launches the same kernels in
    the same order, so after N steps parameters, Adam state and losses equal the eager trainer's bit for bit
    (Adam's step count / learning rate are read from the device inside the graph)."""
    from deadtrees_amd.data.synthetic import synth_batch
    from deadtrees_amd.network.unet import UNetHIP
    from deadtrees_amd.trainer import HipTrainer
    from oracle.unet_ref import make_oracle
    ref = make_oracle(3, 2, seed=0)
    batches = [tuple(t.to(DEV) for t in synth_batch(2, 128, 128, 3, 2, seed=s)) for s in range(3)]
    out = {}
    for mode in ("eager", "graph"):
        m = UNetHIP()
        m.load_state_dict(ref.state_dict())
        m.to(DEV)
        tr = HipTrainer(m, precision=precision, graph=(mode == "graph"))
        losses = []
        for step in range(7):
            if step == 5:
                tr.opt.lr = 1e-4                      # a learning-rate change must reach the replayed step
            img, mask = batches[step % 3]
            sb = tr.static_batch() if mode == "graph" and step >= 4 else None
            if sb is not None:      # a loader writing into the graph's own buffers: step() must not copy, same results
                sb[0].copy_(img)
                sb[1].copy_(mask)
                img, mask = sb[0], sb[1]
            losses.append(float(tr.step(img, mask)))
        out[mode] = (losses, m.flat_params.detach().clone(), tr.opt.m.clone(), tr.opt.v.clone(), m.bn_state.clone(),
                     tr.opt.t)
        if mode == "graph":
            assert tr._graph is not None and "graph" in tr._graph
    le, pe, me, ve, be, te = out["eager"]
    lg, pg, mg, vg, bg, tg = out["graph"]
    assert te == tg == 7
    assert le == lg, (le, lg)
    assert torch.equal(pe, pg) and torch.equal(me, mg) and torch.equal(ve, vg) and torch.equal(be, bg)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_graphed_tile_predictor_equals_eager(precision):
    """GraphedTilePredictor: uint8 tiles -> uint8 class maps through one HIP graph; bit-identical to the eager chain,
    for changing inputs and two batch shapes."""
    from deadtrees_amd import ops
    from deadtrees_amd.data.synthetic import MEAN, STD, synth_u8_batch
    from deadtrees_amd.deployment.inference import GraphedTilePredictor
    from deadtrees_amd.network.unet import UNetHIP
    from oracle.unet_ref import make_oracle
    m = UNetHIP()
    m.load_state_dict(make_oracle(3, 2, seed=2).state_dict())
    m.to(DEV).eval()
    gp = GraphedTilePredictor(m, 3, precision)
    for seed, (B, S) in enumerate([(2, 128), (2, 128), (3, 64), (2, 128)]):
        u8 = synth_u8_batch(B, S, S, seed=seed).to(DEV)
        x = ops.normalize_u8(u8, MEAN, STD, 3).permute(0, 3, 1, 2).contiguous()
        want = m.predict_classes(x, dtype="uint8", precision=precision)
        got = gp(u8).clone()
        assert torch.equal(got, want)
    assert len(gp._graphs) == 2


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_cached_eval_affine_is_invalidated_by_training_and_by_state_loads(precision):
    """repeated inference calls skip the eval-mode BatchNorm scale/shift launches; a training step (running statistics
    and weights rewritten on the device), a load_state_dict and an in-place torch write must each invalidate the cache"""
    from deadtrees_amd.data.synthetic import synth_batch
    from deadtrees_amd.network.unet import UNetHIP
    from deadtrees_amd.trainer import HipTrainer
    m = UNetHIP()
    m.reset_parameters(seed=3)
    m = m.to(DEV)
    m.precision = precision
    img, mask = synth_batch(2, 64, 64, 3, 2, seed=4)
    img, mask = img.to(DEV), mask.to(DEV)

    def fresh_copy_logits():
        m2 = UNetHIP()
        m2.load_state_dict(m.state_dict())
        m2 = m2.to(DEV).eval()
        m2.precision = precision
        with torch.no_grad():
            return m2(img)

    m.eval()
    with torch.no_grad():
        a = m(img)
        b = m(img)                                   # served with cached coefficients
    assert torch.equal(a, b) and torch.equal(a, fresh_copy_logits())
    tr = HipTrainer(m, precision=precision)
    tr.step(img, mask)
    m.eval()
    with torch.no_grad():
        c = m(img)
    assert not torch.equal(a, c) and torch.equal(c, fresh_copy_logits())
    sd = {k: (v * 1.5 if k.endswith("running_var") else v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    with torch.no_grad():
        d = m(img)
    assert not torch.equal(c, d) and torch.equal(d, fresh_copy_logits())
    with torch.no_grad():
        m.flat_params.mul_(1.01)
        e = m(img)
    assert not torch.equal(d, e) and torch.equal(e, fresh_copy_logits())


@pytest.mark.parametrize("h,w,d", [(512, 512, 256), (300, 470, 128), (64, 64, 64)])
def test_split_normalize_gather_matches_tiler_blocks_and_val_transform(h, w, d):
    """dt_split_normalize_u8 = zero-pad (tiler.py:121-134) + make_blocks_vectorized (utils/data_handling.py:9-20) +
    albumentations Normalize + channel selection (scripts/inference.py:94-96, inference.py:57-59) in one gather:
    bit-identical to the host restatement of those steps, for full and ragged rasters, whole grids and sub-ranges."""
    from deadtrees_amd import ops
    from deadtrees_amd.data.deadtreedata import val_transform
    from deadtrees_amd.data.synthetic import MEAN, STD
    from deadtrees_amd.deployment.tiler import make_blocks_vectorized
    rng = np.random.default_rng(h + w)
    raster = rng.integers(0, 256, (4, h, w), dtype=np.uint8)
    nby, nbx = -(-h // d), -(-w // d)
    padded = np.zeros((4, nby * d, nbx * d), np.uint8)
    padded[:, :h, :w] = raster
    blocks = make_blocks_vectorized(padded, d)                                    # [n,4,d,d]
    want = torch.stack([val_transform(image=b.transpose(1, 2, 0))["image"][:3] for b in blocks])   # [n,3,d,d] f32
    dev_r = torch.from_numpy(raster).to(DEV)
    got = ops.split_normalize_u8(dev_r, d, 0, nby * nbx, MEAN, STD, 3).cpu().permute(0, 3, 1, 2)
    assert torch.equal(got, want)
    if nby * nbx > 2:
        part = ops.split_normalize_u8(dev_r, d, 1, nby * nbx - 2, MEAN, STD, 3).cpu().permute(0, 3, 1, 2)
        assert torch.equal(part, want[1:-1])
    with pytest.raises(RuntimeError):
        ops.split_normalize_u8(dev_r, d, 0, nby * nbx + 1, MEAN, STD, 3)


def test_blank_raster_flag_and_nhwc_entry():
    """scripts/inference.py:60-62 is_valid_tile as a device reduction; the NHWC entry of predict_classes gives the class
    maps of the NCHW one; infer_tile through run_blocks equals the host Tiler path"""
    from deadtrees_amd import ops
    from deadtrees_amd.data.synthetic import MEAN, STD
    from deadtrees_amd.deployment.tiler import infer_tile
    from deadtrees_amd.network.unet import UNetHIP
    band = torch.zeros(777, 333, dtype=torch.uint8, device=DEV)
    band[5::7] = 255
    assert int(ops.band_has_data(band)) == 0
    band[700, 300] = 17
    assert int(ops.band_has_data(band)) == 1
    m = UNetHIP().to(DEV).eval()
    rng = np.random.default_rng(3)
    raster = rng.integers(0, 256, (4, 200, 330), dtype=np.uint8)
    x = ops.split_normalize_u8(torch.from_numpy(raster).to(DEV), 128, 0, 6, MEAN, STD, 3)
    a = m.predict_classes(x, dtype="uint8", nhwc=True)
    b = m.predict_classes(x.permute(0, 3, 1, 2).contiguous(), dtype="uint8")
    assert torch.equal(a, b)

    class _Inf:   # the two entry points of PyTorchInference on this model
        def run_u8(self, t, device=None):
            return m.predict_classes(ops.normalize_u8(t.to(DEV), MEAN, STD, 3), dtype="uint8", nhwc=True)

        def run_blocks(self, r, d, first, count):
            return m.predict_classes(ops.split_normalize_u8(r, d, first, count, MEAN, STD, 3), dtype="uint8", nhwc=True)

    class _Old:
        run_u8 = _Inf.run_u8

    on_dev = infer_tile(_Inf(), raster, subtile=128, batch_size=4, device=DEV)
    no_blocks = infer_tile(_Old(), raster, subtile=128, batch_size=4, device=DEV)
    host = infer_tile(_Inf(), raster, subtile=128, batch_size=4, device=DEV, on_device=False)
    assert on_dev.shape == (200, 330) and np.array_equal(on_dev, host) and np.array_equal(no_blocks, host)

    # the directory loop of scripts/inference.py:71-115 over in-memory rasters: blank rasters (band 1 all 0 / 255) are
    # skipped without a forward pass, the others give infer_tile's map; an inference object that reads 3 band planes gets
    # only those uploaded (same map); rank r of 2 takes every second raster
    from deadtrees_amd.deployment.tiler import infer_rasters

    class _Rgb(_Inf):
        in_channels = 3

    blank = raster.copy()
    blank[0] = np.where(blank[0] > 127, 255, 0)
    queue = [("a", raster), ("blank", blank), ("c", raster[:, :128, :256])]
    got = dict(infer_rasters(_Rgb(), queue, subtile=128, batch_size=4, device=DEV))
    assert list(got) == ["a", "blank", "c"] and got["blank"] is None
    assert np.array_equal(got["a"], host) and got["c"].shape == (128, 256)
    assert infer_tile(_Inf(), blank, subtile=128, batch_size=4, device=DEV, on_device=False, skip_blank=True) is None
    assert infer_tile(_Inf(), blank, subtile=128, batch_size=4, device=DEV) is not None      # the filter is opt-in here
    r1 = dict(infer_rasters(_Rgb(), queue, subtile=128, batch_size=4, device=DEV, rank=1, world=2))
    assert list(r1) == ["blank"]
