"""Host-side logic of the product path that needs no GPU: layer spec, state_dict conversion, SemSegment
surface and its error behaviour, block split/merge, transforms, and the loud failure without a device."""
import hashlib
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spec_counts_and_buckets():
    from deadtrees_amd.network.spec import build_spec, smp_param_shapes
    s = build_spec(3, 2)
    assert s.n_true_params == 24_436_514
    assert build_spec(3, 1).n_true_params == 24_436_369
    assert build_spec(4, 3).n_true_params == 24_436_514 + 7 * 7 * 64 + 9 * 16 + 1
    assert len(s.convs) == 47 and sum(c.bn_key is not None for c in s.convs) == 46
    # buckets tile the flat buffer in reverse (gradient-ready) order without gaps
    spans = [(lo, hi) for _, lo, hi in s.buckets]
    assert spans[0][1] == s.n_params and spans[-1][0] == 0
    for (lo, hi), (lo2, hi2) in zip(spans, spans[1:]):
        assert lo == hi2
    assert [round((hi - lo) * 4 / 1e6, 1) for lo, hi in spans] == [12.6, 52.5, 27.3, 4.5, 0.9]
    shapes = smp_param_shapes(s)
    assert shapes["decoder.blocks.0.conv1.0.weight"] == (256, 768, 3, 3)
    assert shapes["segmentation_head.0.bias"] == (2,)


def test_state_dict_matches_oracle_keys_and_roundtrips():
    from deadtrees_amd.network.unet import UNetHIP
    from oracle.unet_ref import make_oracle
    ref = make_oracle(4, 3, seed=3)
    m = UNetHIP(in_channels=4, classes=3)
    missing, unexpected = m.load_state_dict(ref.state_dict())
    assert not missing and not unexpected
    sd = m.state_dict()
    assert list(sd.keys()) != [] and set(sd.keys()) == set(ref.state_dict().keys())
    for k, v in ref.state_dict().items():
        assert torch.equal(sd[k], v), k
    with pytest.raises(RuntimeError):
        m.load_smp_state_dict({k: v for k, v in ref.state_dict().items() if "layer3" not in k})
    bad = dict(ref.state_dict())
    bad["encoder.conv1.weight"] = torch.zeros(64, 3, 7, 7)
    with pytest.raises(RuntimeError):
        m.load_smp_state_dict(bad)


def test_product_path_fails_loudly_without_gpu():
    from deadtrees_amd.network.unet import UNetHIP
    from deadtrees_amd.loss.seg_loss import seg_loss
    m = UNetHIP()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        seg_loss(torch.zeros(1, 2, 8, 8), torch.zeros(1, 8, 8, dtype=torch.int64))
    src = open(os.path.join(ROOT, "deadtrees_amd", "network", "unet.py")).read()
    for mod in ("unet.py", "segmodel.py"):
        assert "import oracle" not in open(os.path.join(ROOT, "deadtrees_amd", "network", mod)).read()
    assert "oracle" not in src.replace("oracle/", "")


def test_no_product_module_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "deadtrees_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert "from oracle" not in txt and "import oracle" not in txt, os.path.join(dirpath, f)


def test_semsegment_surface_and_errors():
    from deadtrees_amd.network.segmodel import SemSegment, create_combined_batch, cosine_lr
    from deadtrees_amd.utils.config import default_network, default_training
    m = SemSegment(default_network(), default_training())
    assert m.classes == ["background", "deadtree"] and m.classes_int == [0, 1] and m.in_channels == 3
    assert m.encoder_weights is None and m.loss_names == ["GDICE", "FOCAL"]
    assert m.alpha == 0.01
    m.current_epoch = 200
    assert m.alpha == 0.99
    (opt,), (sch,) = m.configure_optimizers()
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["lr"] == 3e-4 and sch.T_max == 10
    assert any(k.startswith("model.encoder.conv1") for k in m.state_dict())
    with pytest.raises(NotImplementedError):
        SemSegment(default_network(architecture="fancynet"), default_training())
    with pytest.raises(NotImplementedError):
        SemSegment(default_network(architecture="efficientunet++"), default_training())
    with pytest.raises(AssertionError):
        SemSegment(default_network(losses=["GDICE", "DICE"]), default_training())
    with pytest.raises(NotImplementedError):
        SemSegment(default_network(losses=["GDICE", "HINGE"]), default_training())
    with pytest.raises(AssertionError):
        SemSegment(default_network(losses=["FOCAL"]), default_training())
    # batch plumbing (segmodel.py:31-54)
    a = (torch.zeros(2, 3, 4, 4), torch.zeros(2, 4, 4, dtype=torch.int64), torch.zeros(2, 2, 4, 4),
         torch.zeros(2, 4, 4, dtype=torch.int64), [{"file": "a"}, {"file": "b"}])
    b = (torch.ones(1, 3, 4, 4), torch.ones(1, 4, 4, dtype=torch.int64), torch.ones(1, 2, 4, 4),
         torch.ones(1, 4, 4, dtype=torch.int64), [{"file": "c"}])
    img, mask, dist, lu, stats = create_combined_batch({"main": a, "extra_0": b})
    assert img.shape[0] == 3 and [s["file"] for s in stats] == ["a", "b", "c"]
    # scheduler closed form == torch
    p = torch.nn.Parameter(torch.zeros(1))
    o = torch.optim.Adam([p], lr=3e-4)
    s = torch.optim.lr_scheduler.CosineAnnealingLR(o, T_max=10)
    for ep in range(1, 8):
        o.step()
        s.step()
        assert cosine_lr(3e-4, ep, 10) == pytest.approx(o.param_groups[0]["lr"], rel=1e-9)


def test_checkpoint_roundtrip_and_inference_errors(tmp_path):
    from deadtrees_amd.deployment.inference import PyTorchInference
    from deadtrees_amd.network.segmodel import SemSegment
    from deadtrees_amd.utils.config import default_network, default_training
    m = SemSegment(default_network(in_channels=4, classes=["a", "b", "c"]), default_training())
    p = tmp_path / "m.ckpt"
    m.save_checkpoint(p)
    m2 = SemSegment.load_from_checkpoint(p)
    assert m2.in_channels == 4 and len(m2.classes) == 3
    assert torch.equal(m2.model.flat_params, m.model.flat_params)
    with pytest.raises(ValueError):
        PyTorchInference(tmp_path / "m.onnx")
    inf = PyTorchInference(p)
    with pytest.raises(TypeError):
        inf.run(np.zeros((3, 8, 8)))


def test_blocks_match_reference_known_answer(golden_dir):
    from deadtrees_amd.deployment.tiler import Tiler, make_blocks_vectorized, unmake_blocks_vectorized
    z = np.load(os.path.join(golden_dir, "blocks.npz"))
    np.testing.assert_array_equal(make_blocks_vectorized(z["toy"], 2), z["toy_blocks"])        # tests/test_tiler.py:56-77
    np.testing.assert_array_equal(unmake_blocks_vectorized(z["toy_blocks"][:, 0], 2, 4, 4), z["toy"][0])
    rng = np.random.default_rng(int(z["big_seed"]))
    big = rng.integers(0, 256, (4, 512, 512), dtype=np.uint8)
    blocks = make_blocks_vectorized(big, 256)
    assert hashlib.sha256(blocks.tobytes()).hexdigest() == str(z["big_blocks_sha256"])
    assert hashlib.sha256(unmake_blocks_vectorized(blocks[:, 1], 256, 512, 512).tobytes()).hexdigest() == str(
        z["big_merged_sha256"])
    t = Tiler(tile_shape=(512, 512), subtile_shape=(256, 256))
    t.load_array(big[:, :500, :300])
    batches = t.get_batches()
    assert batches.shape == (4, 4, 256, 256)
    assert t.put_batches(batches[:, 2]) is None
    np.testing.assert_array_equal(t.result, big[2, :500, :300])
    with pytest.raises(ValueError):
        Tiler(tile_shape=(2048, 2048), subtile_shape=(256, 250))


def test_transforms_and_synthetic_data():
    from deadtrees_amd.data.deadtreedata import DeadtreesDataModule, val_transform
    from deadtrees_amd.data.synthetic import MEAN, STD, synth_batch
    img = np.random.default_rng(0).integers(0, 256, (16, 16, 4), dtype=np.uint8)
    out = val_transform(image=img)["image"]
    assert out.shape == (4, 16, 16) and out.dtype == torch.float32
    want = (img.astype(np.float64) / 255.0 - np.array(MEAN)) / np.array(STD)
    np.testing.assert_allclose(out.numpy(), want.transpose(2, 0, 1), rtol=1e-5, atol=1e-5)
    a, la = synth_batch(2, 64, 64, 3, 2, seed=5)
    b, lb = synth_batch(2, 64, 64, 3, 2, seed=5)
    assert torch.equal(a, b) and torch.equal(la, lb) and int(la[0].sum()) == 0 and la.dtype == torch.int64
    dm = DeadtreesDataModule(train_dataloader_conf={"batch_size": 2}, synthetic_batches=2, tile_size=64)
    dm.setup(in_channels=3, classes=2)
    batch = next(iter(dm.train_dataloader()))
    img_t, mask, dist, lu, stats = batch["main"]
    assert img_t.shape == (2, 3, 64, 64) and dist.shape == (2, 2, 64, 64) and len(stats) == 2
    assert isinstance(next(iter(dm.test_dataloader())), tuple)


def test_distmap_matches_reference_golden(golden_dir):
    """product-side one_hot2dist (loader step) against the maps the imported reference produced"""
    import glob
    from deadtrees_amd.data.distmap import distmaps_for_batch
    for path in sorted(glob.glob(os.path.join(golden_dir, "losses_*.npz"))):
        z = np.load(path)
        got = distmaps_for_batch(torch.from_numpy(z["mask"]), z["logits"].shape[1])
        np.testing.assert_array_equal(got.numpy(), z["distmap"])


def test_restricted_lightning_checkpoint_reader(tmp_path):
    """deadtrees_amd.utils.ckpt: a Lightning-style .ckpt that pickles objects of modules we do not have (the
    reference's omegaconf hyper-parameters) and a hostile reduce is read without importing or running anything;
    tensors come back bit-exact (incl. views with offsets / strides) and the network conf follows from the shapes."""
    import sys
    import types
    from deadtrees_amd.utils import ckpt as C
    from oracle.unet_ref import make_oracle
    ref = make_oracle(4, 3, seed=5)
    sd = {f"model.{k}": v for k, v in ref.state_dict().items()}
    base = torch.arange(24, dtype=torch.float32)
    sd["model.extra_view"] = base[4:].view(4, 5).t()            # offset + non-contiguous strides
    fake = types.ModuleType("omegaconf_like")

    class DictConfig(dict):
        def __reduce__(self):
            return (DictConfig, (), {"_content": dict(self), "_flags": None})

        def __setstate__(self, st):
            self.update(st["_content"])
    DictConfig.__module__, DictConfig.__qualname__ = "omegaconf_like", "DictConfig"
    fake.DictConfig = DictConfig
    sys.modules["omegaconf_like"] = fake

    class Boom:
        def __reduce__(self):
            return (os.system, (f"touch {tmp_path}/pwned",))
    p = tmp_path / "lightning.ckpt"
    try:
        torch.save({"state_dict": sd, "epoch": 7, "hyper_parameters": DictConfig(network=DictConfig(a=1)),
                    "callbacks": {"x": Boom()}}, str(p))
    finally:
        del sys.modules["omegaconf_like"]
    with pytest.raises(Exception):
        torch.load(str(p), weights_only=True)                   # what the plain safe loader does with such a file
    ck = C.read_checkpoint_tensors(p)
    assert not (tmp_path / "pwned").exists()                    # the reduce was never executed
    assert ck["epoch"] == 7 and isinstance(ck["hyper_parameters"], C._Inert)
    got = C.lightning_state_dict(p)
    assert set(got) == {k[len("model."):] for k in sd}
    for k, v in sd.items():
        assert torch.equal(got[k[len("model."):]], v), k
    got.pop("extra_view")
    conf = C.infer_network_conf(got)
    assert conf["in_channels"] == 4 and conf["classes"] == ["background", "conifers", "deciduous"]
    assert conf["encoder_name"] == "resnet34"
    with pytest.raises(RuntimeError):
        (tmp_path / "junk.ckpt").write_bytes(b"not a zip")
        C.read_checkpoint_tensors(tmp_path / "junk.ckpt")


def test_semsegment_loads_reference_style_lightning_checkpoint(tmp_path):
    """SemSegment.load_from_checkpoint on a file shaped like the reference's ModelCheckpoint output (state_dict under
    'model.', pickled hyper-parameter objects of an absent package): weights arrive, conf inferred from shapes."""
    import sys
    import types
    from deadtrees_amd.network.segmodel import SemSegment
    from oracle.unet_ref import make_oracle
    ref = make_oracle(3, 2, seed=9)
    fake = types.ModuleType("omegaconf_like2")

    class Cfg:
        pass
    Cfg.__module__, Cfg.__qualname__ = "omegaconf_like2", "Cfg"
    fake.Cfg = Cfg
    sys.modules["omegaconf_like2"] = fake
    p = tmp_path / "bestmodel.ckpt"
    try:
        torch.save({"state_dict": {f"model.{k}": v for k, v in ref.state_dict().items()},
                    "hyper_parameters": {"network": Cfg(), "training": Cfg()}, "pytorch-lightning_version": "1.5.0"},
                   str(p))
    finally:
        del sys.modules["omegaconf_like2"]
    m = SemSegment.load_from_checkpoint(p)
    assert m.in_channels == 3 and m.model.spec.classes == 2
    sd = m.model.state_dict()
    for k, v in ref.state_dict().items():
        if k.endswith("num_batches_tracked"):
            continue
        assert torch.equal(sd[k].cpu(), v), k


def test_load_from_checkpoint_is_inert_for_hostile_files_and_reads_pickled_parameters(tmp_path):
    """SemSegment.load_from_checkpoint (what PyTorchInference calls, deployment/inference.py:39) on a Lightning-style
    file whose pickle names os.system and whose state_dict holds nn.Parameter objects (pickled through
    torch._utils._rebuild_parameter / _rebuild_parameter_with_state): nothing runs, the weights arrive."""
    from deadtrees_amd.network.segmodel import SemSegment
    from oracle.unet_ref import make_oracle
    ref = make_oracle(3, 2, seed=11)
    sd = {}
    for k, v in ref.state_dict().items():
        sd[f"model.{k}"] = v
    first = "model.encoder.conv1.weight"
    par = torch.nn.Parameter(sd[first].clone())
    par.some_attribute = "state travels with the parameter"      # -> _rebuild_parameter_with_state (4 arguments)
    sd[first] = par
    sd["model.encoder.bn1.weight"] = torch.nn.Parameter(sd["model.encoder.bn1.weight"].clone())   # 3 arguments

    class Boom:
        def __reduce__(self):
            return (os.system, (f"touch {tmp_path}/pwned",))
    p = tmp_path / "hostile.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": Boom(), "callbacks": [Boom()]}, str(p))
    m = SemSegment.load_from_checkpoint(p)
    assert not (tmp_path / "pwned").exists()
    got = m.model.state_dict()
    for k, v in ref.state_dict().items():
        if not k.endswith("num_batches_tracked"):
            assert torch.equal(got[k].cpu(), v.detach()), k


def test_resunet_architecture_surface_and_reference_import_names(tmp_path, monkeypatch):
    """architecture "resunet" builds (reference segmodel.py:66-67), its state_dict carries the reference module's
    names (identity_conv, 1x1 head), the other in-tree architectures still raise NotImplementedError with a reason;
    the `deadtrees.*` modules the reference's callers import resolve; teardown writes the two CSV files."""
    import importlib
    from deadtrees.network.segmodel import SemSegment
    from deadtrees_amd.utils.config import default_network, default_training
    from oracle.resunet_ref import make_resunet_oracle
    m = SemSegment(default_network(architecture="ResUnet"), default_training())
    ref = make_resunet_oracle(3, 2, seed=1)
    assert set(m.model.state_dict()) == set(ref.state_dict())
    assert sum(p.numel() for p in ref.parameters()) == m.model.spec.n_true_params == 24_699_410
    m.model.load_state_dict(ref.state_dict())
    for k, v in ref.state_dict().items():
        assert torch.equal(m.model.state_dict()[k], v), k
    for arch in ("resunet++", "efficientunet++"):
        with pytest.raises(NotImplementedError):
            SemSegment(default_network(architecture=arch), default_training())
    # smp.UnetPlusPlus (reference segmodel.py:64-65): both spellings, smp key names, its parameter count
    from oracle.unetpp_ref import make_unetpp_oracle
    for arch in ("unet++", "UnetPlusPlus"):
        mpp = SemSegment(default_network(architecture=arch), default_training())
        assert mpp.model.spec.decoder_kind == "unetplusplus"
    refpp = make_unetpp_oracle(3, 2, seed=1)
    assert set(mpp.model.state_dict()) == set(refpp.state_dict())
    assert sum(p.numel() for p in refpp.parameters()) == mpp.model.spec.n_true_params == 26_078_754
    mpp.model.load_state_dict(refpp.state_dict())
    for k, v in refpp.state_dict().items():
        assert torch.equal(mpp.model.state_dict()[k], v), k
    for name in ("deadtrees.network.segmodel", "deadtrees.data.deadtreedata", "deadtrees.deployment.inference",
                 "deadtrees.deployment.tiler", "deadtrees.loss.losses", "deadtrees.loss.gdl", "deadtrees.loss.gwdl",
                 "deadtrees.utils.data_handling"):
        importlib.import_module(name)
    from deadtrees.deployment.inference import ONNXInference, PyTorchEnsembleInference, PyTorchInference  # noqa: F401
    from deadtrees.data.deadtreedata import val_transform  # noqa: F401
    with pytest.raises(ValueError):
        ONNXInference("model.ckpt")
    m.stats["train"].update(["a.tif", "a.tif", "b.tif"])
    m.stats["val"].update(["c.tif"])
    monkeypatch.chdir(tmp_path)
    m.teardown()
    assert (tmp_path / "train_stats.csv").read_text() == "filename,count\na.tif,2\nb.tif,1\n"
    assert (tmp_path / "val_stats.csv").read_text() == "filename,count\nc.tif,1\n"


def test_winograd_layer_selection_and_weight_image_tables():
    """host side of the Winograd path (no GPU): which layers get a forward / data-gradient weight image, the image
    offsets, and the shape predicates of the C ABI (dt_conv2d_winograd_supported, dt_conv2d_wgrad_winograd_supported)"""
    import ctypes as C
    from deadtrees_amd import _lib, ops
    from deadtrees_amd.network.unet import UNetHIP
    lib = _lib.load()
    eng = UNetHIP().engine
    tab, n, blocks, total, offs = eng._wino_table(torch.device("cpu"), False)
    tabd, nd, blocksd, totald, offsd = eng._wino_table(torch.device("cpu"), True)
    convs = {c.key: c for c in eng.spec.convs}
    # forward: every 3x3 stride-1 layer with Cin % 16 == 0 and Cout % 64 == 0 — the encoder blocks and decoder blocks 0-2
    want = {k for k, c in convs.items() if c.k == 3 and c.stride == 1 and c is not eng.spec.head and c.cin % 16 == 0 and c.cout % 64 == 0}
    assert set(offs) == want and n == len(want) == 35      # 29 encoder convs + 6 of decoder blocks 0-2
    assert not any(k.startswith(("decoder.blocks.3", "decoder.blocks.4")) for k in offs)
    # data gradient: Cin / Cout swap -> also the 32-channel output layers whose input has a multiple of 64 channels
    wantd = {k for k, c in convs.items() if c.k == 3 and c.stride == 1 and c is not eng.spec.head and c.cout % 16 == 0 and c.cin % 64 == 0}
    assert set(offsd) == wantd and len(wantd) >= n
    # images are packed back to back, 16 * Cin * Cout floats each, 16-byte aligned
    end = 0
    for row in tab.tolist():
        w_off, u_off, cin, cout, first = row
        assert u_off == end and u_off % 4 == 0
        end += 16 * cin * cout
    assert end == total and blocks == sum(((r[3] + 63) // 64) * ((r[2] // 4 + 3) // 4) for r in tab.tolist())
    # shape predicates
    d = ops.conv_desc(32, 128, 128, 64, 0, 0, 64, 3, 1, 1)
    assert lib.dt_conv2d_winograd_supported(C.byref(d)) == 1 and lib.dt_conv2d_wgrad_winograd_supported(C.byref(d)) == 1
    assert lib.dt_conv2d_winograd_stat_rows(C.byref(d)) == 256          # one row per persistent workgroup (2048 tiles)
    d192 = ops.conv_desc(2, 32, 32, 64, 0, 0, 192, 3, 1, 1)              # 3 channel blocks: one row per spatial tile
    assert lib.dt_conv2d_winograd_stat_rows(C.byref(d192)) == 2 * 2 * 2
    dsmall = ops.conv_desc(2, 32, 32, 64, 0, 0, 128, 3, 1, 1)            # 16 tiles on 16 workgroups
    assert lib.dt_conv2d_winograd_stat_rows(C.byref(dsmall)) == 16
    for bad in (ops.conv_desc(32, 128, 128, 64, 0, 0, 32, 3, 1, 1),        # Cout 32
                ops.conv_desc(32, 128, 128, 24, 0, 0, 64, 3, 1, 1),        # Cin 24: odd number of 8-channel chunks
                ops.conv_desc(32, 128, 128, 64, 0, 0, 64, 1, 1, 0),        # 1x1
                ops.conv_desc(32, 128, 128, 64, 0, 2, 64, 3, 1, 1),        # zero-insertion (transposed) input
                ops.conv_desc(64, 512, 512, 64, 0, 0, 64, 3, 1, 1)):       # 4 GiB operands
        assert lib.dt_conv2d_winograd_supported(C.byref(bad)) == 0
    s2 = ops.conv_desc(32, 128, 128, 64, 0, 0, 128, 3, 2, 1)
    assert lib.dt_conv2d_winograd_supported(C.byref(s2)) == 0 and lib.dt_conv2d_wgrad_winograd_supported(C.byref(s2)) == 0
    d32 = ops.conv_desc(32, 256, 256, 32, 0, 0, 128, 3, 1, 1)              # dec.3 data gradient: forward form only
    assert lib.dt_conv2d_winograd_supported(C.byref(d32)) == 1 and lib.dt_conv2d_wgrad_winograd_supported(C.byref(d32)) == 0
    assert lib.dt_conv2d_wgrad_winograd_workspace(C.byref(d)) > 0 and lib.dt_conv2d_wgrad_winograd_workspace(C.byref(d32)) == 0
