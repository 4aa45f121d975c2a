"""CPU oracle for the deadtrees U-Net hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / reported baseline.
The product path (``deadtrees_amd``) never imports this package and fails
loudly when its HIP library is missing.

Pinning status:
  * losses / one-hot / distance maps / block split-merge: pinned against the
    reference's own importable modules (``oracle/make_golden.py`` imports
    ``deadtrees.loss`` and ``deadtrees.utils.data_handling`` from
    /root/reference and writes ``tests/golden/*.npz``) and against the
    reference's known-answer test (tests/test_tiler.py:56-77).
  * GWDICE (loss/gwdl.py) values and gradients: pinned the same way (golden vectors from the imported reference).
  * training augmentation (``augment_ref.py``): geometric steps are the numpy calls albumentations uses; the
    brightness/contrast LUT is restated from albumentations' published algorithm -> PARITY UNPINNED for that step
    (albumentations is not installed).  Ensemble vote: checked against ``torch.mode`` itself.
  * network (smp ``Unet`` + ``resnet34``): the arithmetic lives in the
    un-vendored third-party package ``segmentation_models_pytorch>=0.2.1``
    (reference setup.py:47) which is absent here -> PARITY UNPINNED for the
    network topology; the oracle restates the published smp/torchvision
    topology from torch primitives.  Round 2: the DECODER
    wiring (nearest x2 upsample, cat([x, skip]), two Conv2dReLU per block, channel
    arithmetic) is now pinned by execution — ``oracle/make_golden_resunet.py`` loads the
    reference's in-tree twin of smp's decoder block (network/extra/resunet/decoder.py,
    extra/modules.py; they import nothing that is absent) by file path and writes
    ``tests/golden/resunet_decoder.npz``; the resnet34 ENCODER stays unpinned.
  * ResUnet decoder (``resunet_ref.py``): pinned the same way (outputs, input gradients and
    parameter gradients of the reference's own ``ResUnetDecoder``).
  * Unet++ decoder (``unetpp_ref.py``): dense wiring pinned by executing the reference's
    ``extra/efficientunetplusplus/decoder.py`` (smp's UnetPlusPlusDecoder constructor / forward loop) with smp's plain
    decoder block (``oracle/make_golden_unetpp.py`` -> ``tests/golden/unetpp_decoder.npz``).
  * bf16 training oracle (``unet_bf16_ref.py``): the fp32 restatement with bf16 roundings at
    the HIP path's storage points; ``dtype=float32`` reproduces the fp32 oracle exactly.
"""
