"""``deadtrees`` — the reference's import surface, served by the MI355X-native implementation in ``deadtrees_amd``.

The callers of the hot path in the reference import these names (scripts/inference.py:9-12, deployment/server.py:11,
deployment/inference.py:9-10, train.py via Hydra ``_target_: deadtrees.network.segmodel.SemSegment``,
configs/model/default.yaml:1):

    deadtrees.network.segmodel.SemSegment            deadtrees.data.deadtreedata.{DeadtreesDataModule, val_transform}
    deadtrees.deployment.inference.PyTorchInference   deadtrees.deployment.tiler.Tiler
    deadtrees.loss.{losses,gdl,gwdl}                  deadtrees.utils.data_handling.{make,unmake}_blocks_vectorized

Each module here only re-exports the ``deadtrees_amd`` object of the same name, so those callers run unmodified on
the HIP path.  Nothing of the reference's source is contained in this package.
"""
from deadtrees_amd import __version__  # noqa: F401
