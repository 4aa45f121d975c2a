"""reference deadtrees/data/deadtreedata.py -> deadtrees_amd.data.deadtreedata"""
from deadtrees_amd.data.deadtreedata import DeadtreeDatasetConfig, DeadtreesDataModule, train_transform, val_transform  # noqa: F401
