"""reference deadtrees/deployment/tiler.py -> deadtrees_amd.deployment.tiler"""
from deadtrees_amd.deployment.tiler import TileInfo, Tiler, divisible_without_remainder, inspect_tile  # noqa: F401
