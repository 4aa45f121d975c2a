"""reference deadtrees/utils/data_handling.py:9-34 -> deadtrees_amd.deployment.tiler"""
from deadtrees_amd.deployment.tiler import make_blocks_vectorized, unmake_blocks_vectorized  # noqa: F401
