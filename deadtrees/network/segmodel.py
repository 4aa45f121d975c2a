"""reference deadtrees/network/segmodel.py -> deadtrees_amd.network.segmodel"""
from deadtrees_amd.network.segmodel import (SemSegment, concat_extra, create_combined_batch,  # noqa: F401
                                            initialize_weights)
