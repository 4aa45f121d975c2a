"""reference deadtrees/loss/losses.py -> deadtrees_amd.loss.callables (fused HIP loss behind the same callables)"""
from deadtrees_amd.loss.callables import (BoundaryLoss, CrossEntropy, DiceLoss, FocalLoss, SurfaceLoss,  # noqa: F401
                                          class2one_hot, one_hot2dist)
