"""reference deadtrees/loss/gdl.py -> deadtrees_amd.loss.callables"""
from deadtrees_amd.loss.callables import GeneralizedDiceLoss  # noqa: F401
