"""reference deadtrees/loss/gwdl.py -> deadtrees_amd.loss.callables"""
from deadtrees_amd.loss.callables import GeneralizedWassersteinDiceLoss  # noqa: F401
