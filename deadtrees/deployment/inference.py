"""reference deadtrees/deployment/inference.py -> deadtrees_amd.deployment.inference"""
from deadtrees_amd.deployment.inference import (Inference, ONNXInference, PyTorchEnsembleInference,  # noqa: F401
                                                PyTorchInference)
