"""deadtrees_amd — MI355X-native U-Net segmentation hot path of cwerner/deadtrees.

Host side (python, mirrors the reference's SemSegment / DataModule surface) over hand-written gfx950
HIP kernels reached through the C ABI of ``libdeadtrees_hip.so`` (include/deadtrees_hip.h).
"""
__version__ = "0.1.0"

import os as _os

# more hardware queues for the process's HIP streams (ROCclr default: 4, dealt round-robin): the training step runs the
# main chain, the weight-gradient side stream and — data parallel — RCCL's streams side by side; with 4 queues two of
# them can land on one queue and serialise.  Only effective before the HIP runtime initialises;
# a value the user has set wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
