"""deadtrees_amd — MI355X-native U-Net segmentation hot path of cwerner/deadtrees.

Host side (python, mirrors the reference's SemSegment / DataModule surface) over hand-written gfx950
HIP kernels reached through the C ABI of ``libdeadtrees_hip.so`` (include/deadtrees_hip.h).
"""
__version__ = "0.1.0"
