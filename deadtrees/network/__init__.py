from deadtrees.network.segmodel import SemSegment  # noqa: F401

__all__ = ["SemSegment"]
