"""MI355X-native correspondence-and-pose hot path of seqwalt/PoseEstimation.

The compute lives in libposehip.so (hand-written HIP for gfx950) behind the C
ABI of include/posehip.h; this package is the thin Python mirror of the OpenCV
interfaces the reference calls on that path.
"""
from . import _lib
from ._lib import Context
from .matcher import NORM_HAMMING, BFMatcher, DMatch, min_dist_filter, ratio_test_match
from .pnp import SOLVEPNP_EPNP, solvePnP, solvePnPRansac, solvePnPRansacBatch

__all__ = ["_lib", "Context", "NORM_HAMMING", "BFMatcher", "DMatch", "min_dist_filter", "ratio_test_match",
           "SOLVEPNP_EPNP", "solvePnP", "solvePnPRansac", "solvePnPRansacBatch"]
