"""MI355X-native cost-volume hot path of Practical Deep Stereo (NeurIPS 2018).

Matching -> Regularization -> SubpixelMap behind the reference's module surfaces, computed by
hand-written HIP kernels for gfx950 (libpds_hip.so, C ABI in include/pds_hip.h).
"""
from practicaldeepstereo_nips2018_amd import errors
from practicaldeepstereo_nips2018_amd.embedding import Embedding
from practicaldeepstereo_nips2018_amd.estimator import SubpixelMap
from practicaldeepstereo_nips2018_amd.loss import SubpixelCrossEntropy
from practicaldeepstereo_nips2018_amd.matching import Matching, MatchingOperation
from practicaldeepstereo_nips2018_amd.network import PdsNetwork
from practicaldeepstereo_nips2018_amd.regularization import (ContractionBlock3d, ExpansionBlock3d,
                                                            Regularization)

__all__ = ['errors', 'Embedding', 'SubpixelMap', 'SubpixelCrossEntropy', 'Matching', 'MatchingOperation', 'PdsNetwork', 'ContractionBlock3d',
           'ExpansionBlock3d', 'Regularization']
