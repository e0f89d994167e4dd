"""hyslam_amd — MI355X-native ORB extraction + Hamming matching for hySLAM's per-frame hot path.

The compute lives in libhyslam_amd.so (hand-written HIP for gfx950) behind the C ABI of include/hyslam_amd.h;
this package is the thin host-side mirror of the reference's FeatureExtractor / Stereomatcher interfaces.
"""
from .features import (Camera, FeatureExtractorSettings, FeatureMatcher, FeatureMatcherSettings, HsError, KP_DTYPE,  # noqa: F401
                       ORBExtractor, ORBFactory, ORBVocabulary, Stereomatcher, stereo_params)
from ._native import FrameView, LM_DTYPE, ProjParams, VocabTree  # noqa: F401
