"""MI355X-native Point-Pair-Feature matching / voting engine (host-side Python mirror).

The product is the C-ABI HIP library declared in include/ppf_hip.h; this package only loads it,
mirrors the reference's operator interface for the path (PPF3DDetector.trainModel / match /
match_S2B, /root/reference/include/CloudProcessing.h:205-236,442,495) and carries the synthetic
scene generators the measurements use.
"""
__version__ = "0.1.0"
