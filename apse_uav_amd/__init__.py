"""apse_uav_amd -- MI355X-native implementation of the per-frame hot path of
vision-agh/apse_uav's ``dcnn/`` subsystem (Mask R-CNN R-101-FPN detection +
instance masks + triplet-embedding tracking -> ``*_dcnn_data.csv``).

Layout mirrors the reference's ``dcnn/`` packages for the path only:
``engines`` (RcnnTracker, TrackPredictor), ``networks`` (TrackRCNN, AssociationHead),
``structures`` (Instances, Boxes, ObjectInstances, WindowMask), ``utils`` (mask_utils,
csv_log, resample) over ``csrc/`` (hand-written HIP kernels behind the C ABI of
``include/apse_hip.h``).
"""
__version__ = "0.1.0"
