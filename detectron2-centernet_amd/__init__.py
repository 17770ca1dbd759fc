"""MI355X-native CenterNet detection path behind the Detectron2 registry/config surface.

Only the hot path of ShawnNew/Detectron2-CenterNet is implemented here:
DLA-34 + DCNv2 backbone, hm/wh/reg heads, gaussian targets, focal/L1 losses, peak-NMS top-K decode,
all as hand-written HIP kernels for gfx950 reached through the C ABI in include/ctdet_hip.h.
"""
__version__ = "0.1.0"
