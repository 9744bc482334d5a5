"""feature_intertwiner_amd -- MI355X (gfx950) implementation of the Feature
Intertwiner hot path: RoIAlign (crop_and_resize), RoIPool, NMS and the Sinkhorn
optimal-transport intertwiner loss as hand-written HIP kernels behind a C ABI
(include/fi_capi.h), surfaced with the reference's own operator interface:

    from feature_intertwiner_amd.roi_align.crop_and_resize import CropAndResizeFunction
    from feature_intertwiner_amd.roi_align.roi_align import RoIAlign
    from feature_intertwiner_amd.roi_pooling.functions.roi_pool import RoIPoolFunction
    from feature_intertwiner_amd.roi_pooling.modules.roi_pool import _RoIPooling
    from feature_intertwiner_amd.nms.nms_wrapper import nms
    from feature_intertwiner_amd.nms.pth_nms import pth_nms
    from feature_intertwiner_amd.OT_module import OptTrans

(the module paths mirror lib/roi_align, lib/roi_pooling, lib/nms, lib/OT_module.py of
the reference).  GPU only: there is no CPU or eager fallback.
"""
__version__ = "0.1.0"
