"""Hot-path configuration.  Carries only the knobs the train step reads, with the
reference's names and defaults (lib/config.py:47-274) and the derived values of
Config._set_value (:317-330).  The reference's class-of-AttrDicts, yaml merge and CLI
plumbing are out of scope.
"""
import math
from types import SimpleNamespace as NS

import numpy as np


def make_config(backbone="resnet101", image_size=1024, batch_size=4, train_rois_per_image=200,
                dev_switch=True, loss_choice="ot", ot_L=5, roi_method="roi_align", gpu_count=1,
                num_classes=81, buffer_size=1, loss_fac=1000.0, conv_precision="fp32"):
    """Defaults = reference defaults; configs/104/meta_104_conv.yaml sets SWITCH, LOSS_CHOICE='ot',
    BUFFER_SIZE=1, LOSS_FAC=1000 and structure 'beta' with UPSAMPLE_FAC=1 is the only Dev branch
    that runs."""
    c = NS()
    # CONV_PRECISION (not in the reference): 'fp32' = exact fp32 MFMA convolutions (default, the headline);
    # 'bf16' = bf16-input / fp32-accumulate MFMA convolutions, BASELINE configs[4]'s reduced-precision path
    c.MODEL = NS(BACKBONE=backbone, BACKBONE_STRIDES=[4, 8, 16, 32, 64], CONV_PRECISION=conv_precision)
    c.DATASET = NS(NUM_CLASSES=num_classes)
    c.RPN = NS(ANCHOR_SCALES=(32, 64, 128, 256, 512), ANCHOR_RATIOS=[0.5, 1, 2], ANCHOR_STRIDE=1,
               NMS_THRESHOLD=0.7, TRAIN_ANCHORS_PER_IMAGE=256, PRE_NMS_LIMIT=6000,
               POST_NMS_ROIS_TRAINING=2000, POST_NMS_ROIS_INFERENCE=1000,
               TARGET_POS_THRES=0.7, TARGET_NEG_THRES=0.3)
    c.MRCNN = NS(USE_MINI_MASK=True, MINI_MASK_SHAPE=(56, 56), POOL_SIZE=7, MASK_POOL_SIZE=14,
                 MASK_SHAPE=[28, 28],
                 # not in the reference: run the mask head only on the RoI slots that can hold positives
                 # (see MaskRCNN.forward); identical loss and gradients, ~1/3 of the mask-head work
                 MASK_HEAD_ON_POSITIVE_SLOTS=False)
    c.DATA = NS(IMAGE_MAX_DIM=image_size, BBOX_STD_DEV=np.array([0.1, 0.1, 0.2, 0.2], np.float32),
                IMAGE_SHAPE=np.array([image_size, image_size, 3]), MAX_GT_INSTANCES=100)
    c.ROIS = NS(TRAIN_ROIS_PER_IMAGE=train_rois_per_image, ROI_POSITIVE_RATIO=0.33,
                ASSIGN_ANCHOR_BASE=224.0, METHOD=roi_method)
    c.TRAIN = NS(BATCH_SIZE=batch_size, INIT_LR=0.01, MOMENTUM=0.9, WEIGHT_DECAY=0.0001,
                 CLIP_GRAD=True, MAX_GRAD_NORM=5.0, BN_LEARN=False, FPN_OT_LOSS=False,
                 FPN_OT_LOSS_FAC=1.0,
                 # LOSS_SCALE (not in the reference, which has no 16-bit path): the loss is multiplied by it before
                 # backward and the gradients divided by it afterwards (workflow.backward_scaled).  fp16 operands: the
                 # mask head's gradients g = dy * (y > 0) fall below fp16's normal range (6e-5) unscaled -- the
                 # BatchNorm-weight gradients of the default backward form were 15-30 % off the dense form's
                 LOSS_SCALE=1024.0 if conv_precision == "fp16" else 1.0)
    c.DEV = NS(SWITCH=dev_switch, BUFFER_SIZE=buffer_size, EFFECT_AFER_EP_PERCENT=0.0,
               MULTI_UPSAMPLER=False, UPSAMPLE_FAC=1.0, LOSS_CHOICE=loss_choice,
               OT_ONE_DIM_FORM="conv", OT_L=ot_L, OT_EPSILON=1.0, LOSS_FAC=loss_fac, INST_LOSS=False,
               FEAT_BRANCH_POOL_SIZE=14, DIS_REG_LOSS=False, ASSIGN_BOX_ON_ALL_SCALE=False,
               BASELINE=False, BIG_SUPERVISE=False, STRUCTURE="beta", DIS_UPSAMPLER=False,
               BIG_FEAT_DETACH=True, CLS_MERGE_FEAT=False, CLS_MERGE_MANNER='simple_add', CLS_MERGE_FAC=0.5,
               BIG_LOSS_FAC=1.0)
    c.MISC = NS(SEED=2000, GPU_COUNT=gpu_count)
    c.TEST = NS(DET_MAX_INSTANCES=100, DET_MIN_CONFIDENCE=0, DET_NMS_THRESHOLD=0.3)      # lib/config.py:150-157
    c.CTRL = NS(PHASE='train')
    if image_size % 64 != 0:
        raise ValueError("Image size must be dividable by 2 at least 6 times (lib/model.py:43-47)")
    c.MODEL.BACKBONE_SHAPES = np.array([[int(math.ceil(image_size / s)), int(math.ceil(image_size / s))]
                                        for s in c.MODEL.BACKBONE_STRIDES])
    return c
