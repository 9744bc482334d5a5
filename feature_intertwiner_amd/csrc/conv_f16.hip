// conv_f16.hip -- the kernels of conv_bf16.hip on IEEE half operands (v_mfma_f32_32x32x16_f16, fp32 accumulation):
// fi_conv2d_forward_f16, fi_conv3x3_forward_f16w, fi_conv1x1_forward_f16w, fi_conv2d_weight_grad_f16.
// Same tiles, same LDS layouts, same epilogues; only the operand rounding (fp32 -> half, RNE, 11-bit significand,
// 5-bit exponent: values beyond 65504 become inf) and the MFMA instruction differ.
#define FI_E16_HALF 1
#include "conv_bf16.hip"
