/*
 * fi_capi.h -- C ABI of libfi_hip.so, the MI355X (gfx950) implementation of the
 * Feature Intertwiner hot-path operators.
 *
 * This is the drop-in boundary: plain device pointers, sizes and a HIP stream
 * handle; no framework types.  Each entry point names the reference interface it
 * replaces (paths relative to the reference checkout).  The reference built three
 * torch.utils.ffi (cffi) extensions whose inner launchers already took raw device
 * pointers + ints + a stream; those launchers are the level mirrored here.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - tensors are fp32, contiguous, NCHW, exactly as in the reference;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *     calls enqueue work and return without synchronising (the reference's NMS
 *     performed a blocking D2H copy, lib/nms/src/nms_cuda.c:33-34 -- removed);
 *   - the callee owns no persistent device memory: outputs and workspaces are
 *     caller-allocated (sizes via the *_workspace_bytes helpers);
 *   - return value: FI_OK (0) or a negative FI_ERR_* code; the library never
 *     calls exit() (the reference launchers did: crop_and_resize_kernel.cu:186-191,
 *     roi_pooling_kernel.cu:117-122).  fi_last_error() gives a thread-local
 *     message for the last failing call;
 *   - functions are re-entrant and device-agnostic: they launch on the device
 *     that is current for the calling thread (as the reference did under
 *     nn.DataParallel's one-thread-per-GPU model).
 */
#ifndef FI_CAPI_H_
#define FI_CAPI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FI_OK 0
#define FI_ERR_INVALID_ARG (-1)
#define FI_ERR_HIP (-2)          /* a HIP runtime call / launch failed          */
#define FI_ERR_UNSUPPORTED (-3)  /* size outside what the kernels implement     */

typedef void *fi_stream_t;

/* Library identification: "fi_hip <version> gfx950". */
const char *fi_version(void);
/* Message for the most recent failing call on this thread ("" if none). */
const char *fi_last_error(void);

/* ------------------------------------------------------------------------
 * crop_and_resize  (RoIAlign, one bilinear tap per bin)
 * Replaces: crop_and_resize_gpu_forward   lib/roi_align/src/crop_and_resize_gpu.c:7-37
 *           CropAndResizeLaucher          lib/roi_align/src/cuda/crop_and_resize_kernel.cu:168-193
 *           (CPU twin crop_and_resize_forward, lib/roi_align/src/crop_and_resize.c:115-154,
 *            defines the arithmetic: separately rounded fp32 mul/add, floorf/ceilf taps).
 * image  [batch, depth, image_h, image_w]
 * boxes  [num_boxes, 4] = (y1, x1, y2, x2), normalised: 0 -> row 0, 1 -> row H-1
 * box_ind[num_boxes]   int32 image index of each box
 * crops  [num_boxes, depth, crop_h, crop_w]  -- every element is written (the
 *        reference zero-fills first, crop_and_resize_gpu.c:24-25; here the kernel
 *        writes the zeros/extrapolation value itself).
 * A box whose box_ind is outside [0, batch) produces zeros (GPU reference
 * behaviour, crop_and_resize_kernel.cu:35-38) and, if dev_status != NULL, sets
 * bit 0 of *dev_status (the CPU reference aborted: crop_and_resize.c:39-42).
 * ---------------------------------------------------------------------- */
int fi_crop_and_resize_forward(const float *image, const float *boxes,
                               const int32_t *box_ind, int num_boxes, int batch,
                               int depth, int image_h, int image_w, int crop_h,
                               int crop_w, float extrapolation_value, float *crops,
                               int32_t *dev_status, fi_stream_t stream);

/* Replaces: crop_and_resize_gpu_backward  lib/roi_align/src/crop_and_resize_gpu.c:40-69
 *           CropAndResizeBackpropImageLaucher  .../crop_and_resize_kernel.cu:196-221
 * grads [num_boxes, depth, crop_h, crop_w] -> grads_image [batch, depth, H, W].
 * Every element of grads_image is WRITTEN by the call (the reference zero-fills it
 * first, crop_and_resize_gpu.c:57; here the tile-owner kernel's plain stores are
 * the zero fill -- no memset, no global atomics).  Deterministic: a workgroup owns
 * a tile of the map, accumulates every contribution to a cell in fp64 in LDS and
 * rounds once to fp32, so the result does not depend on scheduling (the
 * reference's atomicAdd kernel sums in arrival order; the CPU reference sums in
 * box order in fp32 -- both are within 2e-5 * max|grad| of this).  The round-4
 * scatter kernel (zero fill + fp32 global atomics) stays behind
 * FI_CROP_BWD_SCATTER=1. */
int fi_crop_and_resize_backward(const float *grads, const float *boxes,
                                const int32_t *box_ind, int num_boxes, int batch,
                                int depth, int image_h, int image_w, int crop_h,
                                int crop_w, float *grads_image, fi_stream_t stream);

/* Test hook: the bin assignment alone (tap rows/cols, lerp weights, in-range
 * flags), so parity tests can require it bit-exact.  Outputs [num_boxes, crop]. */
int fi_crop_and_resize_taps(const float *boxes, int num_boxes, int image_h,
                            int image_w, int crop_h, int crop_w, int32_t *y_valid,
                            int32_t *y0, int32_t *y1, float *y_frac,
                            int32_t *x_valid, int32_t *x0, int32_t *x1,
                            float *x_frac, fi_stream_t stream);

/* Pyramid form: one launch over all FPN levels.  Replaces the per-level
 * nonzero/gather/crop/cat/scatter loops of pyramid_roi_align (lib/layers.py:145-218)
 * and Dev.forward + _reshape_result (lib/sub_module.py:429-662): box i is cropped
 * from level_images_host[level[i] - 2] and written to row i of crops (original
 * RoI order, so no scatter-back is needed).  level[i] outside [2, 2+num_levels)
 * or a bad box_ind yields a zero row (the reference leaves unassigned rows zero).
 * level_images_host / level_h_host / level_w_host are HOST arrays of num_levels
 * entries (num_levels <= 8). */
int fi_pyramid_crop_forward(const float *const *level_images_host,
                            const int *level_h_host, const int *level_w_host,
                            int num_levels, const float *boxes,
                            const int32_t *box_ind, const int32_t *level,
                            int num_boxes, int batch, int depth, int crop_h,
                            int crop_w, float extrapolation_value, float *crops,
                            fi_stream_t stream);

/* Backward of the pyramid form; every level's gradient map is zero-filled first. */
int fi_pyramid_crop_backward(const float *grads, float *const *level_grads_host,
                             const int *level_h_host, const int *level_w_host,
                             int num_levels, const float *boxes,
                             const int32_t *box_ind, const int32_t *level,
                             int num_boxes, int batch, int depth, int crop_h,
                             int crop_w, fi_stream_t stream);

/* Channels-last variants of the pyramid form: every level map (and, backward, every level
 * gradient map) is [batch, H, W, depth]; crops / grads stay [num_boxes, depth, crop_h, crop_w];
 * results are bit-identical to the NCHW entry points.  With the channel axis innermost a
 * wavefront reads 64 channels of one tap as 256 contiguous bytes (NCHW: 64 different cache
 * lines), which is what lifts RoIAlign from ~50 % to >60 % of the HBM roofline.  The maps the
 * Dev stage crops (lib/sub_module.py:549-577) are consumed by nothing else, so their producer
 * (fi_conv2d_forward with output_layout = 1) writes them in this layout directly.
 * level may be NULL when num_levels == 1.  crop_h * crop_w <= 220. */
int fi_pyramid_crop_forward_nhwc(const float *const *level_images_host,
                                 const int *level_h_host, const int *level_w_host,
                                 int num_levels, const float *boxes,
                                 const int32_t *box_ind, const int32_t *level,
                                 int num_boxes, int batch, int depth, int crop_h,
                                 int crop_w, float extrapolation_value, float *crops,
                                 fi_stream_t stream);
int fi_pyramid_crop_backward_nhwc(const float *grads, float *const *level_grads_host,
                                  const int *level_h_host, const int *level_w_host,
                                  int num_levels, const float *boxes,
                                  const int32_t *box_ind, const int32_t *level,
                                  int num_boxes, int batch, int depth, int crop_h,
                                  int crop_w, fi_stream_t stream);

/* The two backward entry points WITHOUT the zero fill: the gradients are ADDED to what the level maps hold.
 * A map that is cropped by two launches (the Dev stage pools 7x7 and 14x14 from the same maps,
 * lib/sub_module.py:549-577) receives both contributions in one buffer -- the kernels add with atomics anyway --
 * instead of two cleared buffers and an add pass per level. */
int fi_pyramid_crop_backward_accumulate(const float *grads, float *const *level_grads_host,
                                        const int *level_h_host, const int *level_w_host,
                                        int num_levels, const float *boxes,
                                        const int32_t *box_ind, const int32_t *level,
                                        int num_boxes, int batch, int depth, int crop_h,
                                        int crop_w, fi_stream_t stream);
int fi_pyramid_crop_backward_nhwc_accumulate(const float *grads, float *const *level_grads_host,
                                             const int *level_h_host, const int *level_w_host,
                                             int num_levels, const float *boxes,
                                             const int32_t *box_ind, const int32_t *level,
                                             int num_boxes, int batch, int depth, int crop_h,
                                             int crop_w, fi_stream_t stream);

/* ------------------------------------------------------------------------
 * RoIPool (Caffe max pooling)
 * Replaces: roi_pooling_forward_cuda   lib/roi_pooling/src/roi_pooling_cuda.c:7-47
 *           ROIPoolForwardLaucher      lib/roi_pooling/src/roi_pooling_kernel.cu:95-125
 * features [batch, channels, height, width]
 * rois     [num_rois, 5] = (batch index, x1, y1, x2, y2) in pixels
 * output, argmax [num_rois, channels, pooled_h, pooled_w]; argmax is the flat index
 * into the whole features tensor, -1 for an empty bin (kernel.cu:73-86).
 * ---------------------------------------------------------------------- */
int fi_roi_pool_forward(const float *features, const float *rois, int num_rois,
                        int batch, int channels, int height, int width,
                        int pooled_h, int pooled_w, float spatial_scale,
                        float *output, int32_t *argmax, fi_stream_t stream);

/* Replaces: roi_pooling_backward_cuda  lib/roi_pooling/src/roi_pooling_cuda.c:49-88
 *           ROIPoolBackwardLaucher     lib/roi_pooling/src/roi_pooling_kernel.cu:205-234
 * bottom_grad [batch, channels, height, width] is zero-filled by the call, then
 * every pooled cell scatters its gradient to its argmax (fp32 atomics) under the
 * same feasibility conditions as the reference's gather (kernel.cu:147-190). */
int fi_roi_pool_backward(const float *top_grad, const float *rois,
                         const int32_t *argmax, int num_rois, int batch,
                         int channels, int height, int width, int pooled_h,
                         int pooled_w, float spatial_scale, float *bottom_grad,
                         fi_stream_t stream);

/* ------------------------------------------------------------------------
 * Greedy NMS on score-sorted boxes
 * Replaces: gpu_nms  lib/nms/src/nms_cuda.c:17-67  +  _nms  lib/nms/src/cuda/nms_kernel.cu:73-83
 *           cpu_nms  lib/nms/src/nms.c:4-69 (defines the `>=` comparison used when strict == 0)
 * boxes   [batch, num_boxes, box_stride] fp32, rows sorted by DESCENDING score
 *         (both reference callers pre-sort: lib/layers.py:103-105, 690-691);
 *         columns 0..3 are the two corner points (x1,y1,x2,y2) or (y1,x1,y2,x2) --
 *         IoU with the +1 pixel convention is symmetric in that choice;
 *         box_stride >= 4 (5 for the reference's [.., score] rows).
 * strict  0: suppress when IoU >= thresh (CPU reference, the parity spec);
 *         1: suppress when IoU >  thresh (CUDA reference, nms_kernel.cu:63).
 * max_keep  <= 0: keep everything; > 0: stop after max_keep survivors per image
 *         (the caller truncates to proposal_count anyway, lib/layers.py:130).
 * keep_out [batch, num_boxes] int64 indices into the sorted rows, in visit order;
 *          entries past num_out[b] are left untouched.
 * num_out  [batch] int32.
 * workspace: fi_nms_workspace_bytes(batch, num_boxes) bytes (the suppression
 *          bit-matrix; the reference allocated/freed it per call, nms_cuda.c:28-35).
 * No host synchronisation: the greedy scan runs on the GPU.
 * ---------------------------------------------------------------------- */
size_t fi_nms_workspace_bytes(int batch, int num_boxes);
int fi_nms_sorted(const float *boxes, int batch, int num_boxes, int box_stride,
                  float thresh, int strict, int max_keep, int64_t *keep_out,
                  int32_t *num_out, void *workspace, fi_stream_t stream);

/* ------------------------------------------------------------------------
 * Sinkhorn term of the OT intertwiner loss (no native counterpart in the
 * reference: lib/OT_module.py:104-135 runs 2L+6 small torch kernels per problem
 * inside a Python loop over the batch, :100-101).
 * x, y   [num_problems, S, D] fp32 (rows = samples, e.g. the 256 critic channels)
 * cost_mode 0: cosine cost, rows normalised by (||row||_2 + 1e-20) inside (:110-113);
 *         1: C_ij = ||x_i - y_j||_2 (:106-109);
 *         2: C = 1 - x y^T on rows the caller already normalised (lets an autograd
 *            wrapper own the normalisation and its backward)
 * eps_inv = 1/epsilon (the module stores the inverse, :13), L iterations.
 * loss   [num_problems]   <P, C> with P = a K b^T
 * plan   [num_problems, S, S] or NULL: the transport plan P (treated as a
 *        constant by the reference when no_bp_P_L, :129-131)
 * xn_out, yn_out [num_problems, S, D] or NULL: when both non-NULL (cost_mode 0 only)
 *        receive the normalised rows x^, y^ used for C.
 * Supported: 1 <= S <= 256, D >= 1.
 * ---------------------------------------------------------------------- */
int fi_sinkhorn_forward(const float *x, const float *y, int num_problems, int S,
                        int D, float eps_inv, int L, int cost_mode, float *loss,
                        float *plan, float *xn_out, float *yn_out,
                        fi_stream_t stream);

/* ------------------------------------------------------------------------
 * Per-class feature mean of the intertwiner statistics.
 * Replaces: Dev._assign_feat2cls  lib/sub_module.py:664-684 (Python loop over classes).
 * features [N, F]; gt [N] int32 class ids (0 = background, skipped);
 * feat [F, num_classes] (column c = mean of rows with class c, 0 if absent);
 * cnt  [num_classes] fp32 counts.
 * ---------------------------------------------------------------------- */
size_t fi_class_mean_workspace_bytes(int N, int F, int num_classes);
/* workspace: fi_class_mean_workspace_bytes(N, F, num_classes) bytes of device memory (partial sums
 * of the row chunks; summed in a fixed order, so the result is deterministic). */
int fi_class_mean_forward(const float *features, const int32_t *gt, int N, int F,
                          int num_classes, float *feat, float *cnt, float *workspace,
                          fi_stream_t stream);
/* grad_features[n, f] = grad_feat[f, gt[n]] / cnt[gt[n]] for foreground rows, else 0. */
int fi_class_mean_backward(const float *grad_feat, const int32_t *gt,
                           const float *cnt, int N, int F, int num_classes,
                           float *grad_features, fi_stream_t stream);

/* ------------------------------------------------------------------------
 * Dense convolution on the matrix cores (fp32 MFMA implicit GEMM), NCHW.
 * Replaces: the cuDNN convolutions behind every nn.Conv2d of the backbone / FPN / RPN /
 * Dev make-up layer / mask head (lib/sub_module.py:38-128, 147-228, 234-280, 308-345,
 * 750-787); the reference has no native code of its own for them.
 * x [N,Cin,H,W], weight [Cout,Cin,R,S], bias [Cout] or NULL, y [N,Cout,OH,OW] with
 * OH = (H + 2*pad_h - R)/stride_h + 1.  Epilogue: y = acc*scale[c] + bias[c] (+ residual) then
 * max(.,0) if relu; scale [Cout] and residual [N,Cout,OH,OW] may be NULL.  With scale = gamma /
 * sqrt(var+eps) and bias = beta + (conv_bias - mean)*scale this is conv + eval-mode BatchNorm
 * (the reference always evaluates BN with running statistics) + shortcut + ReLU.
 * out_h/out_w > 0 override the output size (window taps that fall outside the input read
 * zeros) -- used by the strided data gradient, which is a set of stride-1 correlations.
 * weight_layout 0: weight is [Cout,Cin,R,S] (as stored by the model); 1: [Cout,R,S,Cin]
 * (tap-major / channels-last; needs Cin % 16 == 0) -- selects the fast gather path; 2: as 1,
 * but the taps are applied in reverse order (tap (r,s) of the window uses weight[R-1-r][S-1-s]) --
 * with weight = W^T stored [Cin,R,S,Cout] that is the data gradient without a flipped copy.
 * 3: fragment-major 1x1 weights as written by fi_weight_transpose_batch (flag 1): the persistent form of the 1x1 / stride-1
 * kernel (conv1x1_ring_kernel); only for calls for which fi_conv1x1_ring_eligible() returns 1 (FI_ERR otherwise).
 * output_layout 0: y is [N,Cout,OH,OW]; 1: y is [N,OH,OW,Cout] (channels-last; Cout % 4 == 0, no
 * residual) -- used for the maps that only the channels-last RoIAlign consumes.
 * The data gradient of a stride-1 convolution is this same call on dY with the flipped,
 * transposed weight [Cin,Cout,R,S] and padding R-1-pad.
 * fi_conv2d_weight_grad: dweight [Cout,Cin,R,S] = sum over images and pixels of
 * dy (x) patches(x); zero-filled by the call, accumulated with fp32 atomics over a split
 * of the pixel range.  weight_layout 1 writes dweight as [Cout,R,S,Cin] (needs Cin % 128 == 0, or
 * Cin == 64 on a same-size stride-1 layer with H*W % 4 == 0 and 16-byte aligned x, dy).
 * dbias (optional, [Cout]) receives the bias gradient sum(dy) from the same pass over dy.
 * ---------------------------------------------------------------------- */
/* 1 when fi_conv2d_forward(_gated) accepts weight_layout 3 for this call: 1x1 / stride 1 / no padding, NCHW output,
 * Cin % 32 == 0 and >= 128, Cout % 128 == 0, H*W % 4 == 0, 16-byte aligned tensors, y below 4 GB, at least 256 tiles of
 * 128 pixels x 128 channels.  Pointers are only inspected for alignment (NULL = absent). */
int fi_conv1x1_ring_eligible(int N, int Cin, int H, int W, int Cout, int R, int S, int stride_h, int stride_w, int pad_h,
                             int pad_w, int output_layout, const float *x, const float *y, const float *residual,
                             const float *gate);
int fi_conv2d_forward(const float *x, const float *weight, const float *bias,
                      const float *scale, const float *residual, float *y, int N, int Cin,
                      int H, int W, int Cout, int R, int S, int stride_h, int stride_w,
                      int pad_h, int pad_w, int relu, int weight_layout, int out_h, int out_w,
                      int output_layout, fi_stream_t stream);
/* fi_conv2d_forward with one more epilogue operand: gate [N,Cout,OH,OW] (or NULL), y = (...) * (gate > 0).
 * The data gradient of a layer whose input is a ReLU output that nothing else reads leaves the kernel already
 * multiplied by that ReLU's mask (output_layout 0 only). */
int fi_conv2d_forward_gated(const float *x, const float *weight, const float *bias, const float *scale,
                            const float *residual, const float *gate, float *y, int N, int Cin, int H, int W,
                            int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int relu,
                            int weight_layout, int out_h, int out_w, int output_layout, fi_stream_t stream);
/* Backward of a 1x1 convolution y [N,K,HW] = W [K,C] . x [N,C,HW] + b whose output gradient has ONE non-zero channel
 * per row n: dy[n][cls[n]] = d[n] (d is [N,HW]), zeros elsewhere -- the mask head's conv5 under the mask loss
 * (lib/layers.py:905-934 reads the target class's mask of a RoI and nothing else):
 *     dx[n][c] = W[cls[n]][c] * d[n]  (times (x > 0) if gated),   dweight[cls[n]][c] += <d[n], x[n][c]>,
 *     dbias[cls[n]] += sum d[n]
 * dx may be NULL; dweight / dbias (may be NULL) are ACCUMULATED into (rows pre-summed per class in LDS, then fp32
 * atomics); cls int64 in [0, num_classes), num_classes <= 240 (FI_ERR_UNSUPPORTED otherwise); workspace:
 * fi_class_row_conv1x1_workspace_bytes(N, C). */
size_t fi_class_row_conv1x1_workspace_bytes(long N, int C);
int fi_class_row_conv1x1_backward(const float *d, const float *x, const float *weight, const int64_t *cls, float *dx,
                                  float *dweight, float *dbias, long N, int C, int HW, int num_classes, int gated,
                                  float *workspace, fi_stream_t stream);
/* The RPN's training path on SELECTED anchors (lib/layers.py:808-861 reads RPN.TRAIN_ANCHORS_PER_IMAGE sampled anchors
 * per image; lib/sub_module.py:256-280 is then two matrix products on those anchors' 3 x 3 patches).  Row r =
 * (image[r], anchor[r]) of the level-major anchor list (level l: heights[l] * widths[l] * anchors_per_location anchors,
 * anchor a on pixel a / anchors_per_location); image[r] < 0 marks a padding row.
 *   forward:  out [rows][9][channels], out[r][tap][c] = maps[l][image][c][h + tap/3 - 1][w + tap%3 - 1] or 0 outside
 *   backward: grads[l][image][c][h + ..][w + ..] += d[r][tap][c]   (fp32 atomics; the maps are [B][channels][H][W]) */
int fi_pyramid_patch_rows_forward(const void *const *maps, const int *heights, const int *widths, int levels,
                                  int anchors_per_location, const int64_t *image, const int64_t *anchor, long rows,
                                  int channels, float *out, fi_stream_t stream);
int fi_pyramid_patch_rows_backward(const float *d, void *const *grads, const int *heights, const int *widths, int levels,
                                   int anchors_per_location, const int64_t *image, const int64_t *anchor, long rows,
                                   int channels, fi_stream_t stream);
/* dst[i][:] = src[index[i]][:] and dst[index[i]][:] += src[i][:] for [rows][row_len] fp32 tensors and an int64 index
 * vector with DISTINCT entries (the Dev stage hands the feature extractor the 14 x 14 crops of its "small" RoIs in
 * level-major order, lib/sub_module.py:583-598; the backward adds their gradients into the crops' gradient). */
int fi_rows_gather(const float *src, const int64_t *index, float *dst, long n_index, long row_len, fi_stream_t stream);
int fi_rows_scatter_add(const float *src, const int64_t *index, float *dst, long n_index, long row_len, fi_stream_t stream);
/* dst[r][:] = (r < n_front ? front[r][:] : 0) + (src_row[r] >= 0 ? src[src_row[r]][:] : 0) for r < rows: the gradient of the
 * 14 x 14 crops when the mask head read the first n_front rows (a view) and the feature extractor the gathered rows --
 * one pass instead of fill + copy + scatter-add.  front may be NULL when n_front == 0. */
int fi_rows_combine(const float *front, long n_front, const float *src, const int64_t *src_row, float *dst, long rows,
                    long row_len, fi_stream_t stream);
/* The stem's max-pooling (lib/sub_module.py:44-45: SamePad2d + MaxPool2d(3, 2) == 3 x 3 / stride 2 windows clipped at
 * the right / bottom border, i.e. ceil_mode): y [planes][OH][OW], OH = (height - 2) / 2 + 1.  Backward recomputes the
 * arg-max from x with the framework's first-maximum rule and adds the gradients of the (up to 4) windows that select
 * an element in the framework's order; positive_only multiplies the result by (x > 0).  width % 4 == 0. */
int fi_maxpool3x3s2_forward(const float *x, float *y, long planes, int height, int width, fi_stream_t stream);
int fi_maxpool3x3s2_backward(const float *dy, const float *x, float *dx, long planes, int height, int width,
                             int positive_only, fi_stream_t stream);
/* Backward of a x2 nearest-neighbour upsampling (F.interpolate(scale_factor=2, mode='nearest') in the FPN's top-down
 * path, lib/sub_module.py:172-200): out [planes][height][width] = the 2 x 2 block sums of dy [planes][2*height][2*width],
 * summed rows first like the framework's kernel.  width even. */
int fi_sum2x2(const float *dy, float *out, long planes, int height, int width, fi_stream_t stream);
/* out = dy * (y > 0), n elements (out may alias dy): the ReLU part of the backward below on its own. */
int fi_relu_mask(const float *dy, const float *y, float *out, long n, fi_stream_t stream);
/* Gradients of conv + eval-mode BatchNorm from the weight gradient of the UNSCALED masked gradient g
 * (y = act(conv(x, W) * scale + shift), scale = gamma * inv_std, g = dy * (y > 0)):
 *     dW'[co] = sum_p g[co] (x) patches(x)   (fi_conv2d_weight_grad on g),   s[co] = sum_p g[co] = d beta
 *     dW[co]      = scale[co] * dW'[co]                                (in place)
 *     d gamma[co] += inv_std[co] * (<W[co], dW'[co]> + (conv_bias[co] - mean[co]) * s[co])
 *     d conv_bias[co] += scale[co] * s[co]
 * because sum_p g * conv(x, W) = <W, dW'>: no second pass over the activations.  dw [Cout][K] in tap-major
 * (dw_tap_major: k = tap*Cin + ci) or channel-major (k = ci*taps + tap) order, w likewise (w_tap_major).
 * dgamma / dbias may be NULL; conv_bias NULL = no bias. */
int fi_bn_fold_grad(float *dw, const float *w, const float *s, const float *scale, const float *mean,
                    const float *var, float eps, const float *conv_bias, float *dgamma, float *dbias, int Cout,
                    int Cin, int taps, int dw_tap_major, int w_tap_major, fi_stream_t stream);
/* The same for a FULLY CONNECTED layer + eval-mode BatchNorm + ReLU (the heads' full-window 7x7 "fc" convolutions and
 * their 1x1 convolutions on 1x1 maps, lib/sub_module.py:333-340, :707-716 -- nn.Conv2d + nn.BatchNorm2d + nn.ReLU, three
 * kernels each way in the reference): rows [M][N], the channel contiguous.  First pass of the backward:
 *     g = relu ? dy * (y > 0) : dy  -> g  [M][ld_out]   (operand of the weight gradient dW' = g^T x)
 *     g * scale[n]                  -> gs [M][ld_out]   (operand of the data gradient dx = gs W; scale NULL = 1)
 *     colsum[n] += sum_m g                              (= s of fi_bn_fold_grad; zero-filled by the call unless
 *                                                         FI_OUTPUTS_ZEROED)
 * g, gs, colsum may each be NULL; ld_out >= N lets the outputs land in zero-padded operands of the matrix products. */
int fi_rows_mask_scale(const float *dy, const float *y, const float *scale, float *g, float *gs, float *colsum, int M,
                       int N, int ld_out, int relu, int flags, fi_stream_t stream);
/* y = act(y * scale[n] + bias[n]) in place on rows [M][N] (scale / bias NULL = none): the epilogue of the fully
 * connected forward on the 16-bit kernels, whose K split accumulates with atomics (fi_gemm_nt_affine has it in its
 * reduction pass). */
int fi_rows_affine_act(float *y, const float *scale, const float *bias, int M, int N, int relu, fi_stream_t stream);
/* ... for n layers of one geometry in one launch (behind fi_conv2d_weight_grad_batch): HOST arrays of device pointers;
 * conv_bias / dgamma / dbias may be NULL tables or hold NULL entries. */
int fi_bn_fold_grad_batch(float *const *dw, const float *const *w, const float *const *s, const float *const *scale,
                          const float *const *mean, const float *const *var, float eps, const float *const *conv_bias,
                          float *const *dgamma, float *const *dbias, int n, int Cout, int Cin, int taps, int dw_tap_major,
                          int w_tap_major, fi_stream_t stream);
/* Backward of that fused epilogue (eval-mode BatchNorm folded into scale/shift, optional ReLU):
 * g = dy * (y > 0 | 1); dz = g * scale[c]; dshift[c] = sum g; dgamma[c] = sum g*(y-beta[c])/gamma[c].
 * layout 1: dy and y are channels-last [N,HW,C] (dz is still written [N,C,HW]; no residual/g_out).
 * dy, y, dz, g_out are [N,C,HW]; g_out (optional) receives g (the gradient of a fused residual);
 * residual (optional) is the shortcut that was added in the epilogue (y - residual = BN output).
 * dbias (optional, [C]) additionally receives dshift[c] * scale[c] -- the gradient of a convolution bias
 * that was folded into the shift.  flags: FI_OUTPUTS_ZEROED = the caller has already zero-filled the
 * accumulated outputs (dshift, dgamma, dbias; dweight, dbias of fi_conv2d_weight_grad), e.g. as slices of
 * one arena cleared once per step, so the per-call fills are skipped. */
#define FI_OUTPUTS_ZEROED 1
int fi_bn_act_backward(const float *dy, const float *y, const float *scale, const float *gamma,
                       const float *beta, const float *residual, int N, int C, int HW, int relu,
                       float *dz, float *g_out, float *dshift, float *dgamma, float *dbias,
                       int layout, int flags, fi_stream_t stream);
/* fi_conv2d_forward_gated for a batch of STATIC capacity N of which only the first *n_live_dev images are real (a DEVICE
 * count; NULL = all): tiles whose pixels all lie in images >= *n_live_dev are skipped (their outputs are not written) by
 * the general kernel (conv_fwd_kernel: the strided layers); the 3x3 / 1x1 stride-1 fast paths compute every image.  The
 * Dev stage's big branch (lib/sub_module.py:498-535) has a data-dependent number of rows -- n3 + 2 n4 + 3 n5 of up to
 * 3 * RoIs --, which the reference reads back to the host; here the host never learns it. */
int fi_conv2d_forward_live(const float *x, const float *weight, const float *bias, const float *scale,
                           const float *residual, const float *gate, float *y, int N, int Cin, int H, int W,
                           int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int relu,
                           int weight_layout, int out_h, int out_w, int output_layout, const int32_t *n_live_dev,
                           fi_stream_t stream);
int fi_conv2d_weight_grad(const float *x, const float *dy, float *dweight, int N, int Cin,
                          int H, int W, int Cout, int R, int S, int stride_h, int stride_w,
                          int pad_h, int pad_w, int weight_layout, float *dbias, int flags,
                          fi_stream_t stream);
/* n weight gradients of ONE geometry (the identical residual blocks of a ResNet stage: 23 in C4 of ResNet-101,
 * lib/sub_module.py:103-116) in one launch: x[i], dy[i], dweight[i], dbias[i] (dbias NULL, or one pointer per problem)
 * are HOST arrays of device pointers.  A layer of the C4 stage at batch 4 is a single round of short workgroups whose
 * fixed cost (prologue, atomic epilogue) is a third of its time; n layers together take fewer, longer pixel splits
 * per layer.  Same sums as n calls of fi_conv2d_weight_grad (fp32 atomics: the order of the partial sums differs).
 * One launch needs FI_OUTPUTS_ZEROED, tap-major / 1x1 weights with Cin % 128 == 0 and a same-size stride-1 layer;
 * otherwise the call loops over the problems.  At most FI_WGRAD_BATCH_MAX problems travel in one launch.
 * _bf16 / _f16: the same on the 16-bit-operand kernels (conv_bf16_wgrad_flat_kernel; weight_layout is ignored: tap-major). */
#define FI_WGRAD_BATCH_MAX 24
int fi_conv2d_weight_grad_batch(const float *const *x, const float *const *dy, float *const *dweight,
                                float *const *dbias, int n, int N, int Cin, int H, int W, int Cout, int R, int S,
                                int stride_h, int stride_w, int pad_h, int pad_w, int weight_layout, int flags,
                                fi_stream_t stream);
int fi_conv2d_weight_grad_batch_bf16(const float *const *x, const float *const *dy, float *const *dweight,
                                float *const *dbias, int n, int N, int Cin, int H, int W, int Cout, int R, int S,
                                int stride_h, int stride_w, int pad_h, int pad_w, int weight_layout, int flags,
                                fi_stream_t stream);
int fi_conv2d_weight_grad_batch_f16(const float *const *x, const float *const *dy, float *const *dweight,
                                float *const *dbias, int n, int N, int Cin, int H, int W, int Cout, int R, int S,
                                int stride_h, int stride_w, int pad_h, int pad_w, int weight_layout, int flags,
                                fi_stream_t stream);
/* Fully connected layers (the heads' full-window 7x7 "fc" convolutions lib/sub_module.py:707, :333, their nn.Linear
 * layers :744-747, the OT module's centre-tap Conv1d lib/OT_module.py:37-41):
 *     c [M,N] = act(a [M,K] . b [N,K]^T + bias [N])        (torch: F.linear; the reference: cuBLAS through torch)
 * on the row-major weight-gradient kernel (both operands contiguous along the reduction), the reduction split over
 * workgroups that each STORE their partial sums in their own slab of `workspace`
 * (fi_gemm_nt_workspace_bytes(M, N, K) bytes), reduced in a fixed order by a second kernel that adds the bias and
 * applies the ReLU -- deterministic, unlike an atomic split-K.  N % 128 == 0, K % 4 == 0, 16-byte aligned operands.
 * The two backward products (reductions over rows) are 1x1 convolutions: fi_conv2d_forward with the weight-like
 * operand [M,R] as `weight` and the [R,K] operand as a [1,R,1,K] input. */
size_t fi_gemm_nt_workspace_bytes(int M, int N, int K);
int fi_gemm_nt(const float *a, const float *b, const float *bias, float *c, int M, int N, int K, int relu,
               float *workspace, fi_stream_t stream);
/* ... with a DEVICE count of live rows (NULL = M): tiles of rows >= *m_live_dev are skipped and those rows of c (from the
 * count rounded up to the tile height) are written as zeros. */
int fi_gemm_nt_rows(const float *a, const float *b, const float *bias, float *c, int M, int N, int K, int relu,
                    float *workspace, const int32_t *m_live_dev, fi_stream_t stream);
/* ... with a per-column scale in the reduction pass as well: c = act((a . b^T) * scale [N] + bias [N]) -- a fully
 * connected layer with its eval-mode BatchNorm (scale = gamma / sqrt(var + eps), bias = beta + (conv_bias - mean) * scale)
 * and ReLU in one pass (scale NULL = fi_gemm_nt_rows). */
int fi_gemm_nt_affine(const float *a, const float *b, const float *scale, const float *bias, float *c, int M, int N, int K,
                      int relu, float *workspace, const int32_t *m_live_dev, fi_stream_t stream);

/* All layers' W^T for the data-gradient kernel in one launch: for every descriptor, src is
 * [rows][taps][cols] (a weight stored [Cout][R][S][Cin]) and dst becomes [cols][taps][rows]
 * ([Cin][R][S][Cout]).  tile_base = number of 32x32 tiles of all earlier descriptors
 * (taps * ceil(rows/32) * ceil(cols/32) each); the table lives in device memory. */
typedef struct {
    const void *src;
    void *dst;
    int rows, cols, taps;
    int pad_;                /* flags.  1: dst is FRAGMENT-MAJOR for the 1x1 ring kernel (taps == 1, rows % 32 == 0, cols % 32 == 0):
                              * the matrix D [M][K] = src^T (* row_scale) -- or D = src when flag 2 is also set -- stored per block
                              * of 32 rows x 16 columns as [2 halves of 4 k][64 lanes = (k / 8, row)][4 k], blocks in [M/32][K/16]
                              * order; M * K floats.  0: the plain transpose below. */
    long tile_base;
    const void *row_scale;   /* [rows] or NULL: dst[col][tap][row] = src[row][tap][col] * row_scale[row] -- the data
                              * gradient of conv + eval-BatchNorm reads W^T with the BatchNorm scale folded in */
} FiTransposeDesc;
int fi_weight_transpose_batch(const FiTransposeDesc *descs_dev, int n, long total_tiles,
                              fi_stream_t stream);

/* ------------------------------------------------------------------------
 * Proposal layer around NMS.  Replaces the tensor-op chain of proposal_layer,
 * lib/layers.py:71-139 (scores[:, :, 1] -> sort -> slice -> deltas * BBOX_STD_DEV ->
 * tools/box_utils.py apply_box_deltas :7-33 -> clip_boxes :36-60 -> cat) with one launch per call.
 * probs  [batch, A, prob_stride]: the foreground score of anchor a is probs[b][a][prob_offset]
 * deltas [batch, A, 4], anchors [A, 4] pixels (y1, x1, y2, x2), both 16-byte aligned
 * extra  [batch, E, 5] (y1, x1, y2, x2, score) external candidates that compete with the anchors, or NULL
 * dets   [batch, pre_nms, 5]: the pre_nms best candidates, descending score (ties: extra before anchors, then
 *        lower index), boxes decoded and clipped to [0, window_h] x [0, window_w]; pre_nms <= 8192.
 * bbox_std_host: 4 floats on the HOST. */
int fi_proposal_candidates(const float *probs, int prob_stride, int prob_offset, const float *deltas,
                           const float *anchors, const float *extra, int batch, int num_anchors,
                           int num_extra, int pre_nms, const float *bbox_std_host, float window_h,
                           float window_w, float *dets, fi_stream_t stream);
/* The same rows in nine small launches (round 6): the three radix passes on 32 workgroups per image, the compaction on 128
 * (one pair of global atomics per workgroup), the sort of the winners with the keys in registers (one workgroup per image),
 * decode + clip on 24 -- with their state in `workspace` (fi_proposal_workspace_bytes(batch) bytes, 16-byte aligned,
 * contents need not be preserved or cleared between calls).  fi_proposal_candidates is one workgroup per image walking every
 * score four times (4 workgroups on 256 CUs): 322 vs 102 us at 4 x 261 888 anchors. */
size_t fi_proposal_workspace_bytes(int batch);
int fi_proposal_candidates_ws(const float *probs, int prob_stride, int prob_offset, const float *deltas,
                              const float *anchors, const float *extra, int batch, int num_anchors, int num_extra,
                              int pre_nms, const float *bbox_std_host, float window_h, float window_w, float *dets,
                              void *workspace, size_t workspace_bytes, fi_stream_t stream);
/* proposals[b][j] = dets[b][keep[b][j]][0:4] / (norm_h, norm_w, norm_h, norm_w) for j < num[b], zero rows after
 * (lib/layers.py:131-137 without the host-side truncation to the shortest keep list). */
int fi_proposal_gather(const float *dets, int pre_nms, int det_stride, const int64_t *keep, int keep_stride,
                       const int32_t *num, int batch, int proposal_count, float norm_h, float norm_w,
                       float *proposals, fi_stream_t stream);

/* Eval-mode BatchNorm folded into the preceding convolution's epilogue (the reference always evaluates BN with running
 * statistics, lib/model.py:265-267): scale = gamma * rsqrt(var + eps), shift = beta - mean * scale (+ conv_bias * scale)
 * for every (conv, bn) pair of the model in ONE launch; the table lives in device memory. */
typedef struct {
    const void *gamma, *beta, *mean, *var, *conv_bias;   /* conv_bias may be NULL */
    void *scale, *shift;                                 /* outputs, [channels] each */
    int channels;
    float eps;
} FiBnFoldDesc;
int fi_bn_fold_batch(const FiBnFoldDesc *descs_dev, int n, int max_channels, fi_stream_t stream);

/* Data gradient of a stride-2 convolution assembled from its residue classes in ONE pass (no counterpart in the
 * reference: cuDNN's strided backward-data, reached through lib/sub_module.py's stride-2 Conv2d layers).
 * dx[p][h][w] = c<h&1><w&1>[p][h>>1][w>>1] (+ add[p][h][w]); class (a, b) is [planes][ceil((H-a)/2)][ceil((W-b)/2)]
 * or NULL (that class has no taps: zeros); add is [planes][H][W] or NULL.  Every element of dx is written. */
int fi_stride2_interleave(const float *c00, const float *c01, const float *c10, const float *c11,
                          const float *add, float *dx, long planes, int height, int width,
                          fi_stream_t stream);
/* ... with the result multiplied by (gate > 0), gate [planes][H][W] or NULL (see fi_conv2d_forward_gated). */
int fi_stride2_interleave_gated(const float *c00, const float *c01, const float *c10, const float *c11,
                                const float *add, const float *gate, float *dx, long planes, int height, int width,
                                fi_stream_t stream);

/* The live-count entry points on the 16-bit kernels (see fi_conv2d_forward_live / fi_gemm_nt_rows): the forward of a
 * static-capacity batch, and the weight-gradient kernel used as the GEMM of conv.linear (dweight [Cout = rows][Cin], zero
 * filled by the call unless FI_OUTPUTS_ZEROED) with a device count of live rows. */
int fi_conv2d_forward_live_bf16(const float *x, const float *weight, const float *bias, const float *scale,
                                const float *residual, const float *gate, float *y, int N, int Cin, int H, int W,
                                int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int relu,
                                int weight_layout, int out_h, int out_w, int output_layout, const int32_t *n_live_dev,
                                fi_stream_t stream);
int fi_conv2d_forward_live_f16(const float *x, const float *weight, const float *bias, const float *scale,
                               const float *residual, const float *gate, float *y, int N, int Cin, int H, int W,
                               int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int relu,
                               int weight_layout, int out_h, int out_w, int output_layout, const int32_t *n_live_dev,
                               fi_stream_t stream);
int fi_conv2d_weight_grad_rows_bf16(const float *x, const float *dy, float *dweight, int N, int Cin, int H, int W,
                                    int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int flags,
                                    const int32_t *rows_live_dev, fi_stream_t stream);
int fi_conv2d_weight_grad_rows_f16(const float *x, const float *dy, float *dweight, int N, int Cin, int H, int W,
                                   int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int flags,
                                   const int32_t *rows_live_dev, fi_stream_t stream);

/* The index side of the intertwiner RoI stage with static shapes (lib/sub_module.py:437-598 sizes its batches by the
 * RoIs' levels with one nonzero / .any() -- a host synchronisation -- per level): one launch turns level[N] (2..5) and the
 * class ids gt[N] (NULL at inference) into
 *   order[N]      the stable level-major permutation (torch.sort(level, stable=True)[1]);
 *   in that order: small_cls[N] = (level - 2) * K + gt on levels 2..4 with gt > 0, else 0 (the class index of ONE
 *                 fi_class_mean_forward over 3 K classes), small_gt[N] = gt on levels 2..4 else 0 (small_gt_all),
 *                 small_on[N] = level <= 4;
 *   the big batch (capacity >= 3 N rows): every (level l < 5, RoI of a level above l) pair in (l, RoI) order at the
 *                 front -- big_idx = the RoI, big_level = l, big_cls = (l - 2) * K + gt if level l has small rows at all
 *                 and gt > 0, else 0 -- and big_level = -1 behind the live count;
 *   counts[5]     n2, n3, n4, n5, live = n3 + 2 n4 + 3 n5 (device integers; nothing is read back).
 * small_*, big_* and counts may be NULL. */
int fi_dev_stage_index(const int32_t *level, const int32_t *gt, int N, int num_classes, int capacity, int64_t *order,
                       int32_t *small_cls, float *small_gt, uint8_t *small_on, int64_t *big_idx, int32_t *big_level,
                       int32_t *big_cls, int32_t *counts, fi_stream_t stream);

/* The statistics side of the meta loss (MaskRCNN.meta_loss, lib/model.py:143-210) on one rank with a history of one
 * step: _merge_feat_vec (:217-224) of the big and the small class features over L levels, the history-buffer update
 * (:150-166; `buffer` [F][K] and `buffer_cnt` [K] are updated IN PLACE, and left alone when the step has no small-object
 * statistics, lib/workflow.py:190), the class selection (:176-181) and the transposed operands of the pair loss:
 *   SMALL [K-1][F] (merged small features, foreground classes), BIG [K-1][F] (history after the update),
 *   on [K-1] (1 = small count > 0 and history count > 0), active_f [1], s_cnt_out [K] (kept for the backward).
 * feat[g][s][f][k] = base[g * lg + s * lk + f * ld + k] (G = the reference's nn.DataParallel ranks, S = levels; the sums
 * run over g first, then over s, as `.sum(0).sum(0)` does); cnt[g][s][k] = base[(g * S + s) * K + k].  Three launches; the tensor formulation
 * (feature_intertwiner_amd.intertwiner.meta_loss) is ~35.  workspace: fi_meta_stats_workspace_bytes(F, K).
 * Backward: d small_feat[g][s][f][k] = d SMALL[k-1][f] / (s_cnt[k] + 1e-20) * small_cnt[g][s][k], 0 for k = 0. */
size_t fi_meta_stats_workspace_bytes(int F, int K);
int fi_meta_stats_forward(const float *big_feat, const float *big_cnt, int big_ld, int big_lk, int big_lg,
                          const float *small_feat, const float *small_cnt, int small_ld, int small_lk, int small_lg, int G,
                          int S, int F, int K, float *buffer, float *buffer_cnt, float *s_cnt_out, float *SMALL, float *BIG,
                          float *on, float *active_f, float *workspace, fi_stream_t stream);
int fi_meta_stats_backward(const float *dsmall, const float *s_cnt, const float *small_cnt, int G, int S, int F, int K,
                           int out_ld, int out_lk, int out_lg, float *dfeat, fi_stream_t stream);
/* The same under data parallelism (one process per GPU; replaces the reference's gather to GPU 0 + _merge_feat_vec,
 * lib/model.py:217-224): fi_meta_stats_sums leaves this rank's count-weighted sums in ONE flat vector
 *   sums = [ sum big_feat*big_cnt (F K) | sum small_feat*small_cnt (F K) | sum big_cnt (K) | sum small_cnt (K) ]
 * which the caller all-reduces (sum) across the ranks, and fi_meta_stats_from_sums continues from the reduced vector
 * exactly as fi_meta_stats_forward continues from its own sums (means, history update, selection, operands): every rank
 * ends with the same history and the same operands.  1 + 3 launches around the collective.  Backward: the caller scales
 * d SMALL by the world size (gradients are averaged over the ranks afterwards) and calls fi_meta_stats_backward with
 * the GLOBAL s_cnt and its LOCAL small_cnt. */
int fi_meta_stats_sums(const float *big_feat, const float *big_cnt, int big_ld, int big_lk, int big_lg, const float *small_feat,
                       const float *small_cnt, int small_ld, int small_lk, int small_lg, int G, int S, int F, int K, float *sums,
                       fi_stream_t stream);
int fi_meta_stats_from_sums(const float *sums, int F, int K, float *buffer, float *buffer_cnt, float *s_cnt_out, float *SMALL,
                            float *BIG, float *on, float *active_f, float *workspace, fi_stream_t stream);

/* ------------------------------------------------------------------------
 * Target generation of one training step.
 * fi_rpn_targets: lib/layers.py:439-604 (generate_target) for a whole minibatch -- IoU of every anchor with the image's
 * ground-truth boxes (gt_class_ids [batch][max_gt] int64: > 0 object, < 0 COCO crowd box, 0 padding; gt_boxes pixels),
 * negatives IoU < neg_thres and not on a crowd box, positives IoU >= pos_thres plus every object's best anchor, at most
 * n_total / 2 positives and negatives up to n_total, sub-sampled at random: key_pos / key_neg [batch][anchors] hold one
 * uniform key in [1, 2) per anchor and "keep k at random" keeps the k largest keys (ties: lower anchor index).
 * match [batch][anchors] in {1, -1, 0}; deltas [batch][anchors][4] = refinement of the kept positives towards their best
 * object, divided by bbox_std_dev (HOST array of 4), zero elsewhere; row_image / row_anchor (optional, [batch][n_total]
 * int64): the anchors with a non-zero match in anchor order, -1 padded per image.  Workspace:
 * fi_rpn_targets_workspace_bytes.  max_gt <= 256.
 * fi_detection_targets: lib/layers.py:224-376 (generate_roi) -- proposals [batch][n_proposals][4] normalised, the first
 * num_proposals[b] (int64) of them real; positives IoU >= 0.5, negatives < 0.5 off crowd boxes; positive_cap =
 * int(rois_per_image * ROI_POSITIVE_RATIO) positives in descending key order first, then
 * min(floor(negatives_per_positive * pos - pos), available, rois_per_image - pos) negatives, then zero rows.  Outputs per
 * slot: rois, target_class_ids (int32), target_deltas (/ bbox_std_dev), mask_boxes = the RoI in the assigned object's
 * mini-mask frame (:301-322; the RoI itself when use_mini_mask == 0) and mask_box_ids = image * max_gt + object for the
 * crop_and_resize launch that cuts the mask targets, is_positive (1 / 0).  n_proposals <= 2048. */
size_t fi_rpn_targets_workspace_bytes(int batch, int anchors, int max_gt);
int fi_rpn_targets(const float *anchors, const int64_t *gt_class_ids, const float *gt_boxes, const float *key_pos,
                   const float *key_neg, int batch, int n_anchors, int max_gt, float neg_thres, float pos_thres,
                   int n_total, const float *bbox_std_dev, float *match, float *deltas, int64_t *row_image,
                   int64_t *row_anchor, void *workspace, fi_stream_t stream);
int fi_detection_targets(const float *proposals, const int64_t *num_proposals, const int64_t *gt_class_ids,
                         const float *gt_boxes, const float *key_pos, const float *key_neg, int batch, int n_proposals,
                         int max_gt, int rois_per_image, int positive_cap, double negatives_per_positive, int use_mini_mask,
                         const float *bbox_std_dev, float *rois, int32_t *target_class_ids, float *target_deltas,
                         float *mask_boxes, int32_t *mask_box_ids, float *is_positive, fi_stream_t stream);

/* The five detector losses (lib/layers.py:808-934) and their gradients in one pass: the RPN class / box losses on the
 * rows fi_rpn_targets listed (row_image / row_anchor [rpn_rows], -1 = no row; row_logits [rows][2], row_bbox [rows][4];
 * rpn_match [b][anchors], rpn_deltas [b][anchors][4]), the box head's class loss (soft-max cross entropy over all `rois`
 * rows, zero when the batch has no foreground) and box loss (smooth L1 on the positive RoIs' target-class row of
 * roi_bbox [rois][K][4]), the mask loss (sigmoid + binary cross entropy on the positive rows of mask_logits
 * [mask_rows][2][2][h][w] -- the target class's channel, BEFORE the pixel shuffle -- against mask_targets
 * [mask_rows][2h][2w]).  losses_and_factors[0..4] = rpn_class, rpn_bbox, mrcnn_class, mrcnn_bbox, mrcnn_mask;
 * [5..9] = d loss_k / d (stored gradient): grad_* hold each loss's gradient with respect to the network output up to
 * that factor (so backward is one scaling per tensor).  Partial sums are added in a fixed order: deterministic. */
size_t fi_detector_losses_workspace_bytes(int rpn_rows, int rois, int mask_rows);
int fi_detector_losses(const float *rpn_match, const float *rpn_deltas, const int64_t *row_image, const int64_t *row_anchor,
                       const float *row_logits, const float *row_bbox, int rpn_rows, int anchors,
                       const int32_t *roi_class_ids, const float *class_logits, const float *roi_deltas,
                       const float *roi_bbox, int rois, int num_classes, const int32_t *mask_class_ids,
                       const float *mask_logits, const float *mask_targets, int mask_rows, int mask_h, int mask_w,
                       float *grad_row_logits, float *grad_row_bbox, float *grad_class_logits, float *grad_roi_bbox,
                       float *grad_mask_logits, float *losses_and_factors, void *workspace, fi_stream_t stream);

/* ------------------------------------------------------------------------
 * bf16-input, fp32-accumulate variants (v_mfma_f32_32x32x16_bf16) for BASELINE configs[4]'s reduced-
 * precision conv path.  Same tensors as above (fp32 in memory, operands rounded to bf16 on their way into
 * LDS); selected by configuration, never by the fp32 headline.  fi_conv2d_forward_bf16 takes the
 * arguments of fi_conv2d_forward and needs Cin % 32 == 0 with tap-major weights (weight_layout 1 or 2;
 * FI_ERR_UNSUPPORTED otherwise -- callers fall back to the fp32 kernel, e.g. for the 3-channel stem).
 * fi_conv2d_weight_grad_bf16 writes dweight tap-major [Cout][R][S][Cin] (any sizes / strides).
 * ---------------------------------------------------------------------- */
int fi_conv2d_forward_bf16(const float *x, const float *weight, const float *bias,
                           const float *scale, const float *residual, float *y, int N, int Cin,
                           int H, int W, int Cout, int R, int S, int stride_h, int stride_w,
                           int pad_h, int pad_w, int relu, int weight_layout, int out_h, int out_w,
                           int output_layout, fi_stream_t stream);
/* 3x3 / stride 1 / pad 1 forward (flip_taps = 0) or data gradient (flip_taps = 1, weight = W^T [Cin][3][3][Cout])
 * with tap-major weights ALREADY in bf16 (uint16 bit patterns, converted once per step by the caller): the input
 * patch of a tile is staged once per 32 channels in LDS, weights go straight into the MFMA operand registers.
 * Needs W % 4 == 0 and W >= 16 (8 x 16 tiles; W % 16 == 0 for full ones) or W in {12, 14} (flat 128-pixel tiles: the
 * 14 x 14 RoI maps), Cin % 32 == 0,
 * Cout > 64 (FI_ERR_UNSUPPORTED otherwise); same epilogue as fi_conv2d_forward. */
int fi_conv3x3_forward_bf16w(const float *x, const uint16_t *weight_bf16, const float *bias, const float *scale,
                             const float *residual, float *y, int N, int Cin, int H, int W, int Cout, int relu,
                             int flip_taps, fi_stream_t stream);
/* 1x1 / stride 1 forward, or data gradient with weight = W^T [Cin][Cout] and the channel counts swapped, with
 * the [Cout][Cin] weights ALREADY in bf16: 128 pixels x 128 output channels per workgroup, the pixel tile staged 64
 * channels at a time, weights straight into the MFMA operand registers (the bf16 twin of the fp32 1x1 kernel).
 * HW = H * W.  Needs HW % 4 == 0, Cin % 64 == 0, Cout > 64 (FI_ERR_UNSUPPORTED otherwise); same epilogue as
 * fi_conv2d_forward.  Replaces the 1x1 nn.Conv2d modules of lib/sub_module.py:38-128 on the reduced-precision path. */
int fi_conv1x1_forward_bf16w(const float *x, const uint16_t *weight_bf16, const float *bias, const float *scale,
                             const float *residual, float *y, int N, int Cin, int HW, int Cout, int relu,
                             fi_stream_t stream);
int fi_conv2d_weight_grad_bf16(const float *x, const float *dy, float *dweight, int N, int Cin,
                               int H, int W, int Cout, int R, int S, int stride_h, int stride_w,
                               int pad_h, int pad_w, int flags, fi_stream_t stream);
/* The same kernels with the additions of fi_conv2d_forward_gated (epilogue operand gate, y *= (gate > 0)) and of
 * fi_conv2d_weight_grad's dbias (sum of dy over images and pixels, accumulated by the flat weight-gradient kernel from
 * the tiles it stages, by a channel-sum kernel for the other layers). */
int fi_conv2d_forward_gated_bf16(const float *x, const float *weight, const float *bias, const float *scale,
                                 const float *residual, const float *gate, float *y, int N, int Cin, int H, int W,
                                 int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int relu,
                                 int weight_layout, int out_h, int out_w, int output_layout, fi_stream_t stream);
int fi_conv3x3_forward_gated_bf16w(const float *x, const uint16_t *weight_bf16, const float *bias, const float *scale,
                                   const float *residual, const float *gate, float *y, int N, int Cin, int H, int W,
                                   int Cout, int relu, int flip_taps, fi_stream_t stream);
int fi_conv1x1_forward_gated_bf16w(const float *x, const uint16_t *weight_bf16, const float *bias, const float *scale,
                                   const float *residual, const float *gate, float *y, int N, int Cin, int HW, int Cout,
                                   int relu, fi_stream_t stream);
int fi_conv2d_weight_grad_db_bf16(const float *x, const float *dy, float *dweight, float *dbias, int N, int Cin, int H,
                                  int W, int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                                  int flags, fi_stream_t stream);

/* The same four entry points on IEEE half operands (v_mfma_f32_32x32x16_f16, fp32 accumulation) -- BASELINE
 * configs[4] names an "fp16 MFMA conv path".  Identical arguments, tiles and epilogues; weight_f16 holds half bit
 * patterns.  Operands beyond 65504 round to infinity (no loss scaling inside the library). */
int fi_conv2d_forward_f16(const float *x, const float *weight, const float *bias,
                          const float *scale, const float *residual, float *y, int N, int Cin,
                          int H, int W, int Cout, int R, int S, int stride_h, int stride_w,
                          int pad_h, int pad_w, int relu, int weight_layout, int out_h, int out_w,
                          int output_layout, fi_stream_t stream);
int fi_conv3x3_forward_f16w(const float *x, const uint16_t *weight_f16, const float *bias, const float *scale,
                            const float *residual, float *y, int N, int Cin, int H, int W, int Cout, int relu,
                            int flip_taps, fi_stream_t stream);
int fi_conv1x1_forward_f16w(const float *x, const uint16_t *weight_f16, const float *bias, const float *scale,
                            const float *residual, float *y, int N, int Cin, int HW, int Cout, int relu,
                            fi_stream_t stream);
int fi_conv2d_weight_grad_f16(const float *x, const float *dy, float *dweight, int N, int Cin,
                              int H, int W, int Cout, int R, int S, int stride_h, int stride_w,
                              int pad_h, int pad_w, int flags, fi_stream_t stream);
int fi_conv2d_forward_gated_f16(const float *x, const float *weight, const float *bias, const float *scale,
                                const float *residual, const float *gate, float *y, int N, int Cin, int H, int W,
                                int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int relu,
                                int weight_layout, int out_h, int out_w, int output_layout, fi_stream_t stream);
int fi_conv3x3_forward_gated_f16w(const float *x, const uint16_t *weight_f16, const float *bias, const float *scale,
                                  const float *residual, const float *gate, float *y, int N, int Cin, int H, int W,
                                  int Cout, int relu, int flip_taps, fi_stream_t stream);
int fi_conv1x1_forward_gated_f16w(const float *x, const uint16_t *weight_f16, const float *bias, const float *scale,
                                  const float *residual, const float *gate, float *y, int N, int Cin, int HW, int Cout,
                                  int relu, fi_stream_t stream);
int fi_conv2d_weight_grad_db_f16(const float *x, const float *dy, float *dweight, float *dbias, int N, int Cin, int H,
                                 int W, int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                                 int flags, fi_stream_t stream);

/* ------------------------------------------------------------------------
 * In-library kernel timing (HIP events recorded on the launch stream around
 * each kernel launch while enabled).  Used by bench.py for the roofline object;
 * off by default, zero cost when off.
 * ---------------------------------------------------------------------- */
enum {
    /* ids follow the device kernels (one template instance per common crop size); the
     * single-level and pyramid entry points launch the same kernels */
    FI_K_CROP_FWD_7X7 = 0,
    FI_K_CROP_FWD_14X14 = 1,
    FI_K_CROP_FWD_28X28 = 2,
    FI_K_CROP_FWD_GENERIC = 3,
    FI_K_CROP_BWD_7X7 = 4,
    FI_K_CROP_BWD_14X14 = 5,
    FI_K_CROP_BWD_28X28 = 6,
    FI_K_CROP_BWD_GENERIC = 7,
    FI_K_ROIPOOL_FWD = 8,
    FI_K_ROIPOOL_BWD = 9,
    FI_K_NMS_MASK = 10,
    FI_K_NMS_SCAN = 11,
    FI_K_SINKHORN = 12,
    FI_K_CLASS_MEAN = 13,
    /* conv kernels: id = base + 4*(BM == 128) + window class (0: 1x1, 1: 3x3, 2: 7x7, 3: other);
     * conv_fwd_kernel serves the forward pass and the stride-1 data gradient */
    FI_K_CONV_FWD = 14,      /* .. 21 */
    FI_K_CONV_WGRAD = 22,    /* .. 29 */
    FI_K_BN_ACT_BWD = 30,
    FI_K_CROP_FWD_NHWC_7X7 = 31,     /* channels-last RoIAlign: 7x7, 14x14, other */
    FI_K_CROP_FWD_NHWC_14X14 = 32,
    FI_K_CROP_FWD_NHWC_GENERIC = 33,
    FI_K_CROP_BWD_NHWC_7X7 = 34,
    FI_K_CROP_BWD_NHWC_14X14 = 35,
    FI_K_CROP_BWD_NHWC_GENERIC = 36,
    FI_K_CONV_BF16_FWD = 37,         /* bf16-input MFMA convolution (forward + data gradient) */
    FI_K_CONV_BF16_WGRAD = 38,
    FI_K_CONV3X3_PATCH = 39,         /* 3x3/s1/p1 forward + data gradient, input patch in LDS: 2-D tiles */
    FI_K_CONV3X3_PATCH_FLAT = 40,    /* ... flat 128-pixel tiles (14 x 14 RoI maps) */
    FI_K_CONV1X1_REG = 41,           /* 1x1/s1 forward + data gradient, weights in registers */
    FI_K_PROPOSAL_SELECT = 42,       /* fused pre-NMS stage of the proposal layer */
    FI_K_PROPOSAL_GATHER = 43,
    FI_K_STRIDE2_INTERLEAVE = 44,
    FI_K_GEMM_REDUCE = 45,           /* ordered reduction of fi_gemm_nt's split-K slabs (+ bias / ReLU) */
    FI_K_COUNT = 46
};
/* ------------------------------------------------------------------------
 * Optimiser step of the training iteration: torch.nn.utils.clip_grad_norm_(params, max_norm) followed by
 * torch.optim.SGD(momentum, weight_decay).step() (/root/reference/lib/workflow.py:226-230,
 * tools/utils.py:474-501) for all parameters in three launches: sum of squares per 8192-float chunk, one
 * workgroup that reduces the chunk sums in a fixed order to the norm and the clip factor
 * min(1, max_norm / (norm + 1e-6)), and one pass that applies
 *     g = grad * clip;  g += weight_decay * p;  buf = momentum * buf + g;  p -= lr * buf
 * (buf == NULL: no momentum; a zero-initialised buf reproduces torch's first step, buf = g).  The scaled
 * gradient is written back when the clip factor is not 1 (clip_grad_norm_ leaves it scaled).  param / grad /
 * buf of one descriptor must share one dense memory layout.  descs_dev: device array sorted by chunk_base,
 * chunk_base = running sum of fi_sgd_chunks(numel); partial_ws: total_chunks floats; norm_coef: 2 floats
 * (out: norm, clip factor).  max_norm <= 0 disables clipping.
 * ---------------------------------------------------------------------- */
typedef struct {
    void *param;
    void *grad;
    void *buf;
    long numel;
    long chunk_base;
    float weight_decay, lr, momentum, pad_;
} FiSgdDesc;
long fi_sgd_chunks(long numel);
int fi_sgd_clip_step(const FiSgdDesc *descs_dev, int n, long total_chunks, float max_norm, float *partial_ws,
                     float *norm_coef, fi_stream_t stream);
/* The same three launches with a non-finite guard for the 16-bit paths (static loss scale: one fp16 overflow in a
 * data gradient gives an inf / NaN norm, and the plain form -- like clip_grad_norm_ + step in the reference's
 * lib/workflow.py:226-230 -- would then write NaN into every weight).  norm_coef4: 4 floats, [2] is set to 1 when
 * this step's norm is not finite, in which case parameters, momentum buffers and gradients are left untouched (the
 * step is skipped); [3] counts skipped steps (zero it once).  No host synchronisation. */
int fi_sgd_clip_step_guarded(const FiSgdDesc *descs_dev, int n, long total_chunks, float max_norm, float *partial_ws,
                             float *norm_coef4, fi_stream_t stream);

/* Streaming copy of n_floats floats (16 bytes per lane) with exactly known memory traffic: the
 * calibration point for rocprofv3's FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md, HBM section). */
int fi_calib_copy(const float *src, float *dst, size_t n_floats, fi_stream_t stream);
void fi_prof_enable(int on);
void fi_prof_reset(void);
/* Synchronises the recorded events, then returns launches and summed ms. */
int fi_prof_get(int kernel_id, int *launches, float *total_ms);
const char *fi_prof_kernel_name(int kernel_id);

#ifdef __cplusplus
}
#endif
#endif /* FI_CAPI_H_ */
