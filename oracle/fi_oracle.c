/*
 * fi_oracle.c -- CPU restatement of the Feature Intertwiner hot-path operators.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product path
 * (feature_intertwiner_amd/) never imports it and fails loudly without its HIP
 * library.
 *
 * Every function restates, in this file's own structure, the arithmetic of the
 * reference operator it cites (paths relative to the reference checkout).  All
 * arithmetic is fp32 with every multiply and add rounded separately (build with
 * -ffp-contract=off, no -ffast-math), because floor()/ceil() of the sampling
 * coordinate and the IoU-vs-threshold comparison decide integer outputs.
 *
 * PARITY PIN STATUS (see DESIGN.md section "Oracle"):
 *   - orc_sinkhorn / OptTrans : pinned -- checked against golden vectors generated
 *     in the build container by importing the reference's lib/OT_module.py
 *     (oracle/gen_golden_ot.py, fixtures tests/golden/ot_*.npz).
 *   - orc_crop_forward / orc_crop_taps (a1, the RoI bin assignment) : pinned by
 *     the reference compiled here -- lines 2-112 of lib/roi_align/src/
 *     crop_and_resize.c (`CropAndResizePerBox`, TH-free) are compiled UNMODIFIED
 *     in a temporary directory outside the repository by oracle/gen_golden_crop.py
 *     and run on 84 seeded cases (82 M elements, adversarial boxes, 8 crop sizes,
 *     both extrapolation values, the north-star 512x256x7x7 / 14x14); fixture
 *     tests/golden/crop_fwd.npz; tests/test_reference_crop_golden.py holds this
 *     restatement to it bit for bit.
 *   - orc_crop_backward, orc_nms, orc_roi_pool_* : PARITY UNPINNED by reference
 *     execution.  Those reference functions take TH tensors (<TH/TH.h>, PyTorch
 *     0.3 headers, absent from this image) or are CUDA, so no reference build
 *     exists here; the reference ships no tests or golden vectors.  These
 *     restatements are pinned only by hand-derived known-answer cases (written
 *     inline in tests/test_oracle_kat.py) and cross-checks against independent
 *     formulations (the adjoint identity against the pinned forward, float64
 *     brute-force NMS, max_pool2d for aligned RoIPool windows), which the tests
 *     state explicitly.  The Python on both sides of each is pinned by running
 *     the reference's wrappers (oracle/gen_golden_wrappers.py).
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK 0
#define ORC_BAD_BOX_INDEX (-2)
#define ORC_BAD_ARG (-1)

/* --------------------------------------------------------------------------
 * crop_and_resize: per-axis sampling table.
 *
 * Follows lib/roi_align/src/crop_and_resize.c:44-56 (scale), :54-56 (in_y),
 * :58 (range test), :71-73 (floor/ceil/lerp); the x axis is the same code at
 * :48-50, :79-81, :82, :89-91.
 * For one box edge pair (lo, hi) in normalised coordinates, an input extent
 * `extent` (H or W) and an output extent `crop`, entry k gets
 *   valid[k]  0 when the sample falls outside [0, extent-1]
 *   i0[k]     floorf(coord), i1[k] ceilf(coord), frac[k] = coord - i0[k]
 * ------------------------------------------------------------------------ */
typedef struct {
    int valid;
    int i0;
    int i1;
    float frac;
    float coord;
} orc_tap;

static void orc_axis_taps(float lo, float hi, int extent, int crop, orc_tap *t)
{
    const float span = (float)(extent - 1);
    float step = 0.0f;
    if (crop > 1) {
        const float d = hi - lo;           /* (y2 - y1)                  */
        const float m = d * span;          /*          * (H - 1)         */
        step = m / (float)(crop - 1);      /*                  / (ch-1)  */
    }
    for (int k = 0; k < crop; ++k) {
        float c;
        if (crop > 1) {
            const float base = lo * span;      /* y1 * (H-1)   */
            const float off = (float)k * step; /* y * scale    */
            c = base + off;
        } else {
            /* the reference writes 0.5 * (y1 + y2) * (H - 1): the float sum is
             * promoted to double by the 0.5 literal, multiplied by the int
             * extent in double and only then stored to a float. */
            const float s = lo + hi;
            const double dc = 0.5 * (double)s * (double)(extent - 1);
            c = (float)dc;
        }
        t[k].coord = c;
        if (c < 0.0f || c > span) {
            t[k].valid = 0;
            t[k].i0 = t[k].i1 = 0;
            t[k].frac = 0.0f;
        } else {
            t[k].valid = 1;
            t[k].i0 = (int)floorf(c);
            t[k].i1 = (int)ceilf(c);
            t[k].frac = c - (float)t[k].i0;
        }
    }
}

/* Exposes the bin assignment itself (tests compare it bit-for-bit with the HIP
 * library's fi_crop_and_resize_taps).  Outputs are [N, crop] per axis. */
int orc_crop_taps(const float *boxes, int num_boxes, int H, int W, int crop_h,
                  int crop_w, int32_t *y_valid, int32_t *y0, int32_t *y1,
                  float *y_frac, int32_t *x_valid, int32_t *x0, int32_t *x1,
                  float *x_frac)
{
    if (crop_h <= 0 || crop_w <= 0) return ORC_BAD_ARG;
    orc_tap *ty = (orc_tap *)malloc(sizeof(orc_tap) * (size_t)crop_h);
    orc_tap *tx = (orc_tap *)malloc(sizeof(orc_tap) * (size_t)crop_w);
    for (int n = 0; n < num_boxes; ++n) {
        const float *b = boxes + 4 * (size_t)n;
        orc_axis_taps(b[0], b[2], H, crop_h, ty);
        orc_axis_taps(b[1], b[3], W, crop_w, tx);
        for (int k = 0; k < crop_h; ++k) {
            y_valid[n * crop_h + k] = ty[k].valid;
            y0[n * crop_h + k] = ty[k].i0;
            y1[n * crop_h + k] = ty[k].i1;
            y_frac[n * crop_h + k] = ty[k].frac;
        }
        for (int k = 0; k < crop_w; ++k) {
            x_valid[n * crop_w + k] = tx[k].valid;
            x0[n * crop_w + k] = tx[k].i0;
            x1[n * crop_w + k] = tx[k].i1;
            x_frac[n * crop_w + k] = tx[k].frac;
        }
    }
    free(ty);
    free(tx);
    return ORC_OK;
}

/* crop_and_resize forward.
 * Reference: CropAndResizePerBox, lib/roi_align/src/crop_and_resize.c:6-112
 * (OpenMP over boxes :30) and crop_and_resize_forward :115-154 (output is
 * zero-filled first, :130-131).  image [B,C,H,W], boxes [N,4]=(y1,x1,y2,x2)
 * normalised, box_ind [N], crops [N,C,crop_h,crop_w].
 * The reference aborts the process on an out-of-range box index (:39-42); the
 * oracle returns ORC_BAD_BOX_INDEX instead and leaves that box zero. */
int orc_crop_and_resize_forward(const float *image, int B, int C, int H, int W,
                                const float *boxes, const int32_t *box_ind,
                                int num_boxes, int crop_h, int crop_w,
                                float extrapolation_value, float *crops)
{
    if (crop_h <= 0 || crop_w <= 0 || C <= 0) return ORC_BAD_ARG;
    const size_t plane = (size_t)H * (size_t)W;
    const size_t bins = (size_t)crop_h * (size_t)crop_w;
    memset(crops, 0, sizeof(float) * (size_t)num_boxes * (size_t)C * bins);
    int status = ORC_OK;

#pragma omp parallel for schedule(static)
    for (int n = 0; n < num_boxes; ++n) {
        const int img = box_ind[n];
        if (img < 0 || img >= B) {
#pragma omp critical
            status = ORC_BAD_BOX_INDEX;
            continue;
        }
        orc_tap ty[64], tx[64];
        orc_tap *py = ty, *px = tx;
        if (crop_h > 64) py = (orc_tap *)malloc(sizeof(orc_tap) * (size_t)crop_h);
        if (crop_w > 64) px = (orc_tap *)malloc(sizeof(orc_tap) * (size_t)crop_w);
        const float *b = boxes + 4 * (size_t)n;
        orc_axis_taps(b[0], b[2], H, crop_h, py);
        orc_axis_taps(b[1], b[3], W, crop_w, px);

        float *out_box = crops + (size_t)n * (size_t)C * bins;
        const float *img_base = image + (size_t)img * (size_t)C * plane;
        for (int c = 0; c < C; ++c) {
            const float *src = img_base + (size_t)c * plane;
            float *dst = out_box + (size_t)c * bins;
            for (int y = 0; y < crop_h; ++y) {
                for (int x = 0; x < crop_w; ++x) {
                    float v;
                    if (!py[y].valid || !px[x].valid) {
                        v = extrapolation_value; /* :58-69, :82-88 */
                    } else {
                        const float tl = src[(size_t)py[y].i0 * W + px[x].i0];
                        const float tr = src[(size_t)py[y].i0 * W + px[x].i1];
                        const float bl = src[(size_t)py[y].i1 * W + px[x].i0];
                        const float br = src[(size_t)py[y].i1 * W + px[x].i1];
                        /* :102-106: top, bottom, then the vertical blend */
                        const float dt = tr - tl;
                        const float top = tl + dt * px[x].frac;
                        const float db = br - bl;
                        const float bot = bl + db * px[x].frac;
                        const float dv = bot - top;
                        v = top + dv * py[y].frac;
                    }
                    dst[(size_t)y * crop_w + x] = v;
                }
            }
        }
        if (py != ty) free(py);
        if (px != tx) free(px);
    }
    return status;
}

/* crop_and_resize backward (gradient w.r.t. the image only).
 * Reference: crop_and_resize_backward, lib/roi_align/src/crop_and_resize.c:157-252:
 * grads_image zero-filled (:184), boxes visited serially in index order (:190,
 * no OpenMP on purpose), per box y, x, then depth innermost (:236), and the four
 * weighted adds in the order TL, TR, BL, BR (:241-247).  The accumulation order is
 * reproduced exactly so that this oracle is deterministic; the HIP path uses
 * hardware fp32 atomics and is compared within a tolerance. */
int orc_crop_and_resize_backward(const float *grads, const float *boxes,
                                 const int32_t *box_ind, int num_boxes, int B,
                                 int C, int H, int W, int crop_h, int crop_w,
                                 float *grads_image)
{
    if (crop_h <= 0 || crop_w <= 0 || C <= 0) return ORC_BAD_ARG;
    const size_t plane = (size_t)H * (size_t)W;
    const size_t bins = (size_t)crop_h * (size_t)crop_w;
    memset(grads_image, 0, sizeof(float) * (size_t)B * (size_t)C * plane);
    orc_tap *ty = (orc_tap *)malloc(sizeof(orc_tap) * (size_t)crop_h);
    orc_tap *tx = (orc_tap *)malloc(sizeof(orc_tap) * (size_t)crop_w);
    int status = ORC_OK;
    for (int n = 0; n < num_boxes; ++n) {
        const int img = box_ind[n];
        if (img < 0 || img >= B) {
            status = ORC_BAD_BOX_INDEX;
            continue;
        }
        const float *b = boxes + 4 * (size_t)n;
        orc_axis_taps(b[0], b[2], H, crop_h, ty);
        orc_axis_taps(b[1], b[3], W, crop_w, tx);
        const float *g_box = grads + (size_t)n * (size_t)C * bins;
        float *dst_img = grads_image + (size_t)img * (size_t)C * plane;
        for (int y = 0; y < crop_h; ++y) {
            if (!ty[y].valid) continue;
            const float wy1 = ty[y].frac;
            const float wy0 = 1.0f - wy1;
            for (int x = 0; x < crop_w; ++x) {
                if (!tx[x].valid) continue;
                const float wx1 = tx[x].frac;
                const float wx0 = 1.0f - wx1;
                for (int c = 0; c < C; ++c) {
                    float *dst = dst_img + (size_t)c * plane;
                    const float g = g_box[(size_t)c * bins + (size_t)y * crop_w + x];
                    const float gtop = wy0 * g;
                    dst[(size_t)ty[y].i0 * W + tx[x].i0] += wx0 * gtop;
                    dst[(size_t)ty[y].i0 * W + tx[x].i1] += wx1 * gtop;
                    const float gbot = wy1 * g;
                    dst[(size_t)ty[y].i1 * W + tx[x].i0] += wx0 * gbot;
                    dst[(size_t)ty[y].i1 * W + tx[x].i1] += wx1 * gbot;
                }
            }
        }
    }
    free(ty);
    free(tx);
    return status;
}

/* --------------------------------------------------------------------------
 * RoIPool.  The reference's CPU file (lib/roi_pooling/src/roi_pooling.c) is not
 * a usable specification (batch 1 only, NHWC raw-storage read of a permuted view,
 * -1 fill, no argmax/backward -- SURVEY Q4); the CUDA kernel defines the
 * operator, so this follows lib/roi_pooling/src/roi_pooling_kernel.cu.
 * ------------------------------------------------------------------------ */
typedef struct {
    int img;
    int start_w, start_h, end_w, end_h;
    int roi_w, roi_h;
    float bin_h, bin_w;
} orc_roi;

/* roi_pooling_kernel.cu:44-54 (forward) == :148-151, :167-171 (backward). */
static orc_roi orc_roi_decode(const float *r, float scale, int ph, int pw)
{
    orc_roi o;
    o.img = (int)r[0];
    o.start_w = (int)roundf(r[1] * scale);
    o.start_h = (int)roundf(r[2] * scale);
    o.end_w = (int)roundf(r[3] * scale);
    o.end_h = (int)roundf(r[4] * scale);
    o.roi_w = (int)fmaxf((float)(o.end_w - o.start_w + 1), 1.0f);
    o.roi_h = (int)fmaxf((float)(o.end_h - o.start_h + 1), 1.0f);
    o.bin_h = (float)o.roi_h / (float)ph;
    o.bin_w = (float)o.roi_w / (float)pw;
    return o;
}

static int orc_clampi(int v, int lo, int hi)
{
    /* fminf(fmaxf(v, lo), hi) on small ints is exact */
    return (int)fminf(fmaxf((float)v, (float)lo), (float)hi);
}

/* Forward: ROIPoolForward, roi_pooling_kernel.cu:24-93.
 * features [B,C,H,W]; rois [N,5] = (batch, x1, y1, x2, y2) in pixels;
 * out / argmax [N,C,ph,pw]; argmax holds the flat index into the whole NCHW
 * tensor (:85) or -1 for an empty bin (:73-75). */
int orc_roi_pool_forward(const float *features, int B, int C, int H, int W,
                         const float *rois, int num_rois, int ph, int pw,
                         float scale, float *out, int32_t *argmax)
{
    (void)B;
    if (ph <= 0 || pw <= 0) return ORC_BAD_ARG;
    for (int n = 0; n < num_rois; ++n) {
        const orc_roi r = orc_roi_decode(rois + 5 * (size_t)n, scale, ph, pw);
        for (int c = 0; c < C; ++c) {
            const long plane_off = ((long)r.img * C + c) * (long)H * W;
            for (int p = 0; p < ph; ++p) {
                int hs = (int)floorf((float)p * r.bin_h);
                int he = (int)ceilf((float)(p + 1) * r.bin_h);
                hs = orc_clampi(hs + r.start_h, 0, H);
                he = orc_clampi(he + r.start_h, 0, H);
                for (int q = 0; q < pw; ++q) {
                    int ws = (int)floorf((float)q * r.bin_w);
                    int we = (int)ceilf((float)(q + 1) * r.bin_w);
                    ws = orc_clampi(ws + r.start_w, 0, W);
                    we = orc_clampi(we + r.start_w, 0, W);
                    const int empty = (he <= hs) || (we <= ws);
                    float best = empty ? 0.0f : -FLT_MAX;
                    long best_i = -1;
                    for (int h = hs; h < he; ++h)
                        for (int w = ws; w < we; ++w) {
                            const long i = plane_off + (long)h * W + w;
                            if (features[i] > best) { /* strict: first max wins */
                                best = features[i];
                                best_i = i;
                            }
                        }
                    const size_t o = (((size_t)n * C + c) * ph + p) * pw + q;
                    out[o] = best;
                    if (argmax) argmax[o] = (int32_t)best_i;
                }
            }
        }
    }
    return ORC_OK;
}

/* Backward: ROIPoolBackward, roi_pooling_kernel.cu:128-203.  For each input
 * element, RoIs are scanned in index order and the feasible pooled cells in
 * (ph, pw) order (:182-190); a RoI contributes only if it is on the same image
 * (:147), contains the pixel after rounding (:155-160) and the pooled cell's
 * argmax equals this element's flat index.  Note the `in_roi` test uses the raw
 * rounded corners, so a malformed RoI (end < start) that the forward pass forced
 * to 1x1 receives no gradient -- reproduced here. */
int orc_roi_pool_backward(const float *top_grad, const int32_t *argmax,
                          const float *rois, int num_rois, int B, int C, int H,
                          int W, int ph, int pw, float scale, float *bottom_grad)
{
    const long total = (long)B * C * H * W;
    orc_roi *dec = (orc_roi *)malloc(sizeof(orc_roi) * (size_t)(num_rois > 0 ? num_rois : 1));
    for (int n = 0; n < num_rois; ++n)
        dec[n] = orc_roi_decode(rois + 5 * (size_t)n, scale, ph, pw);
    for (long index = 0; index < total; ++index) {
        long t = index;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H); t /= H;
        const int c = (int)(t % C); t /= C;
        const int img = (int)t;
        float g = 0.0f;
        for (int n = 0; n < num_rois; ++n) {
            const orc_roi *r = &dec[n];
            if (r->img != img) continue;
            if (!(w >= r->start_w && w <= r->end_w && h >= r->start_h && h <= r->end_h))
                continue;
            int p0 = (int)floorf((float)(h - r->start_h) / r->bin_h);
            int p1 = (int)ceilf((float)(h - r->start_h + 1) / r->bin_h);
            int q0 = (int)floorf((float)(w - r->start_w) / r->bin_w);
            int q1 = (int)ceilf((float)(w - r->start_w + 1) / r->bin_w);
            p0 = orc_clampi(p0, 0, ph);
            p1 = orc_clampi(p1, 0, ph);
            q0 = orc_clampi(q0, 0, pw);
            q1 = orc_clampi(q1, 0, pw);
            const size_t off = (size_t)n * C * ph * pw;
            for (int p = p0; p < p1; ++p)
                for (int q = q0; q < q1; ++q) {
                    const size_t o = off + ((size_t)c * ph + p) * pw + q;
                    if ((long)argmax[o] == index) g += top_grad[o];
                }
        }
        bottom_grad[index] = g;
    }
    free(dec);
    return ORC_OK;
}

/* --------------------------------------------------------------------------
 * Greedy NMS.
 * Reference (CPU spec): cpu_nms, lib/nms/src/nms.c:4-69 -- boxes visited through
 * `order`, a kept box i suppresses every later unsuppressed j with
 *   inter / (area_i + area_j - inter) >= thresh           (:55-59)
 * where w = max(0, xx2 - xx1 + 1), h likewise (+1 pixel convention, :53-54) and
 * the areas are supplied by the caller (lib/nms/pth_nms.py:13).
 * strict != 0 selects the CUDA variant's `>` (lib/nms/src/cuda/nms_kernel.cu:63);
 * the arithmetic is otherwise the same (devIoU :16-24).
 * boxes [N, dim] (dim >= 4) with columns (c0, c1, c2, c3): the IoU is symmetric
 * under swapping the x and y roles, which is why the reference can feed
 * (y1,x1,y2,x2) rows to a routine written for (x1,y1,x2,y2).
 * keep_out [N] int64, returns the number kept.
 * ------------------------------------------------------------------------ */
long orc_nms(const float *boxes, long num, long dim, const int64_t *order,
             const float *areas, float thresh, int strict, int64_t *keep_out)
{
    unsigned char *dead = (unsigned char *)calloc((size_t)(num > 0 ? num : 1), 1);
    long kept = 0;
    for (long oi = 0; oi < num; ++oi) {
        const long i = (long)order[oi];
        if (dead[i]) continue;
        keep_out[kept++] = i;
        const float a0 = boxes[i * dim + 0], a1 = boxes[i * dim + 1];
        const float a2 = boxes[i * dim + 2], a3 = boxes[i * dim + 3];
        const float area_i = areas[i];
        for (long oj = oi + 1; oj < num; ++oj) {
            const long j = (long)order[oj];
            if (dead[j]) continue;
            const float l = fmaxf(a0, boxes[j * dim + 0]);
            const float t = fmaxf(a1, boxes[j * dim + 1]);
            const float r = fminf(a2, boxes[j * dim + 2]);
            const float bt = fminf(a3, boxes[j * dim + 3]);
            const float dw = r - l;
            const float dh = bt - t;
            const float w = fmaxf(0.0f, dw + 1.0f);
            const float h = fmaxf(0.0f, dh + 1.0f);
            const float inter = w * h;
            const float s = area_i + areas[j];
            const float uni = s - inter;
            const float iou = inter / uni;
            const int hit = strict ? (iou > thresh) : (iou >= thresh);
            if (hit) dead[j] = 1;
        }
    }
    free(dead);
    return kept;
}

/* --------------------------------------------------------------------------
 * Sinkhorn term of the OT intertwiner loss.
 * Reference: OptTrans._sinkhorn_iterate, lib/OT_module.py:104-135.
 *   cosine (:110-113): x <- x / (||x||_2 + 1e-20) per row, same for y,
 *                      C = 1 - x y^T
 *   l2     (:106-109): C_ij = ||x_i - y_j||_2
 *   K = exp(-eps_inv * C) (:116; the module stores 1/epsilon, :13)
 *   b = u = 1/S; L times { a = u / (K b + 1e-20); b = u / (K^T a + 1e-20) } (:118-122)
 *   P = a * K * b^T (:128), loss = <P, C> (:134).
 * x, y are [S, D] row-major.  Stored intermediates are fp32 as in the reference
 * (torch fp32 tensors); reductions use a double accumulator so the result does
 * not depend on a summation order the reference (torch.mm) does not define.
 * plan_out (optional) receives P [S,S].
 * ------------------------------------------------------------------------ */
#define ORC_OT_EPS 1e-20f

float orc_sinkhorn(const float *x, const float *y, int S, int D, float eps_inv,
                   int L, int l2_cost, float *plan_out)
{
    const size_t SS = (size_t)S * (size_t)S;
    float *C = (float *)malloc(sizeof(float) * SS);
    float *K = (float *)malloc(sizeof(float) * SS);
    float *xn = (float *)malloc(sizeof(float) * (size_t)S * D);
    float *yn = (float *)malloc(sizeof(float) * (size_t)S * D);
    float *a = (float *)malloc(sizeof(float) * (size_t)S);
    float *b = (float *)malloc(sizeof(float) * (size_t)S);

    if (!l2_cost) {
        for (int i = 0; i < S; ++i) {
            double sx = 0.0, sy = 0.0;
            for (int d = 0; d < D; ++d) {
                sx += (double)x[(size_t)i * D + d] * (double)x[(size_t)i * D + d];
                sy += (double)y[(size_t)i * D + d] * (double)y[(size_t)i * D + d];
            }
            const float nx = (float)sqrt(sx) + ORC_OT_EPS;
            const float ny = (float)sqrt(sy) + ORC_OT_EPS;
            for (int d = 0; d < D; ++d) {
                xn[(size_t)i * D + d] = x[(size_t)i * D + d] / nx;
                yn[(size_t)i * D + d] = y[(size_t)i * D + d] / ny;
            }
        }
        for (int i = 0; i < S; ++i)
            for (int j = 0; j < S; ++j) {
                double dot = 0.0;
                for (int d = 0; d < D; ++d)
                    dot += (double)xn[(size_t)i * D + d] * (double)yn[(size_t)j * D + d];
                C[(size_t)i * S + j] = 1.0f - (float)dot;
            }
    } else {
        for (int i = 0; i < S; ++i)
            for (int j = 0; j < S; ++j) {
                double ss = 0.0;
                for (int d = 0; d < D; ++d) {
                    const float df = x[(size_t)i * D + d] - y[(size_t)j * D + d];
                    ss += (double)df * (double)df;
                }
                C[(size_t)i * S + j] = (float)sqrt(ss);
            }
    }
    for (size_t k = 0; k < SS; ++k) K[k] = expf(-eps_inv * C[k]);

    const float u = 1.0f / (float)S;
    for (int j = 0; j < S; ++j) b[j] = u;
    for (int i = 0; i < S; ++i) a[i] = u; /* defined even when L == 0 is never used */
    for (int it = 0; it < L; ++it) {
        for (int i = 0; i < S; ++i) {
            double acc = 0.0;
            for (int j = 0; j < S; ++j) acc += (double)K[(size_t)i * S + j] * (double)b[j];
            a[i] = u / ((float)acc + ORC_OT_EPS);
        }
        for (int j = 0; j < S; ++j) {
            double acc = 0.0;
            for (int i = 0; i < S; ++i) acc += (double)K[(size_t)i * S + j] * (double)a[i];
            b[j] = u / ((float)acc + ORC_OT_EPS);
        }
    }
    double loss = 0.0;
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) {
            const float ak = a[i] * K[(size_t)i * S + j];
            const float p = ak * b[j];
            if (plan_out) plan_out[(size_t)i * S + j] = p;
            loss += (double)p * (double)C[(size_t)i * S + j];
        }
    free(C); free(K); free(xn); free(yn); free(a); free(b);
    return (float)loss;
}

/* --------------------------------------------------------------------------
 * Per-class feature mean used by the intertwiner statistics.
 * Reference: Dev._assign_feat2cls, lib/sub_module.py:664-684: for every
 * foreground class c present in gt, feat[:, c] = mean of the rows with that
 * class, cnt[0, c] = their number; background (0) is skipped, absent classes stay 0.
 * features [N, F], gt [N] (class ids as int32), feat [F, num_classes], cnt [num_classes].
 * ------------------------------------------------------------------------ */
int orc_class_mean(const float *features, const int32_t *gt, int N, int F,
                   int num_classes, float *feat, float *cnt)
{
    double *acc = (double *)calloc((size_t)F * (size_t)num_classes, sizeof(double));
    memset(cnt, 0, sizeof(float) * (size_t)num_classes);
    for (int n = 0; n < N; ++n) {
        const int c = gt[n];
        if (c <= 0 || c >= num_classes) continue;
        cnt[c] += 1.0f;
        for (int f = 0; f < F; ++f)
            acc[(size_t)f * num_classes + c] += (double)features[(size_t)n * F + f];
    }
    for (int f = 0; f < F; ++f)
        for (int c = 0; c < num_classes; ++c)
            feat[(size_t)f * num_classes + c] =
                cnt[c] > 0.0f ? (float)(acc[(size_t)f * num_classes + c] / (double)cnt[c]) : 0.0f;
    free(acc);
    return ORC_OK;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
