// glue.hip -- small data-movement kernels of the training step that replace chains of framework launches.
#include <stdint.h>

#include <algorithm>

#include "fi_common.h"

namespace {

// dx[p][h][w] = c[h & 1][w & 1][p][h >> 1][w >> 1] (0 where that residue class has no tensor) (+ add[p][h][w]):
// the data gradient of a stride-2 convolution, assembled from the stride-1 correlations of its residue classes
// (conv._strided_dgrad) in ONE pass that writes every element once -- instead of a zero fill plus one strided
// scatter-copy per class (and a separate add of the second gradient that flows into the same tensor).
template <bool VEC>
__global__ __launch_bounds__(256) void stride2_interleave_kernel(
    const float *__restrict__ c00, const float *__restrict__ c01, const float *__restrict__ c10,
    const float *__restrict__ c11, const float *__restrict__ add, const float *__restrict__ gate,
    float *__restrict__ dx, long planes, int H, int W)
{
    const int qw0 = (W + 1) >> 1, qw1 = W >> 1;
    const int qh0 = (H + 1) >> 1, qh1 = H >> 1;
    if (VEC) {
        // W % 4 == 0: a thread writes 4 consecutive columns of one row (16-byte store), reading 2 + 2 values
        const int wq = W >> 2;
        const long total = planes * H * wq;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int j = (int)(i % wq);
            const long ph = i / wq;
            const int h = (int)(ph % H);
            const long p = ph / H;
            const int qh = h >> 1;
            const float *e = (h & 1) ? c10 : c00;
            const float *o = (h & 1) ? c11 : c01;
            const int qhn = (h & 1) ? qh1 : qh0;
            const long base = (p * qhn + qh) * (long)qw0 + 2 * j;
            float2 ev = make_float2(0.f, 0.f), ov = make_float2(0.f, 0.f);
            if (e) ev = *reinterpret_cast<const float2 *>(e + base);
            if (o) ov = *reinterpret_cast<const float2 *>(o + base);
            float4 r = make_float4(ev.x, ov.x, ev.y, ov.y);
            const long off = (p * H + h) * (long)W + 4 * j;
            if (add) {
                const float4 a = *reinterpret_cast<const float4 *>(add + off);
                r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w;
            }
            if (gate) {          // dx * (gate > 0): the layer's input is a ReLU output that nothing else reads (conv.Gate)
                const float4 g4 = *reinterpret_cast<const float4 *>(gate + off);
                r.x = g4.x > 0.0f ? r.x : 0.0f; r.y = g4.y > 0.0f ? r.y : 0.0f;
                r.z = g4.z > 0.0f ? r.z : 0.0f; r.w = g4.w > 0.0f ? r.w : 0.0f;
            }
            *reinterpret_cast<float4 *>(dx + off) = r;
        }
    } else {
        const long total = planes * H * W;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int w = (int)(i % W);
            const long ph = i / W;
            const int h = (int)(ph % H);
            const long p = ph / H;
            const float *src = (h & 1) ? ((w & 1) ? c11 : c10) : ((w & 1) ? c01 : c00);
            const int qhn = (h & 1) ? qh1 : qh0;
            const int qwn = (w & 1) ? qw1 : qw0;
            float v = src ? src[(p * qhn + (h >> 1)) * (long)qwn + (w >> 1)] : 0.0f;
            if (add) v += add[i];
            if (gate) v = gate[i] > 0.0f ? v : 0.0f;
            dx[i] = v;
        }
    }
}

// scale[c] = gamma[c] * rsqrt(var[c] + eps), shift[c] = beta[c] - mean[c] * scale[c] (+ conv_bias[c] * scale[c]) for EVERY
// (convolution, eval-mode BatchNorm) pair of the model in one launch: the per-step refresh of the folded BatchNorms
// (conv.refresh_bn_folds) was five multi-tensor framework launches over ~100 tensors each -- 2.5 ms of HOST time at
// the step boundary, where the device has nothing queued.  One workgroup per (pair, block of 256 channels).
__global__ __launch_bounds__(256) void bn_fold_kernel(const FiBnFoldDesc *__restrict__ descs, int n)
{
    const int pair = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (pair >= n) return;
    const FiBnFoldDesc d = descs[pair];
    if (c >= d.channels) return;
    const float *gamma = static_cast<const float *>(d.gamma), *beta = static_cast<const float *>(d.beta);
    const float *mean = static_cast<const float *>(d.mean), *var = static_cast<const float *>(d.var);
    const float *cb = static_cast<const float *>(d.conv_bias);
    const float inv = rsqrtf(var[c] + d.eps);
    const float sc = gamma[c] * inv;
    float sh = beta[c] - mean[c] * sc;
    if (cb) sh = sh + cb[c] * sc;
    static_cast<float *>(d.scale)[c] = sc;
    static_cast<float *>(d.shift)[c] = sh;
}

// out[p][h][w] = ((dy[p][2h][2w] + dy[p][2h][2w+1]) + dy[p][2h+1][2w]) + dy[p][2h+1][2w+1]: the backward of a x2
// nearest-neighbour upsampling (the FPN's top-down path, lib/sub_module.py:172-200), in the summation order of the
// framework's kernel (rows, then columns of the 2 x 2 source window).  A thread makes 2 adjacent outputs from two
// 16-byte loads.
__global__ __launch_bounds__(256) void sum2x2_kernel(const float *__restrict__ dy, float *__restrict__ out, long planes,
                                                     int H, int W)
{
    const int wp = W >> 1;                                  // output pairs per row (W even)
    const long total = planes * H * wp;
    const int W2 = 2 * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int j = (int)(i % wp);
        const long ph = i / wp;                             // p * H + h
        const float *__restrict__ r0 = dy + (ph * 2) * W2 + 4 * j;
        const float4 a = *reinterpret_cast<const float4 *>(r0);
        const float4 b = *reinterpret_cast<const float4 *>(r0 + W2);
        float2 o;
        o.x = ((a.x + a.y) + b.x) + b.y;
        o.y = ((a.z + a.w) + b.z) + b.w;
        *reinterpret_cast<float2 *>(out + ph * W + 2 * j) = o;
    }
}

// ---- 3 x 3 / stride 2 max-pooling of the stem (ceil_mode, no padding: windows clipped at the border) -----------------
// The framework's pair of kernels stores an int64 index per output (forward: 0.23 ms) and reads it back in a gather
// over all windows of every input element (backward: 0.45 ms for 268 MB of gradient); the input is at hand in backward
// (it is the stem's ReLU output, kept for the stem's own backward), so the arg-max is recomputed instead, with the
// framework's rule: scan rows then columns, take v if (v > best || isnan(v)) -- the FIRST maximum.
__device__ __forceinline__ bool pool_better(float v, float best) { return v > best || v != v; }

// a thread makes 2 adjacent outputs from 3 rows x (one 16-byte + one 4-byte) loads
__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, long planes,
                                                             int H, int W, int OH, int OW)
{
    const int op = (OW + 1) >> 1;
    const long total = planes * OH * op;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int j = (int)(i % op);
        const long po = i / op;
        const int oh = (int)(po % OH);
        const long p = po / OH;
        const float *__restrict__ base = x + (p * H + 2 * oh) * W + 4 * j;
        float b0 = 0.0f, b1 = 0.0f;
        bool first = true;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (2 * oh + r >= H) break;
            const float4 v = *reinterpret_cast<const float4 *>(base + (long)r * W);      // W % 4 == 0
            const bool more = 4 * j + 4 < W;
            const float e = more ? base[(long)r * W + 4] : 0.0f;
            if (first) { b0 = v.x; b1 = v.z; first = false; } else { b0 = pool_better(v.x, b0) ? v.x : b0; b1 = pool_better(v.z, b1) ? v.z : b1; }
            b0 = pool_better(v.y, b0) ? v.y : b0;
            b0 = pool_better(v.z, b0) ? v.z : b0;
            b1 = pool_better(v.w, b1) ? v.w : b1;
            if (more) b1 = pool_better(e, b1) ? e : b1;
        }
        float *__restrict__ o = y + po * OW + 2 * j;
        o[0] = b0;
        if (2 * j + 1 < OW) o[1] = b1;
    }
}

// a thread owns rows (2m, 2m+1) x columns 4t .. 4t+3 of the input gradient: the windows that can select one of its
// elements are (m-1, m) x (2t-1, 2t, 2t+1); contributions are added in the framework's order (window rows, then columns).
// positive_only: the result is multiplied by (x > 0) -- x is a ReLU output whose mask its producer then skips (Gate).
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                             float *__restrict__ dx, long planes, int H, int W, int OH,
                                                             int OW, int positive_only)
{
    const int tq = W >> 2, mh = (H + 1) >> 1;
    const long total = planes * mh * tq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i % tq);
        const long pm = i / tq;
        const int m = (int)(pm % mh);
        const long p = pm / mh;
        const float *__restrict__ xp = x + p * H * W;
        const float *__restrict__ dyp = dy + p * OH * OW;
        // x rows 2m-2 .. 2m+2, columns 4t-2 .. 4t+4 (absent ones are never read below)
        float xv[5][7];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int h = 2 * m - 2 + r;
#pragma unroll
            for (int c = 0; c < 7; ++c) xv[r][c] = 0.0f;
            if (h < 0 || h >= H) continue;
            const float *__restrict__ row = xp + (long)h * W + 4 * t;
            const float4 v = *reinterpret_cast<const float4 *>(row);
            xv[r][2] = v.x; xv[r][3] = v.y; xv[r][4] = v.z; xv[r][5] = v.w;
            if (t > 0) {
                const float2 l = *reinterpret_cast<const float2 *>(row - 2);
                xv[r][0] = l.x; xv[r][1] = l.y;
            }
            if (4 * t + 4 < W) xv[r][6] = row[4];
        }
        float acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = 0.0f;
#pragma unroll
        for (int wr = 0; wr < 2; ++wr) {                      // window row oh = m - 1 + wr: local x rows 2*wr .. 2*wr+2
            const int oh = m - 1 + wr;
            if (oh < 0 || oh >= OH) continue;
#pragma unroll
            for (int wc = 0; wc < 3; ++wc) {                  // window column ow = 2t - 1 + wc: local x columns 2*wc .. 2*wc+2
                const int ow = 2 * t - 1 + wc;
                if (ow < 0 || ow >= OW) continue;
                float best = 0.0f;
                int br = -1, bc = -1;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    if (2 * oh + r >= H) continue;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        if (2 * ow + c >= W) continue;
                        const float v = xv[2 * wr + r][2 * wc + c];
                        if (br < 0 || pool_better(v, best)) { best = v; br = 2 * wr + r; bc = 2 * wc + c; }
                    }
                }
                const float g = dyp[(long)oh * OW + ow];
                // own elements: local rows 2, 3 (input rows 2m, 2m+1), local columns 2 .. 5
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] += (br == 2 + a && bc == 2 + b) ? g : 0.0f;
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int h = 2 * m + a;
            if (h >= H) continue;
            float4 o = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
            if (positive_only) {
                o.x = xv[2 + a][2] > 0.0f ? o.x : 0.0f; o.y = xv[2 + a][3] > 0.0f ? o.y : 0.0f;
                o.z = xv[2 + a][4] > 0.0f ? o.z : 0.0f; o.w = xv[2 + a][5] > 0.0f ? o.w : 0.0f;
            }
            *reinterpret_cast<float4 *>(dx + (p * H + h) * W + 4 * t) = o;
        }
    }
}

// Row gather / scatter-add of a [rows][row_len] tensor by an index vector whose entries are DISTINCT (a permutation
// prefix): dst[i] = src[index[i]]  /  dst[index[i]] += src[i].  One workgroup column per row, 16-byte accesses; the
// framework's index kernels handle arbitrary (repeating) indices and run at a third of the copy rate here.
__global__ __launch_bounds__(256) void rows_gather_kernel(const float *__restrict__ src, const long *__restrict__ index,
                                                          float *__restrict__ dst, long row_len4)
{
    const long row = blockIdx.y;
    const float4 *__restrict__ s4 = reinterpret_cast<const float4 *>(src) + index[row] * row_len4;
    float4 *__restrict__ d4 = reinterpret_cast<float4 *>(dst) + row * row_len4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < row_len4; i += (long)gridDim.x * 256) d4[i] = s4[i];
}

__global__ __launch_bounds__(256) void rows_scatter_add_kernel(const float *__restrict__ src, const long *__restrict__ index,
                                                               float *__restrict__ dst, long row_len4)
{
    const long row = blockIdx.y;
    const float4 *__restrict__ s4 = reinterpret_cast<const float4 *>(src) + row * row_len4;
    float4 *__restrict__ d4 = reinterpret_cast<float4 *>(dst) + index[row] * row_len4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < row_len4; i += (long)gridDim.x * 256) {
        float4 a = d4[i];
        const float4 b = s4[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        d4[i] = a;
    }
}

// dst[r] = (r < n_front ? front[r] : 0) + (src_row[r] >= 0 ? src[src_row[r]] : 0): the gradient of a [rows][row_len] tensor
// with two readers -- one took its first n_front rows as a view, the other a set of distinct rows -- written once.
__global__ __launch_bounds__(256) void rows_combine_kernel(const float *__restrict__ front, long n_front,
                                                           const float *__restrict__ src, const long *__restrict__ src_row,
                                                           float *__restrict__ dst, long row_len4)
{
    const long row = blockIdx.y;
    const long sr = src_row[row];
    const float4 *__restrict__ f4 = reinterpret_cast<const float4 *>(front) + row * row_len4;
    const float4 *__restrict__ s4 = reinterpret_cast<const float4 *>(src) + (sr >= 0 ? sr : 0) * row_len4;
    float4 *__restrict__ d4 = reinterpret_cast<float4 *>(dst) + row * row_len4;
    const bool hf = row < n_front, hs = sr >= 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < row_len4; i += (long)gridDim.x * 256) {
        float4 a = hf ? f4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (hs) {
            const float4 b = s4[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        d4[i] = a;
    }
}

// Backward of a 1x1 convolution whose output gradient has ONE non-zero channel per image row: dy[n][cls[n]][p] = d[n][p],
// zero elsewhere (the mask head's conv5 under the mask loss, lib/layers.py:905-934: only the target class's mask of a
// RoI enters the loss).  The dense kernels would multiply 80 of 81 channels of zeros:
//     dx[n][c][p]  = W[cls[n]][c] * d[n][p]      (* (x > 0) when gated: x is the ReLU output of the deconv)
//     dW[k][c] += sum over rows n of class k, pixels p of d[n][p] * x[n][c][p]        db[k] += sum of d[n][p]
// Phase 1, one workgroup per (row n, block of 64 channels): a wavefront owns a channel at a time, lanes run over pixels
// (the plane's HW floats are contiguous); the per-row sums go to a workspace v[n][c], t[n].  Rows whose d is all zero
// (RoIs that are not positives: three quarters of them) write zeros without reading x.  Phase 2, one workgroup per
// (class, 64 channels): the rows of the class are summed in row order -- no atomics (thousands of rows share one
// class: atomics on dW[k][c] serialise), deterministic.
__global__ __launch_bounds__(256) void class_row_conv1x1_bwd_kernel(const float *__restrict__ d, const float *__restrict__ x,
                                                                    const float *__restrict__ w, const long *__restrict__ cls,
                                                                    float *__restrict__ dx, float *__restrict__ v,
                                                                    float *__restrict__ t, float *__restrict__ live, int C,
                                                                    int HW, int gated)
{
    extern __shared__ float s_d[];                   // d[n][0 .. HW)
    __shared__ int s_any;
    const long n = blockIdx.x;
    const int c0 = blockIdx.y * 64;
    const long k = cls[n];
    const float *__restrict__ dn = d + n * HW;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    bool any = false;
    for (int p = threadIdx.x; p < HW; p += 256) {
        const float q = dn[p];
        s_d[p] = q;
        any |= q != 0.0f;
    }
    if (any) s_any = 1;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cend = min(c0 + 64, C);
    if (!s_any) {                                    // workgroup-uniform: a row that is not a positive RoI
        if (blockIdx.y == 0 && threadIdx.x == 0) live[n] = 0.0f;
        if (dx) {
            float *__restrict__ op = dx + (n * C + c0) * (long)HW;
            const long len = (long)(cend - c0) * HW;
            if ((len & 3) == 0 && ((uintptr_t)op & 15) == 0) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                for (long i = threadIdx.x; i < (len >> 2); i += 256) reinterpret_cast<float4 *>(op)[i] = z;
            } else {
                for (long i = threadIdx.x; i < len; i += 256) op[i] = 0.0f;
            }
        }
        return;
    }
    if (blockIdx.y == 0 && wave == 0) {
        float s = 0.0f;
        for (int p = lane; p < HW; p += 64) s += s_d[p];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) {
            t[n] = s;
            live[n] = 1.0f;
        }
    }
    // two channels per trip (c and c + 4): their loads are in flight together
    for (int c = c0 + wave; c < cend; c += 8) {
        const bool two = c + 4 < cend;
        const int cb = two ? c + 4 : c;
        const float wa = w[k * C + c], wb = w[k * C + cb];
        const float *__restrict__ xa = x + (n * C + c) * (long)HW;
        const float *__restrict__ xb = x + (n * C + cb) * (long)HW;
        float acca = 0.0f, accb = 0.0f;
        for (int p = lane; p < HW; p += 64) {
            const float va = xa[p], vb = xb[p], dv = s_d[p];
            acca += dv * va;
            accb += dv * vb;
            if (dx) {
                dx[(n * C + c) * (long)HW + p] = (!gated || va > 0.0f) ? wa * dv : 0.0f;
                if (two) dx[(n * C + cb) * (long)HW + p] = (!gated || vb > 0.0f) ? wb * dv : 0.0f;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            acca += __shfl_xor(acca, off, 64);
            accb += __shfl_xor(accb, off, 64);
        }
        if (lane == 0) {
            v[n * C + c] = acca;
            if (two) v[n * C + cb] = accb;
        }
    }
}

// Phase 2: (block of 64 channels, chunk of 256 rows) workgroups add their live rows into a [K][64] LDS table (ds_add_f32:
// rows of one class meet inside the workgroup, not in memory) and flush the classes they saw with one global atomic per
// (class, channel) -- 32 per address for 8192 rows, whatever the class distribution.
__global__ __launch_bounds__(256) void class_row_reduce_kernel(const float *__restrict__ v, const float *__restrict__ t,
                                                               const float *__restrict__ live, const long *__restrict__ cls,
                                                               float *__restrict__ dw, float *__restrict__ db, long N, int C,
                                                               int K)
{
    extern __shared__ float s_acc[];                 // [K][64], then [K] for the bias sums
    float *s_t = s_acc + (size_t)K * 64;
    for (int i = threadIdx.x; i < K * 65; i += 256) s_acc[i] = 0.0f;
    __syncthreads();
    const int cc = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cc;
    const long n0 = (long)blockIdx.y * 256;
    for (long n = n0 + slice; n < min(N, n0 + 256); n += 4) {
        if (live[n] == 0.0f) continue;               // wavefront-uniform (a wavefront = one slice)
        const long k = cls[n];
        if (c < C) atomicAdd(&s_acc[k * 64 + cc], v[n * C + c]);
        if (blockIdx.x == 0 && cc == 0) atomicAdd(&s_t[k], t[n]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * 64; i += 256) {
        const int k = i >> 6, ch = blockIdx.x * 64 + (i & 63);
        const float a = s_acc[i];
        if (a != 0.0f && ch < C && dw) atomicAdd(dw + (size_t)k * C + ch, a);
    }
    if (blockIdx.x == 0 && db)
        for (int k = threadIdx.x; k < K; k += 256)
            if (s_t[k] != 0.0f) atomicAdd(db + k, s_t[k]);
}

// ---- 3 x 3 patches of pyramid maps at selected anchors (the RPN's training path, sub_module.RPN.forward_rows) -------
// Row r = (image[r], anchor[r]) of the level-major anchor list: level l holds H_l * W_l * per_loc anchors starting at
// start[l]; anchor a of level l sits on pixel (a - start[l]) / per_loc.  out[r][tap][c] = map_l[image][c][h + dy][w + dx]
// (zero outside the map and for rows with image[r] < 0); backward adds d[r][tap][c] into the level's gradient map.
struct PatchLevels {
    const float *map[8];
    float *grad[8];
    int H[8], W[8];
    long start[8];
    int levels, per_loc;
};

__device__ __forceinline__ bool patch_locate(const PatchLevels &lv, long img, long a, int &l, int &h, int &w)
{
    if (img < 0 || a < 0) return false;
    l = 0;
    while (l + 1 < lv.levels && a >= lv.start[l + 1]) ++l;
    const long pix = (a - lv.start[l]) / lv.per_loc;
    h = (int)(pix / lv.W[l]);
    w = (int)(pix - (long)h * lv.W[l]);
    return h < lv.H[l];
}

// one workgroup per (row, tap): lanes over channels (stride H*W in the NCHW map, contiguous in the output)
__global__ __launch_bounds__(256) void patch_rows_fwd_kernel(PatchLevels lv, const long *__restrict__ image,
                                                             const long *__restrict__ anchor, float *__restrict__ out, int C)
{
    const long r = blockIdx.x;
    const int tap = blockIdx.y;
    int l, h, w;
    float *__restrict__ o = out + (r * 9 + tap) * (long)C;
    bool ok = patch_locate(lv, image[r], anchor[r], l, h, w);
    if (ok) {
        h += tap / 3 - 1;
        w += tap % 3 - 1;
        ok = h >= 0 && h < lv.H[l] && w >= 0 && w < lv.W[l];
    }
    if (!ok) {
        for (int c = threadIdx.x; c < C; c += 256) o[c] = 0.0f;
        return;
    }
    const long HW = (long)lv.H[l] * lv.W[l];
    const float *__restrict__ src = lv.map[l] + image[r] * C * HW + (long)h * lv.W[l] + w;
    for (int c = threadIdx.x; c < C; c += 256) o[c] = src[c * HW];
}

__global__ __launch_bounds__(256) void patch_rows_bwd_kernel(PatchLevels lv, const long *__restrict__ image,
                                                             const long *__restrict__ anchor, const float *__restrict__ d,
                                                             int C)
{
    const long r = blockIdx.x;
    const int tap = blockIdx.y;
    int l, h, w;
    if (!patch_locate(lv, image[r], anchor[r], l, h, w)) return;
    h += tap / 3 - 1;
    w += tap % 3 - 1;
    if (h < 0 || h >= lv.H[l] || w < 0 || w >= lv.W[l]) return;
    const long HW = (long)lv.H[l] * lv.W[l];
    float *__restrict__ dst = lv.grad[l] + image[r] * C * HW + (long)h * lv.W[l] + w;
    const float *__restrict__ src = d + (r * 9 + tap) * (long)C;
    for (int c = threadIdx.x; c < C; c += 256) atomicAdd(dst + c * HW, src[c]);
}

__global__ __launch_bounds__(256) void relu_mask_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                        float *__restrict__ out, long n4, long n)
{
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 d = reinterpret_cast<const float4 *>(dy)[i];
        const float4 v = reinterpret_cast<const float4 *>(y)[i];
        float4 o;
        o.x = v.x > 0.0f ? d.x : 0.0f; o.y = v.y > 0.0f ? d.y : 0.0f;
        o.z = v.z > 0.0f ? d.z : 0.0f; o.w = v.w > 0.0f ? d.w : 0.0f;
        reinterpret_cast<float4 *>(out)[i] = o;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = y[i] > 0.0f ? dy[i] : 0.0f;
}

// Backward of a fully connected layer + eval-mode BatchNorm + ReLU, first pass (fi_rows_mask_scale): rows [M][N], the
// channel is the contiguous index.  g = dy * (y > 0) -> g_out, g * scale[n] -> gs_out (the operand of the data gradient),
// colsum[n] += sum_m g (d beta; fi_bn_fold_grad turns it and the weight gradient of g into d gamma / d bias / dW).
// A workgroup takes 256 columns x rows_per_block rows: 64 column quads x 4 row lanes.
__global__ __launch_bounds__(256) void rows_mask_scale_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                              const float *__restrict__ scale, float *__restrict__ g_out,
                                                              float *__restrict__ gs_out, float *__restrict__ colsum, int M,
                                                              int N, int ld_out, int relu, int rows_per_block)
{
    __shared__ float4 s_part[4][64];
    const int tid = threadIdx.x, lane = tid & 63, rl = tid >> 6;
    const int c0 = (blockIdx.x * 64 + lane) * 4;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c0 < N) {
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
        if (scale) sc = *reinterpret_cast<const float4 *>(scale + c0);
        for (int r = r0 + rl; r < r1; r += 4) {
            float4 d = *reinterpret_cast<const float4 *>(dy + (size_t)r * N + c0);
            if (relu) {
                const float4 v = *reinterpret_cast<const float4 *>(y + (size_t)r * N + c0);
                d.x = v.x > 0.0f ? d.x : 0.0f; d.y = v.y > 0.0f ? d.y : 0.0f;
                d.z = v.z > 0.0f ? d.z : 0.0f; d.w = v.w > 0.0f ? d.w : 0.0f;
            }
            acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
            if (g_out) *reinterpret_cast<float4 *>(g_out + (size_t)r * ld_out + c0) = d;
            if (gs_out)
                *reinterpret_cast<float4 *>(gs_out + (size_t)r * ld_out + c0) =
                    make_float4(d.x * sc.x, d.y * sc.y, d.z * sc.z, d.w * sc.w);
        }
    }
    if (!colsum) return;
    s_part[rl][lane] = acc;
    __syncthreads();
    if (rl == 0 && c0 < N) {
        float4 t = s_part[0][lane];
        for (int k = 1; k < 4; ++k) {
            const float4 v = s_part[k][lane];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        atomicAdd(colsum + c0, t.x); atomicAdd(colsum + c0 + 1, t.y);
        atomicAdd(colsum + c0 + 2, t.z); atomicAdd(colsum + c0 + 3, t.w);
    }
}

// y = act(y * scale[n] + bias[n]) in place, rows [M][N]: the epilogue of the 16-bit GEMM (fp32 atomics over its K split,
// so it has no reduction pass to fold this into)
__global__ __launch_bounds__(256) void rows_affine_act_kernel(float *__restrict__ y, const float *__restrict__ scale,
                                                              const float *__restrict__ bias, long total4, int N, int relu)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        float4 a = *reinterpret_cast<float4 *>(y + 4 * i);
        const int c = (int)((4 * i) % N);
        if (scale) {
            const float4 m = *reinterpret_cast<const float4 *>(scale + c);
            a.x *= m.x; a.y *= m.y; a.z *= m.z; a.w *= m.w;
        }
        if (bias) {
            const float4 b = *reinterpret_cast<const float4 *>(bias + c);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (relu) {
            a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
        }
        *reinterpret_cast<float4 *>(y + 4 * i) = a;
    }
}

// one workgroup per output channel: <W[co], dW'[co]> while dW'[co] is scaled in place (see fi_capi.h)
__global__ __launch_bounds__(256) void bn_fold_grad_kernel(float *__restrict__ dw, const float *__restrict__ w,
                                                           const float *__restrict__ s, const float *__restrict__ scale,
                                                           const float *__restrict__ mean, const float *__restrict__ var,
                                                           float eps, const float *__restrict__ conv_bias,
                                                           float *__restrict__ dgamma, float *__restrict__ dbias, int Cin,
                                                           int taps, int vec, int same_order, int dw_tap_major)
{
    __shared__ float part[4];
    const int co = blockIdx.x, K = Cin * taps;
    float *__restrict__ drow = dw + (size_t)co * K;
    const float *__restrict__ wrow = w + (size_t)co * K;
    const float sc = scale[co];
    float dot = 0.0f;
    if (vec) {
        for (int k = threadIdx.x * 4; k < K; k += 1024) {
            float4 d = *reinterpret_cast<float4 *>(drow + k);
            const float4 v = *reinterpret_cast<const float4 *>(wrow + k);
            dot += (d.x * v.x + d.y * v.y) + (d.z * v.z + d.w * v.w);
            d.x *= sc; d.y *= sc; d.z *= sc; d.w *= sc;
            *reinterpret_cast<float4 *>(drow + k) = d;
        }
    } else {
        for (int k = threadIdx.x; k < K; k += 256) {
            int kw = k;
            if (!same_order) {
                // k indexes dW'; the same (ci, tap) element of W sits at the other order's index
                const int ci = dw_tap_major ? k % Cin : k / taps, tap = dw_tap_major ? k / Cin : k % taps;
                kw = dw_tap_major ? ci * taps + tap : tap * Cin + ci;
            }
            const float d = drow[k];
            dot += d * wrow[kw];
            drow[k] = d * sc;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) dot += __shfl_xor(dot, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float total = (part[0] + part[1]) + (part[2] + part[3]);
        const float sv = s[co];
        const float inv = rsqrtf(var[co] + eps);
        const float cb = conv_bias ? conv_bias[co] : 0.0f;
        // atomics: a layer applied twice per step has its other use accumulate here from another stream
        if (dgamma) atomicAdd(dgamma + co, inv * (total + (cb - mean[co]) * sv));
        if (dbias) atomicAdd(dbias + co, sc * sv);
    }
}

// the same for up to FI_WGRAD_BATCH_MAX layers of one geometry (behind fi_conv2d_weight_grad_batch): operand pointers by
// value in the kernel arguments, blockIdx.y = layer
struct FoldGradBatch {
    float *dw[FI_WGRAD_BATCH_MAX];
    const float *w[FI_WGRAD_BATCH_MAX];
    const float *s[FI_WGRAD_BATCH_MAX];
    const float *scale[FI_WGRAD_BATCH_MAX];
    const float *mean[FI_WGRAD_BATCH_MAX];
    const float *var[FI_WGRAD_BATCH_MAX];
    const float *conv_bias[FI_WGRAD_BATCH_MAX];
    float *dgamma[FI_WGRAD_BATCH_MAX];
    float *dbias[FI_WGRAD_BATCH_MAX];
};

__global__ __launch_bounds__(256) void bn_fold_grad_batch_kernel(FoldGradBatch b, float eps, int Cin, int taps, int vec,
                                                                 int same_order, int dw_tap_major)
{
    __shared__ float part[4];
    const int co = blockIdx.x, K = Cin * taps, l = blockIdx.y;
    float *__restrict__ drow = b.dw[l] + (size_t)co * K;
    const float *__restrict__ wrow = b.w[l] + (size_t)co * K;
    const float sc = b.scale[l][co];
    float dot = 0.0f;
    if (vec) {
        for (int k = threadIdx.x * 4; k < K; k += 1024) {
            float4 d = *reinterpret_cast<float4 *>(drow + k);
            const float4 v = *reinterpret_cast<const float4 *>(wrow + k);
            dot += (d.x * v.x + d.y * v.y) + (d.z * v.z + d.w * v.w);
            d.x *= sc; d.y *= sc; d.z *= sc; d.w *= sc;
            *reinterpret_cast<float4 *>(drow + k) = d;
        }
    } else {
        for (int k = threadIdx.x; k < K; k += 256) {
            int kw = k;
            if (!same_order) {
                const int ci = dw_tap_major ? k % Cin : k / taps, tap = dw_tap_major ? k / Cin : k % taps;
                kw = dw_tap_major ? ci * taps + tap : tap * Cin + ci;
            }
            const float d = drow[k];
            dot += d * wrow[kw];
            drow[k] = d * sc;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) dot += __shfl_xor(dot, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float total = (part[0] + part[1]) + (part[2] + part[3]);
        const float sv = b.s[l][co];
        const float inv = rsqrtf(b.var[l][co] + eps);
        const float cb = b.conv_bias[l] ? b.conv_bias[l][co] : 0.0f;
        if (b.dgamma[l]) atomicAdd(b.dgamma[l] + co, inv * (total + (cb - b.mean[l][co]) * sv));
        if (b.dbias[l]) atomicAdd(b.dbias[l] + co, sc * sv);
    }
}

}  // namespace

extern "C" {

size_t fi_class_row_conv1x1_workspace_bytes(long N, int C) { return (size_t)(N > 0 ? N : 0) * (size_t)(C + 2) * sizeof(float); }

int fi_class_row_conv1x1_backward(const float *d, const float *x, const float *weight, const int64_t *cls, float *dx,
                                  float *dweight, float *dbias, long N, int C, int HW, int num_classes, int gated,
                                  float *workspace, fi_stream_t stream)
{
    FI_REQUIRE(N >= 0 && C >= 1 && HW >= 1 && HW <= 16384 && num_classes >= 1, "bad sizes (HW <= 16384)");
    if (N == 0) return FI_OK;
    FI_REQUIRE(d && x && weight && cls && workspace, "null pointer");
    FI_REQUIRE(N <= 2147483647L, "too many rows");
    if (num_classes > 240) {
        fi::set_error("fi_class_row_conv1x1_backward keeps a [num_classes][64] table in LDS: num_classes <= 240 (got %d)", num_classes);
        return FI_ERR_UNSUPPORTED;
    }
    float *v = workspace, *t = workspace + (size_t)N * C, *live = t + N;
    hipLaunchKernelGGL(class_row_conv1x1_bwd_kernel, dim3((unsigned)N, (unsigned)((C + 63) / 64)), dim3(256),
                       (size_t)HW * sizeof(float), (hipStream_t)stream, d, x, weight, reinterpret_cast<const long *>(cls), dx,
                       v, t, live, C, HW, gated);
    FI_HIP_CHECK(hipGetLastError());
    if (dweight || dbias) {
        hipLaunchKernelGGL(class_row_reduce_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)((N + 255) / 256)), dim3(256),
                           (size_t)num_classes * 65 * sizeof(float), (hipStream_t)stream, v, t, live,
                           reinterpret_cast<const long *>(cls), dweight, dbias, N, C, num_classes);
        FI_HIP_CHECK(hipGetLastError());
    }
    return FI_OK;
}

static int patch_levels(PatchLevels &lv, const void *const *maps, void *const *grads, const int *heights, const int *widths,
                        int levels, int per_loc)
{
    FI_REQUIRE(levels >= 1 && levels <= 8 && per_loc >= 1, "1..8 pyramid levels");
    long start = 0;
    for (int l = 0; l < levels; ++l) {
        FI_REQUIRE(heights[l] >= 1 && widths[l] >= 1, "bad map size");
        lv.map[l] = maps ? static_cast<const float *>(maps[l]) : nullptr;
        lv.grad[l] = grads ? static_cast<float *>(grads[l]) : nullptr;
        FI_REQUIRE((maps && lv.map[l]) || (grads && lv.grad[l]), "null level map");
        lv.H[l] = heights[l];
        lv.W[l] = widths[l];
        lv.start[l] = start;
        start += (long)heights[l] * widths[l] * per_loc;
    }
    lv.levels = levels;
    lv.per_loc = per_loc;
    return FI_OK;
}

int fi_pyramid_patch_rows_forward(const void *const *maps, const int *heights, const int *widths, int levels,
                                  int anchors_per_location, const int64_t *image, const int64_t *anchor, long rows,
                                  int channels, float *out, fi_stream_t stream)
{
    FI_REQUIRE(rows >= 0 && channels >= 1, "bad sizes");
    if (rows == 0) return FI_OK;
    FI_REQUIRE(maps && heights && widths && image && anchor && out, "null pointer");
    PatchLevels lv = {};
    const int rc = patch_levels(lv, maps, nullptr, heights, widths, levels, anchors_per_location);
    if (rc != FI_OK) return rc;
    hipLaunchKernelGGL(patch_rows_fwd_kernel, dim3((unsigned)rows, 9), dim3(256), 0, (hipStream_t)stream, lv,
                       reinterpret_cast<const long *>(image), reinterpret_cast<const long *>(anchor), out, channels);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_pyramid_patch_rows_backward(const float *d, void *const *grads, const int *heights, const int *widths, int levels,
                                   int anchors_per_location, const int64_t *image, const int64_t *anchor, long rows,
                                   int channels, fi_stream_t stream)
{
    FI_REQUIRE(rows >= 0 && channels >= 1, "bad sizes");
    if (rows == 0) return FI_OK;
    FI_REQUIRE(d && grads && heights && widths && image && anchor, "null pointer");
    PatchLevels lv = {};
    const int rc = patch_levels(lv, nullptr, grads, heights, widths, levels, anchors_per_location);
    if (rc != FI_OK) return rc;
    hipLaunchKernelGGL(patch_rows_bwd_kernel, dim3((unsigned)rows, 9), dim3(256), 0, (hipStream_t)stream, lv,
                       reinterpret_cast<const long *>(image), reinterpret_cast<const long *>(anchor), d, channels);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_rows_gather(const float *src, const int64_t *index, float *dst, long n_index, long row_len, fi_stream_t stream)
{
    FI_REQUIRE(n_index >= 0 && row_len >= 4 && row_len % 4 == 0, "row_len must be a positive multiple of 4");
    if (n_index == 0) return FI_OK;
    FI_REQUIRE(src && index && dst, "null pointer");
    FI_REQUIRE((uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0 && n_index <= 65535, "16-byte aligned tensors, <= 65535 rows");
    const long per = std::min<long>(std::max<long>((row_len / 4 + 1023) / 1024, 1), 64);
    hipLaunchKernelGGL(rows_gather_kernel, dim3((unsigned)per, (unsigned)n_index), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<const long *>(index), dst, row_len / 4);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_rows_scatter_add(const float *src, const int64_t *index, float *dst, long n_index, long row_len, fi_stream_t stream)
{
    FI_REQUIRE(n_index >= 0 && row_len >= 4 && row_len % 4 == 0, "row_len must be a positive multiple of 4");
    if (n_index == 0) return FI_OK;
    FI_REQUIRE(src && index && dst, "null pointer");
    FI_REQUIRE((uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0 && n_index <= 65535, "16-byte aligned tensors, <= 65535 rows");
    const long per = std::min<long>(std::max<long>((row_len / 4 + 1023) / 1024, 1), 64);
    hipLaunchKernelGGL(rows_scatter_add_kernel, dim3((unsigned)per, (unsigned)n_index), dim3(256), 0, (hipStream_t)stream,
                       src, reinterpret_cast<const long *>(index), dst, row_len / 4);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_rows_combine(const float *front, long n_front, const float *src, const int64_t *src_row, float *dst, long rows,
                    long row_len, fi_stream_t stream)
{
    FI_REQUIRE(rows >= 0 && n_front >= 0 && n_front <= rows && row_len >= 4 && row_len % 4 == 0,
               "row_len must be a positive multiple of 4, n_front <= rows");
    if (rows == 0) return FI_OK;
    FI_REQUIRE(src_row && dst && (front || n_front == 0), "null pointer");
    FI_REQUIRE((uintptr_t)dst % 16 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)front % 16 == 0 && rows <= 65535,
               "16-byte aligned tensors, <= 65535 rows");
    const long per = std::min<long>(std::max<long>((row_len / 4 + 1023) / 1024, 1), 64);
    hipLaunchKernelGGL(rows_combine_kernel, dim3((unsigned)per, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, front,
                       n_front, src, reinterpret_cast<const long *>(src_row), dst, row_len / 4);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_maxpool3x3s2_forward(const float *x, float *y, long planes, int height, int width, fi_stream_t stream)
{
    FI_REQUIRE(planes >= 0 && height >= 3 && width >= 4 && width % 4 == 0, "x is [planes][height][width], width % 4 == 0");
    if (planes == 0) return FI_OK;
    FI_REQUIRE(x && y, "null pointer");
    FI_REQUIRE((uintptr_t)x % 16 == 0, "x must be 16-byte aligned");
    const int OH = (height - 2) / 2 + 1, OW = (width - 2) / 2 + 1;          // ceil((n - 3) / 2) + 1, last window starts inside
    const long total = planes * OH * ((OW + 1) / 2);
    const long blocks = std::min<long>((total + 255) / 256, 256L * 32);
    hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, planes, height,
                       width, OH, OW);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_maxpool3x3s2_backward(const float *dy, const float *x, float *dx, long planes, int height, int width,
                             int positive_only, fi_stream_t stream)
{
    FI_REQUIRE(planes >= 0 && height >= 3 && width >= 4 && width % 4 == 0, "x is [planes][height][width], width % 4 == 0");
    if (planes == 0) return FI_OK;
    FI_REQUIRE(dy && x && dx, "null pointer");
    FI_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)dx % 16 == 0, "x and dx must be 16-byte aligned");
    const int OH = (height - 2) / 2 + 1, OW = (width - 2) / 2 + 1;
    const long total = planes * ((height + 1) / 2) * (width / 4);
    const long blocks = std::min<long>((total + 255) / 256, 256L * 64);
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, x, dx, planes,
                       height, width, OH, OW, positive_only);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_sum2x2(const float *dy, float *out, long planes, int height, int width, fi_stream_t stream)
{
    FI_REQUIRE(planes >= 0 && height >= 1 && width >= 2 && width % 2 == 0, "out is [planes][height][width], width even");
    if (planes == 0) return FI_OK;
    FI_REQUIRE(dy && out, "null pointer");
    FI_REQUIRE((uintptr_t)dy % 16 == 0 && (uintptr_t)out % 8 == 0, "dy must be 16-byte, out 8-byte aligned");
    const long total = planes * height * (width / 2);
    const long blocks = std::min<long>((total + 255) / 256, 256L * 32);
    hipLaunchKernelGGL(sum2x2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, out, planes, height,
                       width);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_relu_mask(const float *dy, const float *y, float *out, long n, fi_stream_t stream)
{
    FI_REQUIRE(n >= 0, "bad size");
    if (n == 0) return FI_OK;
    FI_REQUIRE(dy && y && out, "null pointer");
    const bool vec = ((uintptr_t)dy % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)out % 16 == 0);
    const long n4 = vec ? n / 4 : 0;
    const long blocks = std::min<long>(std::max<long>((std::max<long>(n4, 1) + 511) / 512, 1), 256L * 16);
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, y, out, n4, n);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_rows_mask_scale(const float *dy, const float *y, const float *scale, float *g, float *gs, float *colsum, int M,
                       int N, int ld_out, int relu, int flags, fi_stream_t stream)
{
    FI_REQUIRE(M >= 0 && N >= 4 && N % 4 == 0 && ld_out >= N && ld_out % 4 == 0, "N and ld_out must be multiples of 4");
    if (M == 0) return FI_OK;
    FI_REQUIRE(dy && (y || !relu) && (g || gs || colsum), "null pointer");
    FI_REQUIRE((((uintptr_t)dy | (uintptr_t)y | (uintptr_t)scale | (uintptr_t)g | (uintptr_t)gs) & 15) == 0,
               "fi_rows_mask_scale needs 16-byte aligned operands");
    hipStream_t st = (hipStream_t)stream;
    if (colsum && !(flags & FI_OUTPUTS_ZEROED)) FI_HIP_CHECK(hipMemsetAsync(colsum, 0, sizeof(float) * (size_t)N, st));
    const int panels = (N + 255) / 256;
    int rpb = 32;                                  // enough workgroups to fill the chip, few atomics per column
    while (rpb < 256 && (long)panels * ((M + rpb - 1) / rpb) > 2048) rpb *= 2;
    hipLaunchKernelGGL(rows_mask_scale_kernel, dim3((unsigned)panels, (unsigned)((M + rpb - 1) / rpb)), dim3(256), 0, st, dy, y,
                       scale, g, gs, colsum, M, N, ld_out, relu ? 1 : 0, rpb);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_rows_affine_act(float *y, const float *scale, const float *bias, int M, int N, int relu, fi_stream_t stream)
{
    FI_REQUIRE(M >= 0 && N >= 4 && N % 4 == 0, "N must be a multiple of 4");
    if (M == 0) return FI_OK;
    FI_REQUIRE(y != nullptr, "null pointer");
    FI_REQUIRE((((uintptr_t)y | (uintptr_t)scale | (uintptr_t)bias) & 15) == 0, "fi_rows_affine_act needs 16-byte aligned operands");
    const long total4 = (long)M * N / 4;
    const long blocks = (total4 + 255) / 256;
    hipLaunchKernelGGL(rows_affine_act_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       (hipStream_t)stream, y, scale, bias, total4, N, relu ? 1 : 0);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_bn_fold_grad(float *dw, const float *w, const float *s, const float *scale, const float *mean,
                    const float *var, float eps, const float *conv_bias, float *dgamma, float *dbias, int Cout,
                    int Cin, int taps, int dw_tap_major, int w_tap_major, fi_stream_t stream)
{
    FI_REQUIRE(Cout > 0 && Cin > 0 && taps > 0, "bad sizes");
    FI_REQUIRE(dw && w && s && scale && mean && var, "null pointer");
    const int same_order = (taps == 1 || (dw_tap_major != 0) == (w_tap_major != 0)) ? 1 : 0;
    // 16-byte accesses need aligned rows; otherwise the scalar loop (with the index map when the orders differ)
    const int vec = (same_order && ((long)Cin * taps) % 4 == 0 && (uintptr_t)dw % 16 == 0 && (uintptr_t)w % 16 == 0) ? 1 : 0;
    hipLaunchKernelGGL(bn_fold_grad_kernel, dim3((unsigned)Cout), dim3(256), 0, (hipStream_t)stream, dw, w, s, scale, mean,
                       var, eps, conv_bias, dgamma, dbias, Cin, taps, vec, same_order, dw_tap_major);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_bn_fold_grad_batch(float *const *dw, const float *const *w, const float *const *s, const float *const *scale,
                          const float *const *mean, const float *const *var, float eps, const float *const *conv_bias,
                          float *const *dgamma, float *const *dbias, int n, int Cout, int Cin, int taps, int dw_tap_major,
                          int w_tap_major, fi_stream_t stream)
{
    FI_REQUIRE(n >= 1 && Cout > 0 && Cin > 0 && taps > 0, "bad sizes");
    FI_REQUIRE(dw && w && s && scale && mean && var, "null pointer table");
    const int same_order = (taps == 1 || (dw_tap_major != 0) == (w_tap_major != 0)) ? 1 : 0;
    for (int i0 = 0; i0 < n; i0 += FI_WGRAD_BATCH_MAX) {
        const int m = n - i0 < FI_WGRAD_BATCH_MAX ? n - i0 : FI_WGRAD_BATCH_MAX;
        FoldGradBatch b;
        int vec = (same_order && ((long)Cin * taps) % 4 == 0) ? 1 : 0;
        for (int i = 0; i < FI_WGRAD_BATCH_MAX; ++i) {
            const int j = i0 + (i < m ? i : 0);
            FI_REQUIRE(dw[j] && w[j] && s[j] && scale[j] && mean[j] && var[j], "null pointer in the batch");
            b.dw[i] = dw[j]; b.w[i] = w[j]; b.s[i] = s[j]; b.scale[i] = scale[j]; b.mean[i] = mean[j]; b.var[i] = var[j];
            b.conv_bias[i] = conv_bias ? conv_bias[j] : nullptr;
            b.dgamma[i] = dgamma ? dgamma[j] : nullptr;
            b.dbias[i] = dbias ? dbias[j] : nullptr;
            if ((uintptr_t)dw[j] % 16 != 0 || (uintptr_t)w[j] % 16 != 0) vec = 0;
        }
        hipLaunchKernelGGL(bn_fold_grad_batch_kernel, dim3((unsigned)Cout, (unsigned)m), dim3(256), 0, (hipStream_t)stream, b, eps,
                           Cin, taps, vec, same_order, dw_tap_major);
    }
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_bn_fold_batch(const FiBnFoldDesc *descs_dev, int n, int max_channels, fi_stream_t stream)
{
    FI_REQUIRE(n >= 0 && max_channels >= 0, "bad sizes");
    if (n == 0 || max_channels == 0) return FI_OK;
    FI_REQUIRE(descs_dev != nullptr, "null descriptor table");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)n, (unsigned)((max_channels + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, descs_dev, n);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}


int fi_stride2_interleave(const float *c00, const float *c01, const float *c10, const float *c11,
                          const float *add, float *dx, long planes, int height, int width, fi_stream_t stream)
{
    return fi_stride2_interleave_gated(c00, c01, c10, c11, add, nullptr, dx, planes, height, width, stream);
}

int fi_stride2_interleave_gated(const float *c00, const float *c01, const float *c10, const float *c11,
                                const float *add, const float *gate, float *dx, long planes, int height, int width,
                                fi_stream_t stream)
{
    FI_REQUIRE(dx && planes >= 0 && height >= 1 && width >= 1, "bad interleave arguments");
    if (planes == 0) return FI_OK;
    const uintptr_t all = (uintptr_t)c00 | (uintptr_t)c01 | (uintptr_t)c10 | (uintptr_t)c11;
    const bool vec = width % 4 == 0 && all % 8 == 0 && (uintptr_t)dx % 16 == 0 && (uintptr_t)add % 16 == 0 &&
                     (uintptr_t)gate % 16 == 0;
    const long work = vec ? planes * height * (width / 4) : planes * height * width;
    const long blocks = (work + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 65536 ? (blocks < 1 ? 1 : blocks) : 65536);
    fi::ProfScope prof(FI_K_STRIDE2_INTERLEAVE, (hipStream_t)stream);
    if (vec)
        hipLaunchKernelGGL(stride2_interleave_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, c00, c01,
                           c10, c11, add, gate, dx, planes, height, width);
    else
        hipLaunchKernelGGL(stride2_interleave_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, c00, c01,
                           c10, c11, add, gate, dx, planes, height, width);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

}  // extern "C"
