// roi_pool.hip -- Caffe-style max RoIPool forward (+argmax) and backward, gfx950.
//
// Specification: ROIPoolForward / ROIPoolBackward,
// lib/roi_pooling/src/roi_pooling_kernel.cu:24-93 / :128-203 (the CUDA kernel is the
// only usable definition of the operator -- SURVEY Q4).  Oracle: orc_roi_pool_*.
//
// Forward (roi_pool_fwd_kernel): one workgroup per (RoI, channel chunk); the RoI's bin
// boundaries (hstart/hend per pooled row, wstart/wend per pooled column, after rounding,
// offsetting and clipping) are computed once into LDS.  Each wavefront then works on
// (channel, window-column) pairs: a lane owns ONE column of the RoI window, so that a
// wavefront load instruction fetches a contiguous row segment of the plane (the first
// version gave every lane its own bin window: adjacent lanes on different rows, one cache
// line per lane and step).  Narrow windows pack 2..8 channels into the 64 lanes.
//   phase 1  per pooled row p: every lane walks its column down rows [hstart_p, hend_p)
//            keeping (max, first row of the max) -> wavefront-private LDS colv/colh[p][lane]
//   phase 2  per bin (p, q): one lane scans the <= bin-width columns of LDS in w order,
//            v > best || (v == best && row < best_row)  ==  the reference's strict `>` in
//            (h, w) scan order (first maximum wins, roi_pooling_kernel.cu:75-86)
// Windows wider than 64 columns are processed in 64-column slabs with the running bin
// state kept in LDS.  Pooled sizes above 28x28 use the simple per-bin kernel.
// Measured (512 RoIs x 256 ch on a 2x256x256x256 map): 7x7 1269 -> 400 us, 14x14 -> 640 us.
// What bounds THIS form: one row segment (<= 256 B) per vector-memory instruction; whole-map
// windows read at 6.6 TB/s even when every byte is an L2 hit (scripts/roipool_probe.py), i.e.
// the CU's load-instruction rate, not memory.  roi_pool_fwd_v2_kernel (below) therefore gives a
// lane 16 bytes of ONE pooled row: 7x7 181 us, 14x14 408 us, whole-map windows at 14-17 TB/s from
// L2.  This form remains for pooled heights 17..28 and maps narrower than 4.
// Backward: the reference gathers -- every INPUT element loops over ALL RoIs
// (O(B*C*H*W*N), 67 M threads x N at P2).  Here each pooled cell scatters its
// gradient to its argmax with a hardware fp32 atomic, after re-checking the
// reference's feasibility conditions (same image, pixel inside the rounded RoI,
// pooled cell inside the pixel's candidate window) so that the set of summed terms
// is identical; only the summation order differs.
#include <float.h>

#include "fi_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxPool = 64;

struct Roi {
    int img;
    int start_w, start_h, end_w, end_h;
    float bin_h, bin_w;
};

// roi_pooling_kernel.cu:44-54.  roundf = round half away from zero, as CUDA round().
__device__ __forceinline__ Roi decode_roi(const float *__restrict__ r, float scale, int ph, int pw)
{
    Roi o;
    o.img = (int)r[0];
    o.start_w = (int)roundf(r[1] * scale);
    o.start_h = (int)roundf(r[2] * scale);
    o.end_w = (int)roundf(r[3] * scale);
    o.end_h = (int)roundf(r[4] * scale);
    const int roi_w = max(o.end_w - o.start_w + 1, 1);
    const int roi_h = max(o.end_h - o.start_h + 1, 1);
    o.bin_h = (float)roi_h / (float)ph;
    o.bin_w = (float)roi_w / (float)pw;
    return o;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

__global__ __launch_bounds__(kThreads) void roi_pool_fwd_simple_kernel(
    const float *__restrict__ features, const float *__restrict__ rois, int num_rois, int batch,
    int channels, int height, int width, int ph, int pw, float scale, int chan_per_block,
    int chunks, float *__restrict__ output, int *__restrict__ argmax)
{
    __shared__ int s_h0[kMaxPool], s_h1[kMaxPool], s_w0[kMaxPool], s_w1[kMaxPool];
    const int tid = threadIdx.x;
    const int n = blockIdx.x / chunks;
    const int chunk = blockIdx.x - n * chunks;
    const int c_begin = chunk * chan_per_block;
    const int c_count = min(chan_per_block, channels - c_begin);
    const int bins = ph * pw;
    const int total = c_count * bins;

    const Roi r = decode_roi(rois + 5 * (size_t)n, scale, ph, pw);
    if (tid < ph) {
        const int hs = (int)floorf((float)tid * r.bin_h);
        const int he = (int)ceilf((float)(tid + 1) * r.bin_h);
        s_h0[tid] = clampi(hs + r.start_h, 0, height);
        s_h1[tid] = clampi(he + r.start_h, 0, height);
    }
    if (tid >= 64 && tid < 64 + pw) {
        const int q = tid - 64;
        const int ws = (int)floorf((float)q * r.bin_w);
        const int we = (int)ceilf((float)(q + 1) * r.bin_w);
        s_w0[q] = clampi(ws + r.start_w, 0, width);
        s_w1[q] = clampi(we + r.start_w, 0, width);
    }
    __syncthreads();

    const size_t o_base = ((size_t)n * channels + c_begin) * bins;
    const bool img_ok = r.img >= 0 && r.img < batch;
    for (int idx = tid; idx < total; idx += kThreads) {
        const int c = idx / bins;
        const int bin = idx - c * bins;
        const int p = bin / pw;
        const int q = bin - p * pw;
        const int hs = s_h0[p], he = s_h1[p], ws = s_w0[q], we = s_w1[q];
        const bool empty = (he <= hs) || (we <= ws) || !img_ok;
        float best = empty ? 0.0f : -FLT_MAX;
        int best_i = -1;
        if (!empty) {
            const int plane_off = ((r.img * channels) + c_begin + c) * height * width;
            const float *__restrict__ src = features + plane_off;
            for (int h = hs; h < he; ++h) {
                const float *__restrict__ row = src + h * width;
                for (int w = ws; w < we; ++w) {
                    const float v = row[w];
                    if (v > best) {  // strict: the first maximum in (h, w) order wins
                        best = v;
                        best_i = plane_off + h * width + w;
                    }
                }
            }
        }
        output[o_base + idx] = best;
        if (argmax) argmax[o_base + idx] = best_i;
    }
}

__device__ __forceinline__ void wave_lds_sync()
{
    // LDS traffic of one wavefront is issued and completed in order; this only stops the compiler from
    // moving the reads of other lanes' data above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kWaves = kThreads / 64;

// dynamic LDS per wavefront: colv[ph][64] floats, colh[ph][64] ints, then (wide windows only) the running
// bin state binv[bins], binh[bins], binw[bins]
__host__ __device__ inline size_t roi_pool_wave_lds(int ph, int pw) { return (size_t)ph * 64 * 8 + (size_t)ph * pw * 12; }

__global__ __launch_bounds__(kThreads) void roi_pool_fwd_kernel(
    const float *__restrict__ features, const float *__restrict__ rois, int num_rois, int batch,
    int channels, int height, int width, int ph, int pw, float scale, int chan_per_block,
    int chunks, float *__restrict__ output, int *__restrict__ argmax)
{
    __shared__ int s_h0[kMaxPool], s_h1[kMaxPool], s_w0[kMaxPool], s_w1[kMaxPool];
    extern __shared__ __align__(16) unsigned char s_dyn[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // XCD-aware order: workgroups are dispatched round-robin over the 8 XCDs (blockIdx % 8), each with its
    // own 4 MB L2.  XCD x works through channel chunks x, x + 8, ... one after the other, each for ALL RoIs:
    // the planes of one chunk (8 channels x batch x H x W floats) then stay L2-resident while every RoI
    // window on them is read, instead of every RoI streaming its windows of all channels from memory.
    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int chunk = (j / num_rois) * 8 + xcd;
    const int n = j % num_rois;
    if (chunk >= chunks) return;
    const int c_begin = chunk * chan_per_block;
    const int c_count = min(chan_per_block, channels - c_begin);
    const int bins = ph * pw;

    const Roi r = decode_roi(rois + 5 * (size_t)n, scale, ph, pw);
    if (tid < ph) {
        const int hs = (int)floorf((float)tid * r.bin_h);
        const int he = (int)ceilf((float)(tid + 1) * r.bin_h);
        s_h0[tid] = clampi(hs + r.start_h, 0, height);
        s_h1[tid] = clampi(he + r.start_h, 0, height);
    }
    if (tid >= 64 && tid < 64 + pw) {
        const int q = tid - 64;
        const int ws = (int)floorf((float)q * r.bin_w);
        const int we = (int)ceilf((float)(q + 1) * r.bin_w);
        s_w0[q] = clampi(ws + r.start_w, 0, width);
        s_w1[q] = clampi(we + r.start_w, 0, width);
    }
    __syncthreads();

    const size_t o_base = ((size_t)n * channels + c_begin) * bins;
    const bool img_ok = r.img >= 0 && r.img < batch;
    // window columns [w_lo, w_hi): bin bounds are monotone in q
    const int w_lo = s_w0[0];
    const int w_hi = s_w1[pw - 1];
    const int w_win = max(w_hi - w_lo, 0);
    const int h_hi = s_h1[ph - 1];                 // bin bounds are monotone in p as well
    const bool stream_ok = r.bin_h >= 1.0f;
    if (!img_ok || w_win == 0) {      // every bin is empty: 0 / -1 (roi_pooling_kernel.cu:70-72)
        for (int idx = tid; idx < c_count * bins; idx += kThreads) {
            output[o_base + idx] = 0.0f;
            if (argmax) argmax[o_base + idx] = -1;
        }
        return;
    }
    // lanes = (channel sub-index, column): wp columns per channel, cpw channels per wavefront pass
    // (narrow windows are latency-bound, not lane-bound: never pack so far that wavefronts sit idle)
    const int cpw_max = max(1, chan_per_block / kWaves);
    const int wp = max(w_win <= 8 ? 8 : (w_win <= 16 ? 16 : (w_win <= 32 ? 32 : 64)), 64 / cpw_max);
    const int cpw = 64 / wp;
    const int slabs = (w_win + 63) / 64;          // > 1 only when wp == 64
    const int c_sub = lane / wp;
    const int w_off = lane - c_sub * wp;

    unsigned char *mine = s_dyn + (size_t)wave * roi_pool_wave_lds(ph, pw);
    float *colv = (float *)mine;
    int *colh = (int *)(mine + (size_t)ph * 64 * 4);
    float *binv = (float *)(mine + (size_t)ph * 64 * 8);
    int *binh = (int *)(binv + bins);
    int *binw = binh + bins;

    const int groups = (c_count + cpw - 1) / cpw;
    for (int g = wave; g < groups; g += kWaves) {
        const int c_loc = g * cpw + c_sub;
        const bool c_ok = c_loc < c_count;
        const int plane_off = ((r.img * channels) + c_begin + min(c_loc, c_count - 1)) * height * width;
        const float *__restrict__ src = features + plane_off;
        if (slabs > 1) {
            for (int b = lane; b < bins; b += 64) {
                binv[b] = -FLT_MAX;
                binh[b] = -1;
                binw[b] = -1;
            }
        }
        for (int slab = 0; slab < slabs; ++slab) {
            const int col0 = w_lo + slab * 64;                 // first column of this slab
            const int wcol = col0 + w_off;
            const bool col_ok = c_ok && wcol < w_hi;
            const float *__restrict__ colp = src + min(wcol, width - 1);
            // ---- phase 1: column maxima per pooled row --------------------------------------
            // Bin bounds are the same for every lane: they are pulled into scalar registers (UNI) so that
            // the bin bookkeeping is SALU work and scalar branches; the vector pipe only sees the loads and
            // one compare + two selects per element.  Lanes without a column (window narrower than the lane
            // group, channel past the end) read a valid dummy column and are reset when a bin is stored.
#define UNI(x) __builtin_amdgcn_readfirstlane(x)
            const int hi_u = UNI(h_hi);
            if (stream_ok) {
                // ONE pass down the window, 8 rows (8 independent loads) at a time.  With bin_h >= 1
                // consecutive bins share at most their boundary row (hstart[p+1] >= hend[p] - 1, also after
                // clamping), so two accumulators suffice: A for bin p, B for bin p + 1.
                int p = 0;
                while (p < ph && UNI(s_h1[p]) <= UNI(s_h0[p])) {     // leading empty bins (above the map)
                    colv[p * 64 + lane] = -FLT_MAX;
                    colh[p * 64 + lane] = -1;
                    ++p;
                }
                float a_v = -FLT_MAX, b_v = -FLT_MAX;
                int a_h = -1, b_h = -1;
                const int h_lo = p < ph ? UNI(s_h0[p]) : 0;
                int he_p = p < ph ? UNI(s_h1[p]) : 0x7fffffff;
                int hs_n = p + 1 < ph ? UNI(s_h0[p + 1]) : 0x7fffffff;
                int he_n = p + 1 < ph ? UNI(s_h1[p + 1]) : 0;
                for (int h0 = h_lo; h0 < hi_u && p < ph; h0 += 8) {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (h0 + i < hi_u) v[i] = colp[(h0 + i) * width];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int h = h0 + i;
                        if (h < hi_u && p < ph) {
                            if (v[i] > a_v) { a_v = v[i]; a_h = h; }
                            if (h >= hs_n && h < he_n) {                 // boundary row shared with bin p + 1
                                if (v[i] > b_v) { b_v = v[i]; b_h = h; }
                            }
                            while (p < ph && h + 1 >= he_p) {            // bin p is complete
                                colv[p * 64 + lane] = col_ok ? a_v : -FLT_MAX;
                                colh[p * 64 + lane] = col_ok ? a_h : -1;
                                a_v = b_v; a_h = b_h;
                                b_v = -FLT_MAX; b_h = -1;
                                ++p;
                                while (p < ph && UNI(s_h1[p]) <= UNI(s_h0[p])) {   // trailing empty bins
                                    colv[p * 64 + lane] = -FLT_MAX;
                                    colh[p * 64 + lane] = -1;
                                    ++p;
                                }
                                he_p = p < ph ? UNI(s_h1[p]) : 0x7fffffff;
                                hs_n = p + 1 < ph ? UNI(s_h0[p + 1]) : 0x7fffffff;
                                he_n = p + 1 < ph ? UNI(s_h1[p + 1]) : 0;
                            }
                        }
                    }
                }
                for (; p < ph; ++p) {                            // not reached for consistent tables
                    colv[p * 64 + lane] = -FLT_MAX;
                    colh[p * 64 + lane] = -1;
                }
            } else {
                // RoIs shorter than the pooled height (bin_h < 1): a row can belong to several bins.  The
                // window has fewer than ph rows: fetch them 8 at a time (independent loads -- one memory
                // latency per batch instead of one per bin), then fold them into every bin they belong to;
                // the per-bin accumulators live in the LDS column table.
                for (int p = 0; p < ph; ++p) {
                    colv[p * 64 + lane] = -FLT_MAX;
                    colh[p * 64 + lane] = -1;
                }
                const int h_lo = UNI(s_h0[0]);
                for (int h0 = h_lo; h0 < hi_u; h0 += 8) {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        v[i] = (col_ok && h0 + i < hi_u) ? colp[(h0 + i) * width] : -FLT_MAX;
                    for (int p = 0; p < ph; ++p) {
                        const int hs = UNI(s_h0[p]), he = UNI(s_h1[p]);
                        if (he <= h0 || hs >= h0 + 8 || he <= hs) continue;
                        float best = colv[p * 64 + lane];
                        int bh = colh[p * 64 + lane];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int h = h0 + i;
                            if (h >= hs && h < he && v[i] > best) { best = v[i]; bh = h; }
                        }
                        colv[p * 64 + lane] = best;
                        colh[p * 64 + lane] = bh;
                    }
                }
            }
#undef UNI
            wave_lds_sync();
            // ---- phase 2: bins from column maxima (w ascending; ties -> smaller row) ---------
            const int pairs = cpw * bins;
            for (int idx = lane; idx < pairs; idx += 64) {
                const int cs = idx / bins;
                const int bin = idx - cs * bins;
                const int p = bin / pw;
                const int q = bin - p * pw;
                const int hs = s_h0[p], he = s_h1[p], ws = s_w0[q], we = s_w1[q];
                const bool empty = (he <= hs) || (we <= ws);
                float best = -FLT_MAX;
                int bh = -1, bw = -1;
                if (slabs > 1) {
                    best = binv[bin];
                    bh = binh[bin];
                    bw = binw[bin];
                }
                const int a = max(ws, col0), e = min(we, col0 + 64);
                const float *cv = colv + p * 64 + cs * wp - col0;
                const int *ch = colh + p * 64 + cs * wp - col0;
                for (int w = a; w < e; ++w) {
                    const float v = cv[w];
                    const int hh = ch[w];
                    if (v > best || (v == best && hh >= 0 && (bh < 0 || hh < bh))) {
                        best = v;
                        bh = hh;
                        bw = w;
                    }
                }
                if (slabs > 1 && slab + 1 < slabs) {
                    binv[bin] = best;
                    binh[bin] = bh;
                    binw[bin] = bw;
                } else {
                    const int cl = g * cpw + cs;
                    if (cl < c_count) {
                        const size_t o = o_base + (size_t)cl * bins + bin;
                        const int po = ((r.img * channels) + c_begin + cl) * height * width;
                        output[o] = empty ? 0.0f : best;
                        if (argmax) argmax[o] = (empty || bh < 0) ? -1 : po + bh * width + bw;
                    }
                }
            }
            wave_lds_sync();
        }
    }
}

// ---- forward, second form: lanes = (pooled row p, group of 4 columns) -------------------------------------
// The kernel above issues one vector-memory instruction per window ROW (<= 256 bytes), and the CU's rate
// of such instructions is what bounds it (6.6 TB/s even on L2 hits).  Here a lane owns FOUR consecutive
// columns (one 16-byte load) of ONE pooled row p and walks only the rows of that bin: a wavefront load
// covers ph bins x up to 64/ph column groups at once, so a 33 x 32 window is read in ~6 instructions per
// channel instead of 32, and no cross-lane reduction is needed -- each lane ends with the (max, first row)
// of its 4 columns for its pooled row, exactly the column table phase 2 consumes.  Rows shared by two bins
// are simply read by both lanes.  Windows wider than 4 * (64 / ph) columns take several slabs; narrow ones
// pack several channels into the wavefront.  (ph <= 16 and W >= 4; the first form handles the rest.)
typedef float quad_t __attribute__((ext_vector_type(4)));
typedef quad_t quad_a4 __attribute__((aligned(4)));

__host__ __device__ inline size_t roi_pool_v2_wave_lds(int ph, int pw) { return 64 * 4 * 8 + (size_t)ph * pw * 12; }

__global__ __launch_bounds__(kThreads) void roi_pool_fwd_v2_kernel(
    const float *__restrict__ features, const float *__restrict__ rois, int num_rois, int batch,
    int channels, int height, int width, int ph, int pw, float scale, int chan_per_block,
    int chunks, float *__restrict__ output, int *__restrict__ argmax)
{
    __shared__ int s_h0[kMaxPool], s_h1[kMaxPool], s_w0[kMaxPool], s_w1[kMaxPool];
    extern __shared__ __align__(16) unsigned char s_dyn[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int xcd = blockIdx.x & 7;              // XCD chunk-major order, as above
    const int jb = blockIdx.x >> 3;
    const int chunk = (jb / num_rois) * 8 + xcd;
    const int n = jb % num_rois;
    if (chunk >= chunks) return;
    const int c_begin = chunk * chan_per_block;
    const int c_count = min(chan_per_block, channels - c_begin);
    const int bins = ph * pw;

    const Roi r = decode_roi(rois + 5 * (size_t)n, scale, ph, pw);
    if (tid < ph) {
        const int hs = (int)floorf((float)tid * r.bin_h);
        const int he = (int)ceilf((float)(tid + 1) * r.bin_h);
        s_h0[tid] = clampi(hs + r.start_h, 0, height);
        s_h1[tid] = clampi(he + r.start_h, 0, height);
    }
    if (tid >= 64 && tid < 64 + pw) {
        const int q = tid - 64;
        const int ws = (int)floorf((float)q * r.bin_w);
        const int we = (int)ceilf((float)(q + 1) * r.bin_w);
        s_w0[q] = clampi(ws + r.start_w, 0, width);
        s_w1[q] = clampi(we + r.start_w, 0, width);
    }
    __syncthreads();

    const size_t o_base = ((size_t)n * channels + c_begin) * bins;
    const bool img_ok = r.img >= 0 && r.img < batch;
    const int w_lo = s_w0[0];
    const int w_hi = s_w1[pw - 1];
    const int w_win = max(w_hi - w_lo, 0);
    if (!img_ok || w_win == 0) {      // every bin is empty: 0 / -1 (roi_pooling_kernel.cu:70-72)
        for (int idx = tid; idx < c_count * bins; idx += kThreads) {
            output[o_base + idx] = 0.0f;
            if (argmax) argmax[o_base + idx] = -1;
        }
        return;
    }
    const int CG = min(64 / ph, (w_win + 3) >> 2);          // column groups per slab
    const int SW = 4 * CG;                                  // slab width in columns
    const int slabs = (w_win + SW - 1) / SW;
    const int per_ch = ph * CG;                             // lanes per channel
    const int cpw = max(1, min(64 / per_ch, max(1, chan_per_block / kWaves)));
    const int cs = lane / per_ch;
    const int rem = lane - cs * per_ch;
    const int p = rem / CG, g = rem - p * CG;
    const bool lane_on = cs < cpw;

    unsigned char *mine = s_dyn + (size_t)wave * roi_pool_v2_wave_lds(ph, pw);
    float *colv = (float *)mine;                             // [cpw][ph][SW]   (cpw * ph * SW <= 256)
    int *colh = (int *)(mine + 64 * 4 * 4);
    float *binv = (float *)(mine + 64 * 4 * 8);              // running bin state, windows of several slabs
    int *binh = (int *)(binv + bins);
    int *binw = binh + bins;

    const int hs = lane_on ? s_h0[p] : 0, he = lane_on ? s_h1[p] : 0;
    const int groups = (c_count + cpw - 1) / cpw;
    for (int g0 = wave; g0 < groups; g0 += kWaves) {
        const int c_loc = g0 * cpw + cs;
        const bool c_ok = lane_on && c_loc < c_count;
        const int plane_off = ((r.img * channels) + c_begin + min(c_loc, c_count - 1)) * height * width;
        const float *__restrict__ src = features + plane_off;
        if (slabs > 1) {
            for (int b = lane; b < bins; b += 64) {
                binv[b] = -FLT_MAX;
                binh[b] = -1;
                binw[b] = -1;
            }
        }
        for (int slab = 0; slab < slabs; ++slab) {
            const int col0 = w_lo + slab * SW;
            const int startcol = min(col0 + 4 * g, width - 4);   // the last group is pulled back inside the row
            const bool on = c_ok && (col0 + 4 * g < w_hi);
            // ---- phase 1: this lane's 4 columns over the rows of its bin ------------------------------
            quad_t best = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
            int bh0 = -1, bh1 = -1, bh2 = -1, bh3 = -1;
            if (on) {
                const float *__restrict__ colp = src + startcol;
                // rows of the bin, 4 independent 16-byte loads in flight; rows past the bin's end re-read its
                // last row (an equal value never replaces the recorded first maximum)
                for (int h = hs; h < he; h += 4) {
                    quad_t v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        v[u] = *reinterpret_cast<const quad_a4 *>(colp + (size_t)min(h + u, he - 1) * width);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int hu = min(h + u, he - 1);
                        if (v[u].x > best.x) { best.x = v[u].x; bh0 = hu; }
                        if (v[u].y > best.y) { best.y = v[u].y; bh1 = hu; }
                        if (v[u].z > best.z) { best.z = v[u].z; bh2 = hu; }
                        if (v[u].w > best.w) { best.w = v[u].w; bh3 = hu; }
                    }
                }
            }
            if (on) {
                // (columns at or past w_hi are never read by phase 2, so lanes without a column write nothing;
                // a pulled-back last group rewrites columns of its neighbour with identical values)
                const int base = (cs * ph + p) * SW + (startcol - col0);
                const float bv[4] = {best.x, best.y, best.z, best.w};
                const int bhh[4] = {bh0, bh1, bh2, bh3};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ce = startcol - col0 + e;              // column inside the slab
                    if (ce >= 0 && ce < SW) {
                        colv[base + e] = bv[e];
                        colh[base + e] = bhh[e];
                    }
                }
            }
            wave_lds_sync();
            // ---- phase 2: bins from column maxima (w ascending; ties -> smaller row) -------------------
            const int pairs = cpw * bins;
            for (int idx = lane; idx < pairs; idx += 64) {
                const int cs2 = idx / bins;
                const int bin = idx - cs2 * bins;
                const int p2 = bin / pw;
                const int q2 = bin - p2 * pw;
                const int h0 = s_h0[p2], h1 = s_h1[p2], ws = s_w0[q2], we = s_w1[q2];
                const bool empty = (h1 <= h0) || (we <= ws);
                float bst = -FLT_MAX;
                int bh = -1, bw = -1;
                if (slabs > 1) {
                    bst = binv[bin];
                    bh = binh[bin];
                    bw = binw[bin];
                }
                const int a = max(ws, col0), e = min(we, col0 + SW);
                const float *cv = colv + (cs2 * ph + p2) * SW - col0;
                const int *ch = colh + (cs2 * ph + p2) * SW - col0;
                for (int w = a; w < e; ++w) {
                    const float v = cv[w];
                    const int hh = ch[w];
                    if (v > bst || (v == bst && hh >= 0 && (bh < 0 || hh < bh))) {
                        bst = v;
                        bh = hh;
                        bw = w;
                    }
                }
                if (slabs > 1 && slab + 1 < slabs) {
                    binv[bin] = bst;
                    binh[bin] = bh;
                    binw[bin] = bw;
                } else {
                    const int cl = g0 * cpw + cs2;
                    if (cl < c_count) {
                        const size_t o = o_base + (size_t)cl * bins + bin;
                        const int po = ((r.img * channels) + c_begin + cl) * height * width;
                        output[o] = empty ? 0.0f : bst;
                        if (argmax) argmax[o] = (empty || bh < 0) ? -1 : po + bh * width + bw;
                    }
                }
            }
            wave_lds_sync();
        }
    }
}

__global__ __launch_bounds__(kThreads) void roi_pool_bwd_kernel(
    const float *__restrict__ top_grad, const float *__restrict__ rois,
    const int *__restrict__ argmax, int num_rois, int batch, int channels, int height, int width,
    int ph, int pw, float scale, int per_roi, int blocks_per_roi, float *__restrict__ bottom_grad)
{
    // blockIdx.x = roi * blocks_per_roi + j; j covers channels*ph*pw of that roi
    const int n = blockIdx.x / blocks_per_roi;
    const int idx = (blockIdx.x - n * blocks_per_roi) * kThreads + threadIdx.x;
    if (idx >= per_roi) return;
    const size_t o = (size_t)n * per_roi + idx;
    const int a = argmax[o];
    if (a < 0) return;
    const Roi r = decode_roi(rois + 5 * (size_t)n, scale, ph, pw);
    const int q = idx % pw;
    const int p = (idx / pw) % ph;
    const int c = idx / (pw * ph);
    // decode the argmax position (flat NCHW index)
    const int w = a % width;
    const int h = (a / width) % height;
    const int ca = (a / (width * height)) % channels;
    const int img = a / (width * height * channels);
    if (img != r.img || ca != c) return;  // kernel.cu:147 (+ the index identity)
    if (!(w >= r.start_w && w <= r.end_w && h >= r.start_h && h <= r.end_h)) return;  // :155-160
    int p0 = (int)floorf((float)(h - r.start_h) / r.bin_h);       // :173-176
    int p1 = (int)ceilf((float)(h - r.start_h + 1) / r.bin_h);
    int q0 = (int)floorf((float)(w - r.start_w) / r.bin_w);
    int q1 = (int)ceilf((float)(w - r.start_w + 1) / r.bin_w);
    p0 = clampi(p0, 0, ph);
    p1 = clampi(p1, 0, ph);
    q0 = clampi(q0, 0, pw);
    q1 = clampi(q1, 0, pw);
    if (p < p0 || p >= p1 || q < q0 || q >= q1) return;
    atomicAdd(bottom_grad + a, top_grad[o]);
}

}  // namespace

extern "C" {

int fi_roi_pool_forward(const float *features, const float *rois, int num_rois, int batch,
                        int channels, int height, int width, int pooled_h, int pooled_w,
                        float spatial_scale, float *output, int32_t *argmax, fi_stream_t stream)
{
    FI_REQUIRE(num_rois >= 0 && batch > 0 && channels > 0 && height > 0 && width > 0, "bad sizes");
    FI_REQUIRE(pooled_h >= 1 && pooled_w >= 1, "pooled size must be >= 1");
    if (pooled_h > kMaxPool || pooled_w > kMaxPool) {
        fi::set_error("pooled size %dx%d exceeds the supported maximum %d", pooled_h, pooled_w,
                      kMaxPool);
        return FI_ERR_UNSUPPORTED;
    }
    FI_REQUIRE((long)batch * channels * height * width < 2147483647L,
               "features tensor too large for int32 argmax (reference limitation)");
    if (num_rois == 0) return FI_OK;
    FI_REQUIRE(features && rois && output, "null pointer");
    // 8 channels per workgroup: the largest RoIs (whole-map windows) set the critical path, so their
    // channels are spread over many workgroups; narrow RoIs pack all 8 channels into one wavefront pass
    const int cpb = pooled_h >= 14 ? 4 : 8;     // measured: 7x7 181 us at 4 or 8 (237 at 2, 235 at 16); 14x14 408 / 458 us
    const int chunks = fi::ceil_div(channels, cpb);
    FI_REQUIRE((long)num_rois * (chunks + 8) < 2147483647L, "grid too large");
    hipStream_t st = (hipStream_t)stream;
    fi::ProfScope prof(FI_K_ROIPOOL_FWD, st);
    if (pooled_h > 28 || pooled_w > 28) {
        hipLaunchKernelGGL(roi_pool_fwd_simple_kernel, dim3((unsigned)((long)num_rois * chunks)), dim3(kThreads),
                           0, st, features, rois, num_rois, batch, channels, height, width, pooled_h,
                           pooled_w, spatial_scale, cpb, chunks, output, argmax);
    } else if (pooled_h <= 16 && width >= 4) {
        const size_t lds = roi_pool_v2_wave_lds(pooled_h, pooled_w) * kWaves;  // 4 * (2 KB + bins * 12 B)
        const long grid = (long)num_rois * fi::ceil_div(chunks, 8) * 8;
        hipLaunchKernelGGL(roi_pool_fwd_v2_kernel, dim3((unsigned)grid), dim3(kThreads), lds, st, features, rois,
                           num_rois, batch, channels, height, width, pooled_h, pooled_w, spatial_scale, cpb, chunks,
                           output, argmax);
    } else {
        const size_t lds = roi_pool_wave_lds(pooled_h, pooled_w) * kWaves;     // <= 4 * (14 KB + 9.2 KB)
        const long grid = (long)num_rois * fi::ceil_div(chunks, 8) * 8;
        hipLaunchKernelGGL(roi_pool_fwd_kernel, dim3((unsigned)grid), dim3(kThreads),
                           lds, st, features, rois, num_rois, batch, channels, height, width, pooled_h,
                           pooled_w, spatial_scale, cpb, chunks, output, argmax);
    }
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

int fi_roi_pool_backward(const float *top_grad, const float *rois, const int32_t *argmax,
                         int num_rois, int batch, int channels, int height, int width, int pooled_h,
                         int pooled_w, float spatial_scale, float *bottom_grad, fi_stream_t stream)
{
    FI_REQUIRE(num_rois >= 0 && batch > 0 && channels > 0 && height > 0 && width > 0, "bad sizes");
    FI_REQUIRE(pooled_h >= 1 && pooled_w >= 1, "pooled size must be >= 1");
    FI_REQUIRE(bottom_grad != nullptr, "null bottom_grad");
    hipStream_t st = (hipStream_t)stream;
    FI_HIP_CHECK(hipMemsetAsync(bottom_grad, 0,
                                sizeof(float) * (size_t)batch * channels * height * width, st));
    if (num_rois == 0) return FI_OK;
    FI_REQUIRE(top_grad && rois && argmax, "null pointer");
    const int per_roi = channels * pooled_h * pooled_w;
    fi::ProfScope prof(FI_K_ROIPOOL_BWD, st);
    const int bpr = fi::ceil_div(per_roi, kThreads);
    FI_REQUIRE((long)bpr * num_rois < 2147483647L, "grid too large");
    hipLaunchKernelGGL(roi_pool_bwd_kernel, dim3((unsigned)((long)bpr * num_rois)), dim3(kThreads), 0,
                       st, top_grad, rois, argmax, num_rois, batch, channels, height, width, pooled_h,
                       pooled_w, spatial_scale, per_roi, bpr, bottom_grad);
    FI_HIP_CHECK(hipGetLastError());
    return FI_OK;
}

}  // extern "C"
