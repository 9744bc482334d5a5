// nms.hip -- greedy IoU NMS on score-sorted boxes, entirely on the GPU (gfx950).
//
// Specification: cpu_nms, lib/nms/src/nms.c:35-63 (IoU arithmetic, `>=`); the CUDA
// variant's `>` (lib/nms/src/cuda/nms_kernel.cu:63) is selectable.  Oracle:
// orc_nms() in oracle/fi_oracle.c; keep indices must match it exactly.
//
// Two kernels, no host round trip (the reference copied a 4.5 MB bit-matrix to the
// host and scanned it there, lib/nms/src/nms_cuda.c:33-58):
//
//  1. nms_mask_kernel: the N x N/64 suppression bit-matrix, UPPER triangle only
//     (the reference also computed the unused lower half).  One wavefront (64
//     lanes = 64 row boxes) per 64x64 tile -- the tile width equals the CDNA
//     wavefront, so a row's 64 comparisons are one u64 built in a register; the 64
//     column boxes sit in LDS and are read as broadcasts (all lanes, one address).
//     A 256-thread workgroup covers 4 column tiles of one row block.
//
//  2. nms_scan_kernel: the greedy pass, one workgroup per image.  The running
//     "removed" bit-vector lives in registers (one u64 word per thread).  Work
//     proceeds in blocks of 64 boxes:
//       a. the diagonal 64x64 tile of the block resolves the block's internal
//          order dependence -- 64 scalar steps on one wavefront using
//          v_readlane (no memory traffic);
//       b. the block's survivors then OR their mask rows into the removed
//          vector: every thread owns one column word, so each survivor's row is
//          one coalesced read, and rows are fetched 8 at a time so that the
//          ~1 us L2 latency is paid once per block, not once per kept box;
//       c. the next block's diagonal words are prefetched during (b).
//     The pass stops as soon as max_keep survivors exist.
#include "fi_common.h"

namespace {

typedef unsigned long long u64;
constexpr int kTile = 64;

// IoU with the +1 pixel convention; operation order as nms.c:49-58.
__device__ __forceinline__ bool suppresses(float a0, float a1, float a2, float a3, float area_a,
                                           float b0, float b1, float b2, float b3, float area_b,
                                           float thresh, int strict)
{
    const float l = fmaxf(a0, b0);
    const float t = fmaxf(a1, b1);
    const float r = fminf(a2, b2);
    const float bt = fminf(a3, b3);
    const float dw = r - l;
    const float dh = bt - t;
    const float w = fmaxf(0.0f, dw + 1.0f);
    const float h = fmaxf(0.0f, dh + 1.0f);
    const float inter = w * h;
    const float s = area_a + area_b;
    const float uni = s - inter;
    const float iou = inter / uni;
    return strict ? (iou > thresh) : (iou >= thresh);
}

__device__ __forceinline__ float box_area(float b0, float b1, float b2, float b3)
{
    // (x2 - x1 + 1) * (y2 - y1 + 1), lib/nms/pth_nms.py:13
    const float dw = b2 - b0;
    const float dh = b3 - b1;
    return (dw + 1.0f) * (dh + 1.0f);
}

// grid: (ceil(col_blocks / 4), row_blocks, batch), block: 256 = 4 wavefronts.
__global__ __launch_bounds__(256) void nms_mask_kernel(const float *__restrict__ boxes,
                                                       int num_boxes, int box_stride, float thresh,
                                                       int strict, int col_blocks,
                                                       u64 *__restrict__ mask)
{
    __shared__ float s_box[4][kTile][5];  // x1,y1,x2,y2,area of the column boxes
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int row_block = blockIdx.y;
    const int col_block = blockIdx.x * 4 + wave;
    const float *img_boxes = boxes + (size_t)blockIdx.z * num_boxes * box_stride;
    u64 *img_mask = mask + (size_t)blockIdx.z * num_boxes * col_blocks;

    // whole workgroup below the diagonal: nothing to do (never read by the scan)
    if ((int)(blockIdx.x * 4 + 3) < row_block) return;

    const bool active = (col_block < col_blocks) && (col_block >= row_block);
    const int col_index = col_block * kTile + lane;
    if (active && col_index < num_boxes) {
        const float *b = img_boxes + (size_t)col_index * box_stride;
        const float b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
        s_box[wave][lane][0] = b0;
        s_box[wave][lane][1] = b1;
        s_box[wave][lane][2] = b2;
        s_box[wave][lane][3] = b3;
        s_box[wave][lane][4] = box_area(b0, b1, b2, b3);
    }
    __syncthreads();
    if (!active) return;

    const int row_index = row_block * kTile + lane;
    if (row_index >= num_boxes) return;
    const float *a = img_boxes + (size_t)row_index * box_stride;
    const float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
    const float area_a = box_area(a0, a1, a2, a3);

    const int col_size = min(kTile, num_boxes - col_block * kTile);
    const int start = (row_block == col_block) ? lane + 1 : 0;
    u64 bits = 0;
    for (int j = 0; j < col_size; ++j) {
        const float *b = s_box[wave][j];
        const bool hit = suppresses(a0, a1, a2, a3, area_a, b[0], b[1], b[2], b[3], b[4], thresh,
                                    strict);
        if (hit && j >= start) bits |= (1ULL << j);
    }
    img_mask[(size_t)row_index * col_blocks + col_block] = bits;
}

__device__ __forceinline__ u64 readlane_u64(u64 v, int lane)
{
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)(v & 0xffffffffULL), lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ u64 uniform_u64(u64 v)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(v & 0xffffffffULL));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}

// One workgroup per image.  WPT = removed-words per thread (col_blocks <= 256*WPT).
template <int WPT>
__global__ __launch_bounds__(256) void nms_scan_kernel(const u64 *__restrict__ mask, int num_boxes,
                                                       int col_blocks, int max_keep,
                                                       long long *__restrict__ keep_out,
                                                       int *__restrict__ num_out)
{
    __shared__ u64 s_cur;    // removed-word of the current block
    __shared__ u64 s_kept;   // survivors of the current block
    __shared__ int s_count;  // survivors so far
    const int tid = threadIdx.x;
    const u64 *img_mask = mask + (size_t)blockIdx.x * num_boxes * col_blocks;
    long long *img_keep = keep_out + (size_t)blockIdx.x * num_boxes;

    u64 removed[WPT];
#pragma unroll
    for (int w = 0; w < WPT; ++w) removed[w] = 0;
    if (tid == 0) s_count = 0;

    const int limit = (max_keep > 0) ? max_keep : num_boxes;
    // prefetch the diagonal words of block 0 (wave 0 only)
    u64 diag_next = 0;
    if (tid < kTile && tid < num_boxes) diag_next = img_mask[(size_t)tid * col_blocks + 0];
    __syncthreads();

    for (int blk = 0; blk < col_blocks; ++blk) {
        // -- publish this block's removed word ---------------------------------
#pragma unroll
        for (int w = 0; w < WPT; ++w)
            if (tid + w * 256 == blk) s_cur = removed[w];
        __syncthreads();

        // -- (a) resolve the block on wavefront 0 -------------------------------
        if (tid < kTile) {
            const u64 diag = diag_next;
            const int nxt = (blk + 1) * kTile + tid;
            diag_next = 0;
            if (blk + 1 < col_blocks && nxt < num_boxes)
                diag_next = img_mask[(size_t)nxt * col_blocks + (blk + 1)];  // (c) prefetch

            const int valid = min(kTile, num_boxes - blk * kTile);
            u64 cur = uniform_u64(s_cur);  // scalar registers: the 64 steps below run on the SALU
            if (valid < kTile) cur |= ~0ULL << valid;  // boxes past the end count as removed
            u64 kept = 0;
#pragma unroll
            for (int k = 0; k < kTile; ++k) {
                const u64 dk = readlane_u64(diag, k);  // uniform
                if (!((cur >> k) & 1ULL)) {
                    kept |= 1ULL << k;
                    cur |= dk;
                }
            }
            const int count = __builtin_amdgcn_readfirstlane(s_count);
            int n_kept = __popcll(kept);
            if (count + n_kept > limit) {  // keep only the first (limit - count) survivors
                int room = limit - count;
                u64 trimmed = 0, rest = kept;
                while (room > 0 && rest) {
                    const u64 low = rest & (~rest + 1ULL);
                    trimmed |= low;
                    rest ^= low;
                    --room;
                }
                kept = trimmed;
                n_kept = __popcll(kept);
            }
            if ((kept >> tid) & 1ULL) {
                const int rank = __popcll(kept & ((1ULL << tid) - 1ULL));
                img_keep[count + rank] = (long long)(blk * kTile + tid);
            }
            if (tid == 0) {
                s_kept = kept;
                s_count = count + n_kept;
            }
        }
        __syncthreads();
        const u64 kept = s_kept;
        const bool done = s_count >= limit;
        if (done) break;

        // -- (b) OR the survivors' rows into the removed vector ------------------
        if (kept) {
#pragma unroll
            for (int w = 0; w < WPT; ++w) {
                const int col = tid + w * 256;
                if (col > blk && col < col_blocks) {
                    const u64 *base = img_mask + (size_t)blk * kTile * col_blocks + col;
                    u64 m = kept;
                    u64 acc = 0;
                    while (m) {
                        u64 v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            v[u] = 0;
                            if (m) {
                                const int k = __ffsll((long long)m) - 1;
                                m &= m - 1ULL;
                                v[u] = base[(size_t)k * col_blocks];
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc |= v[u];
                    }
                    removed[w] |= acc;
                }
            }
        }
        __syncthreads();  // s_cur / s_kept are rewritten next iteration
    }
    if (tid == 0) num_out[blockIdx.x] = s_count;
}

}  // namespace

extern "C" {

size_t fi_nms_workspace_bytes(int batch, int num_boxes)
{
    if (batch <= 0 || num_boxes <= 0) return 0;
    const size_t col_blocks = (size_t)fi::ceil_div(num_boxes, kTile);
    return sizeof(u64) * (size_t)batch * (size_t)num_boxes * col_blocks;
}

int fi_nms_sorted(const float *boxes, int batch, int num_boxes, int box_stride, float thresh,
                  int strict, int max_keep, int64_t *keep_out, int32_t *num_out, void *workspace,
                  fi_stream_t stream)
{
    FI_REQUIRE(batch >= 1 && num_boxes >= 0, "batch >= 1, num_boxes >= 0");
    FI_REQUIRE(box_stride >= 4, "box_stride >= 4");
    FI_REQUIRE(num_out != nullptr, "null num_out");
    hipStream_t st = (hipStream_t)stream;
    if (num_boxes == 0) {
        FI_HIP_CHECK(hipMemsetAsync(num_out, 0, sizeof(int32_t) * batch, st));
        return FI_OK;
    }
    FI_REQUIRE(boxes && keep_out && workspace, "null pointer");
    const int col_blocks = fi::ceil_div(num_boxes, kTile);
    if (col_blocks > 1024) {
        fi::set_error("fi_nms_sorted supports at most 65536 boxes per image (got %d)", num_boxes);
        return FI_ERR_UNSUPPORTED;
    }
    u64 *mask = (u64 *)workspace;
    {
        fi::ProfScope prof(FI_K_NMS_MASK, st);
        dim3 grid(fi::ceil_div(col_blocks, 4), col_blocks, batch);
        hipLaunchKernelGGL(nms_mask_kernel, grid, dim3(256), 0, st, boxes, num_boxes, box_stride,
                           thresh, strict, col_blocks, mask);
        FI_HIP_CHECK(hipGetLastError());
    }
    {
        fi::ProfScope prof(FI_K_NMS_SCAN, st);
        long long *keep = (long long *)keep_out;
        if (col_blocks <= 256)
            hipLaunchKernelGGL(nms_scan_kernel<1>, dim3(batch), dim3(256), 0, st, mask, num_boxes,
                               col_blocks, max_keep, keep, num_out);
        else if (col_blocks <= 512)
            hipLaunchKernelGGL(nms_scan_kernel<2>, dim3(batch), dim3(256), 0, st, mask, num_boxes,
                               col_blocks, max_keep, keep, num_out);
        else
            hipLaunchKernelGGL(nms_scan_kernel<4>, dim3(batch), dim3(256), 0, st, mask, num_boxes,
                               col_blocks, max_keep, keep, num_out);
        FI_HIP_CHECK(hipGetLastError());
    }
    return FI_OK;
}

}  // extern "C"
