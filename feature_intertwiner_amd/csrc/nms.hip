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

// The same predicate without the IEEE division in all but the borderline cases: q = inter * rcp(uni) is within a few
// 2^-24 of the rounded quotient (v_rcp_f32: 1 ulp), so a q that is more than 2^-20 (relative) away from the threshold
// decides the comparison the division would make; everything else -- and every non-finite case: 0 * inf, NaN -- falls
// through both tests and takes the division.  The division is 10 of the ~25 vector instructions of a pair test.
__device__ __forceinline__ bool suppresses_fast(float a0, float a1, float a2, float a3, float area_a,
                                                float b0, float b1, float b2, float b3, float area_b,
                                                float thresh, float t_lo, float t_hi, int strict)
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
    const float q = inter * __builtin_amdgcn_rcpf(uni);
    if (q < t_lo) return false;
    if (q > t_hi) return true;
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

    // whole workgroup below the first sub-diagonal: nothing to do (never read by the scan).  The sub-diagonal tile
    // (row block b+1, column block b) IS written: by symmetry it is the transpose of tile (b, b+1), lane c = "which boxes
    // of block b suppress box c of block b+1" -- the wide scan's resolver turns it into block b's contribution to the next
    // removed word with one AND and one ballot
    if ((int)(blockIdx.x * 4 + 3) < row_block - 1) return;

    const bool active = (col_block < col_blocks) && (col_block >= row_block - 1);
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
    // diagonal tile: every pair but (k, k) -- the IoU test is symmetric bit for bit, so the bits BELOW the diagonal of row
    // k are "who among the earlier boxes of the block suppresses k" (what nms_scan_wide_kernel's resolver reads); the
    // one-word-per-thread scan masks them off
    const int self = (row_block == col_block) ? lane : -1;
    // (a positive finite threshold only: the margins below assume it)
    const bool fast = thresh > 1e-6f && thresh < 1e6f;
    const float t_lo = thresh * (1.0f - 0x1p-20f), t_hi = thresh * (1.0f + 0x1p-20f);
    u64 bits = 0;
    for (int j = 0; j < col_size; ++j) {
        const float *b = s_box[wave][j];
        const bool hit = fast ? suppresses_fast(a0, a1, a2, a3, area_a, b[0], b[1], b[2], b[3], b[4], thresh, t_lo, t_hi, strict)
                              : suppresses(a0, a1, a2, a3, area_a, b[0], b[1], b[2], b[3], b[4], thresh, strict);
        if (hit && j != self) bits |= (1ULL << j);
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
    const u64 above = (~0ULL << 1) << (tid & 63);             // bits k+1 .. 63 of row k (the tile is stored whole)
    if (tid < kTile && tid < num_boxes) diag_next = img_mask[(size_t)tid * col_blocks + 0] & above;
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
                diag_next = img_mask[(size_t)nxt * col_blocks + (blk + 1)] & above;  // (c) prefetch

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


// ---------------------------------------------------------------------------------------------------------------------
// nms_scan_wide_kernel (round 6): the same greedy pass with the memory latency and the row ORs taken off the serial chain.
// One workgroup of 1024 threads per image:
//   * wavefront 0 is the RESOLVER: per block of 64 boxes it walks the block's survivors only (s_ff1 over the not-removed
//     bits: one step per KEPT box instead of 64 steps), and folds the block's own contribution to the NEXT column word
//     (tile (blk, blk+1), one u64 per lane) with six DPP steps -- so the chain from block to block never waits for memory;
//   * the other 960 threads are APPLIERS, T per column word, R = 64 / T rows of a block each: they OR block blk-1's kept
//     rows into their private part of the removed vector one block BEHIND the resolver, from registers that were loaded
//     D blocks ahead (the addresses do not depend on the outcome, only the use does), and the T parts of column blk+1
//     meet in LDS (64-bit atomic OR) one iteration before the resolver needs that word.
// One barrier per block.  Same results: the order of ORs does not matter.
// ---------------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ unsigned dpp0(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

// OR over the wavefront, result valid in lane 63 (row_shr 1/2/4/8, row_bcast:15, row_bcast:31)
__device__ __forceinline__ unsigned wave_or_to_lane63(unsigned v)
{
    v |= dpp0<0x111>(v);
    v |= dpp0<0x112>(v);
    v |= dpp0<0x114>(v);
    v |= dpp0<0x118>(v);
    v |= dpp0<0x142>(v);
    v |= dpp0<0x143>(v);
    return v;
}

constexpr int kWideThreads = 1024;
constexpr int kWideUnroll = 4;

// Loads of the wide scan are inline asm with hand-counted waits: the compiler's own waits in front of a register ring
// that is refilled inside a loop drain the queue (measured on the ISA: vmcnt(7..0) where vmcnt(31..24) is enough), and a
// __syncthreads() makes every wavefront with a store in flight wait for it.  Rules ("Inline-asm
// memory operations"): a wavefront that issues asm loads issues NO other vector-memory operation (loads retire in order
// among themselves only), the loaded registers reach their first use through the wait's "+v" operands, addresses are
// whole VGPR pairs (no SGPR base).
__device__ __forceinline__ void wide_load(u64 &dst, const u64 *p)
{
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
// (scalar base + 32-bit lane offset: no vector instruction per load; the base is computed on the scalar unit)
__device__ __forceinline__ void wide_load(u64 &dst, unsigned byte_off, const u64 *base)
{
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void wide_wait(u64 &a, u64 &b)
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wide_wait(u64 (&r)[8])
{
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wide_wait(u64 (&r)[16])
{
    asm volatile("s_waitcnt vmcnt(%16)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "n"(N) : "memory");
}
typedef unsigned int wide_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wide_load16(wide_u32x4 &dst, unsigned byte_off, const u64 *base)
{
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void wide_wait(wide_u32x4 (&r)[4])
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wide_wait(wide_u32x4 (&r)[8])
{
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 : "n"(N) : "memory");
}
// LDS-only barrier: this wavefront's LDS operations have completed, then s_barrier (no vmcnt wait)
#if defined(FI_NMS_EXP) && (FI_NMS_EXP & 8)
__device__ __forceinline__ void wide_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
__device__ __forceinline__ void wide_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

struct WideSlot {                                  // what the resolver publishes per block (one 16-byte LDS store)
    u64 kept;                                      // survivors of the block
    int before;                                    // survivors before it
    int done;                                      // the limit is reached
};

__device__ __forceinline__ void lds_or_b64(u64 *p, u64 v)
{
    // (a plain LDS atomic: the compiler would rewrite atomicOr into a scalar loop over the active lanes)
    asm volatile("ds_or_b64 %0, %1" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) u64 *)p), "v"(v) : "memory");
}

template <int SUBS>
__global__ __launch_bounds__(kWideThreads) void nms_scan_wide_kernel(const u64 *__restrict__ mask, int num_boxes,
                                                                     int col_blocks, int max_keep,
                                                                     long long *__restrict__ keep_out,
                                                                     int *__restrict__ num_out)
{
    constexpr int R = kTile / SUBS;                // rows of a block per applier thread (two column words each)
    constexpr int CPW = kTile / SUBS;              // column pairs per applier wavefront
    constexpr int D = (SUBS == 16) ? 4 : 2;        // blocks loaded ahead (2 * R * D u64 registers)
    static_assert(kWideUnroll % D == 0 && (R == 4 || R == 8), "ring indexing / wait helper");
    __shared__ u64 s_rem[2];                       // removed word of column it: contributions of blocks <= it-2, ORed in
                                                   // by the column's T appliers during iteration it-1
    __shared__ __attribute__((aligned(16))) WideSlot s_slot[2];
    const int tid = threadIdx.x;
    const u64 *img_mask = mask + (size_t)blockIdx.x * num_boxes * col_blocks;
    long long *img_keep = keep_out + (size_t)blockIdx.x * num_boxes;
    const int limit = (max_keep > 0) ? max_keep : num_boxes;
    const int last_col = col_blocks - 1;
    const int tail = num_boxes - last_col * kTile;           // boxes in the last block (1..64)
    const unsigned row_bytes = (unsigned)col_blocks * (unsigned)sizeof(u64);

    if (tid == 0) {
        s_rem[0] = s_rem[1] = 0;
        s_slot[0].done = s_slot[1].done = 0;
    }
    __syncthreads();
    // Three roles, each with its own copy of the block loop and ONE barrier per block.
    if (tid < kTile) {
        // ---- resolver (wavefront 0) ------------------------------------------------------------------------------------
        // lane k: row k of the diagonal tile (who suppresses whom inside the block) and row k of the sub-diagonal tile of
        // the NEXT block (which boxes of this block suppress box k of the next), blocks it .. it+3 in registers.
        // Addresses: a scalar base per block + the lane's row offset; in the last block the rows past the end read the
        // last box's row (never used).
        u64 diag[kWideUnroll], sub[kWideUnroll];
        const unsigned off_full = (unsigned)tid * row_bytes;
        const unsigned off_tail = (unsigned)min(tid, tail - 1) * row_bytes;
        auto load_diag = [&](int blk, u64 &dg, u64 &sb) {
            const int b = min(blk, last_col);                // past the end: the last block again
            const int b1 = min(blk + 1, last_col);           // no next block: the last block's own rows (unused)
            wide_load(dg, b == last_col ? off_tail : off_full, img_mask + (size_t)b * kTile * col_blocks + b);
            wide_load(sb, b1 == last_col ? off_tail : off_full, img_mask + (size_t)b1 * kTile * col_blocks + b);
        };
#pragma unroll
        for (int d = 0; d < kWideUnroll; ++d) load_diag(d, diag[d], sub[d]);
        u64 fast = 0;                              // block it-1's contribution to column it
        int count = 0;
        bool done = false;
        for (int base = 0; base < col_blocks && !done; base += kWideUnroll) {
#pragma unroll
            for (int d = 0; d < kWideUnroll; ++d) {
                const int it = base + d;
                if (it >= col_blocks) break;
                u64 cur = uniform_u64(s_rem[it & 1]) | fast;
                if (tid == 0) s_rem[it & 1] = 0;            // the appliers OR into it again in iteration it + 1
                if (it == last_col && tail < kTile) cur |= ~0ULL << tail;   // boxes past the end count as removed
#if !(defined(FI_NMS_EXP) && (FI_NMS_EXP & 4))
                wide_wait<2 * (kWideUnroll - 1)>(diag[d], sub[d]);
#endif
                // The block's own order dependence, all 64 boxes at once: lane k holds the earlier boxes of the block
                // that suppress k.  A candidate no candidate suppresses is kept for certain; whoever such a box
                // suppresses is out; repeat until nothing leaves (the lowest undecided candidate is always decided, so
                // the rounds are bounded by the longest suppression chain -- 2 or 3 in practice, not one step per box).
                const u64 below = diag[d] & ((1ULL << tid) - 1ULL);
                u64 cand = ~cur;
                for (;;) {
                    const bool c = (cand >> tid) & 1ULL;
                    const u64 sure = __ballot(c && (below & cand) == 0);
                    const u64 out = __ballot(c && (below & sure) != 0);
                    if (out == 0) break;                    // then sure == cand
                    cand &= ~out;
                }
                u64 kept = cand;
                int n_kept = __popcll(kept);
                if (count + n_kept > limit) {               // keep only the first (limit - count) survivors
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
                // this block's contribution to the next removed word: lane c = box c of the next block
                fast = (it < last_col) ? __ballot((sub[d] & kept) != 0) : 0ULL;
                const int is_done = (count + n_kept >= limit) ? 1 : 0;
                if (tid == 0) {
                    WideSlot w;
                    w.kept = kept; w.before = count; w.done = is_done;
                    s_slot[it & 1] = w;
                }
                count += n_kept;
                load_diag(it + kWideUnroll, diag[d], sub[d]);
                wide_barrier();
                if (is_done) { done = true; break; }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the last prefetches, before this wavefront's only store
        if (tid == 0) num_out[blockIdx.x] = count;
    } else if (tid >= kWideThreads - kTile) {
        // ---- writer (last wavefront): the kept indices of block it-1, one block behind; stores only ---------------------
        const int lane = tid - (kWideThreads - kTile);
        for (int it = 0;; ++it) {
            if (it > 0) {
                const WideSlot w = s_slot[(it - 1) & 1];
                if ((w.kept >> lane) & 1ULL)
                    img_keep[w.before + __popcll(w.kept & ((1ULL << lane) - 1ULL))] = (long long)((it - 1) * kTile + lane);
                if (w.done || it == col_blocks) break;
            }
            wide_barrier();
        }
    } else {
        // ---- appliers ------------------------------------------------------------------------------------------------
        // The bound of this role is the NUMBER of vector loads the CU issues per block (measured: a quarter of the loads,
        // half the time; the same loads on two cache lines, the same time), so: 16-byte loads (two column words per
        // lane), and a wavefront owns a narrow band of columns -- CPW pairs x SUBS row groups -- which it leaves for good
        // once the diagonal has passed it (on average half of the wavefronts are loading).
        const int wave = (tid - kTile) >> 6, lane = tid & 63;
        const int sub = lane / CPW, cp = wave * CPW + lane % CPW;
        const int col0 = 2 * cp;
        const bool applier = col0 < col_blocks && col_blocks >= 3;
        const int wave_max_col = min(2 * (wave * CPW + CPW - 1) + 1, last_col);
        const int j_last = (2 * wave * CPW < col_blocks && col_blocks >= 3) ? wave_max_col - 2 : -1;   // last block this wavefront applies
        // The per-thread part of an address never changes (R byte offsets in registers); the block's part is a scalar base.
        // Blocks up to j_last <= last_col - 2 only: the word after an odd row's end is the next row's first, inside the mask.
        unsigned off[R];
#pragma unroll
        for (int r = 0; r < R; ++r) off[r] = applier ? (unsigned)(sub * R + r) * row_bytes + (unsigned)col0 * 8u : 0u;
        u64 rem0 = 0, rem1 = 0;
        wide_u32x4 buf[D][R];
        auto load_rows = [&](int blk, wide_u32x4(&dst)[R]) {
            if (blk > j_last) return;                        // (wavefront-uniform)
            const u64 *base = img_mask + (size_t)blk * kTile * col_blocks;
#if defined(FI_NMS_EXP) && (FI_NMS_EXP & 128)
            for (int r = 0; r < R; r += 4) wide_load16(dst[r], off[r], base);    // (timing only: a quarter of the loads)
#else
#pragma unroll
            for (int r = 0; r < R; ++r) wide_load16(dst[r], off[r], base);
#endif
        };
#pragma unroll
        for (int d = 0; d < D; ++d) load_rows(d, buf[d]);
        bool done = false;
        for (int base = 0; !done; base += kWideUnroll) {
#pragma unroll
            for (int d = 0; d < kWideUnroll; ++d) {
                const int it = base + d;
                if (it > 0) {
                    const int j = it - 1;                    // the block whose survivors are known by now
                    const WideSlot w = s_slot[j & 1];
                    if (w.done || it == col_blocks) { done = true; break; }
                    if (j <= j_last) {
                        wide_u32x4(&rows)[R] = buf[(d + D - 1) % D];     // == buf[j % D]: base is a multiple of D
#if !(defined(FI_NMS_EXP) && (FI_NMS_EXP & 1))
                        // loads issued after this block's: the blocks j+1 .. min(j+D-1, j_last)
                        const int younger = min(D - 1, j_last - j);
                        if (younger == D - 1) wide_wait<(D - 1) * R>(rows);
                        else if (D > 2 && younger == 2) wide_wait<2 * R>(rows);
                        else if (D > 2 && younger == 1) wide_wait<R>(rows);
                        else wide_wait<0>(rows);
#endif
                        const unsigned bits = (unsigned)(uniform_u64(w.kept) >> (sub * R));
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            // (columns at or below the diagonal pick up bits too: after their word was published)
                            const unsigned m = 0u - ((bits >> r) & 1u);
                            rem0 |= ((u64)(rows[r].y & m) << 32) | (rows[r].x & m);
                            rem1 |= ((u64)(rows[r].w & m) << 32) | (rows[r].z & m);
                        }
                        if (applier && col0 == it + 1 && rem0) lds_or_b64(&s_rem[(it + 1) & 1], rem0);
                        if (applier && col0 + 1 == it + 1 && rem1) lds_or_b64(&s_rem[(it + 1) & 1], rem1);
                        load_rows(j + D, rows);
                    }
                }
                wide_barrier();
            }
        }
    }
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
        static const bool narrow = getenv("FI_NMS_SCAN_NARROW") != nullptr;       // A/B: the round-1 scan
        const int applier_waves = kWideThreads / kTile - 2;
        if (!narrow && col_blocks <= 8 * applier_waves)                 // 16 row groups x 4 column pairs per wavefront
            hipLaunchKernelGGL(nms_scan_wide_kernel<16>, dim3(batch), dim3(kWideThreads), 0, st, mask, num_boxes,
                               col_blocks, max_keep, keep, num_out);
        else if (!narrow && col_blocks <= 16 * applier_waves)           // 8 row groups x 8 column pairs
            hipLaunchKernelGGL(nms_scan_wide_kernel<8>, dim3(batch), dim3(kWideThreads), 0, st, mask, num_boxes,
                               col_blocks, max_keep, keep, num_out);
        else if (col_blocks <= 256)
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
