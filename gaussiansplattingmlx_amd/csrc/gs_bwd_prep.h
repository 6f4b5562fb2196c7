// gs_bwd_prep.h -- what the fused blend backward needs before its first wave starts: the (block, segment) work-item
// list from the forward's per-block sweep lengths, the queue head, and a cleared accumulator.  Shared by the stand-alone
// bwd_items_kernel (blend_v2.hip) and by loss_fused_kernel (ssim.hip), which carries the same work along as extra
// blocks when the loss of a fused forward is taken through the library: the serial item scan (one block, ~12 us) and the
// clear then run UNDER the loss kernel instead of between it and the backward.
#pragma once
#include "gs_ctx.h"

#ifndef GS_ITEMS_OWN
#define GS_ITEMS_OWN 12u     // deepest block (in segments) still written by its own thread (bwd_items_scan)
#endif

namespace gs {

struct BwdPrepArgs {
    int nBlocks;
    const uint32_t* blockWork;
    uint32_t* itemBlock;
    uint32_t* itemRow;       // per item: its row of the checkpoint-slot table (segBase[block] + segment - 1)
    const uint32_t* segBase;
    uint32_t itemCap;
    uint32_t* counters;
    uint32_t queueStart;
    uint32_t* bwdQueue;      // [8][32]: the fused backward's work-queue heads, one per XCD (a cache line each)
    float4* clearBuf;        // gradAcc16 as float4s
    size_t clearCount;
    // renewal of the view's depth cuts (binning.hip), one tile per thread; cutStore == nullptr: none kept
    uint32_t* cutStore;
    int cutsInForce;
    const uint32_t* tileRanges;
    const uint32_t* sortedIdx;
    uint32_t idxMask;
    const float* rec12;
};

// The cut of a tile whose sweep stopped `work` entries into a list of `len` is the depth key of entry 2 work + 128 if the
// list goes on beyond that (on the bench scene sweeps grow by up to 1.85x between two visits of a view; with
// work + work/4 + 64 a handful of the 2500 tiles missed in nearly every forward); a cut that was in force this forward
// and still leaves 1.5 work + 64 entries is kept; anything else means "bin everything next time".
__device__ __forceinline__ void bwd_cut_renew(const BwdPrepArgs& a, int b)
{
    if (!a.cutStore || b >= a.nBlocks) return;
    const uint32_t work = a.blockWork[b];
    const uint32_t s0 = a.tileRanges[2 * b], e0 = a.tileRanges[2 * b + 1];
    const uint32_t len = e0 > s0 ? e0 - s0 : 0u;
    const uint32_t cur = a.cutsInForce ? a.cutStore[b] : 0u;
    const uint32_t pm = 2u * work + 128u;
    uint32_t nxt = 0u;
    if (work < len && pm + 1u < len) {
        const uint32_t gg = a.sortedIdx[s0 + pm] & a.idxMask;
        nxt = 0xFFFFFFFFu - __float_as_uint(a.rec12[(size_t)gg * 12 + 10]);      // the entry's depth key
    } else if (cur != 0u && work + work / 2u + 64u <= len) nxt = cur;
    a.cutStore[b] = nxt;
}

// GS_ITEM_PARTS workgroups of blockDim.x = 64 k threads (k <= 16): item list + queue head.  Every one of them scans all the
// sweep lengths (2500 words: cheap) and writes the items of every eighth block -- lane l of each wave belongs to part
// l mod 8 --, so the write loops, which are what takes time, run eight abreast.  sm: >= 17 words of LDS.
constexpr int GS_ITEM_PARTS = 8;
template <int SEG>
__device__ __forceinline__ void bwd_items_scan(const BwdPrepArgs& a, uint32_t* sm, int part)
{
    uint32_t& carry = sm[16];
    const int nT = (int)blockDim.x, nW = nT >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // the next chunk's sweep lengths are requested before the current chunk is scanned (each load is a ~2 us miss)
    uint32_t wNext = (int)threadIdx.x < a.nBlocks ? a.blockWork[threadIdx.x] : 0u;
    for (int base = 0; base < a.nBlocks; base += nT) {
        const int b = base + (int)threadIdx.x;
        const uint32_t work = wNext;
        const uint32_t v = b < a.nBlocks ? min((work + SEG - 1) / SEG, 1024u) : 0u;
        wNext = b + nT < a.nBlocks ? a.blockWork[b + nT] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) sm[w] = incl;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
        for (int i = 0; i < nW; i++) { const uint32_t s = sm[i]; if (i < w) wbase += s; tot += s; }
        const uint32_t c = carry;
        const uint32_t off = c + wbase + incl - v;
        // (the item's row of the checkpoint-slot table rides along, so the backward needs no look-up of the block's first
        // row: item -> row -> the four quadrants' slot ids -> state, as many dependent loads as item -> first row -> state was)
        const uint32_t sb = v > 1 ? a.segBase[b] : 0u;
        // A block of up to 12 segments is written by its own thread, a deeper one by its whole wave, consecutive segments
        // from consecutive lanes (coalesced).  Every thread of ONE workgroup writing its own v items, 4 B at a time and a
        // stride apart from its neighbours', was 0.12 ms once a densified scene had 100 k items (40 per block) -- more than
        // the loss kernel this rides in.  The wave loop costs ~450 cycles per block whatever its depth (its stores queue
        // behind the accumulator clear's), hence the eight workgroups.  Loss stage at N = 1 M (tools/stages_at_n.py, one
        // box): one workgroup 0.134 ms (own thread) / 0.080 (wave loop; 0.066 instead of 0.050 at 300 k); eight workgroups
        // 0.067 (own thread) / 0.056 (wave loop) / 0.055 (both, as here); 0.050 at 300 k for all three.
        constexpr uint32_t OWN = GS_ITEMS_OWN;
        const bool mine = (lane & (GS_ITEM_PARTS - 1)) == part;
        if (mine && v <= OWN)
            for (uint32_t s = 0; s < v; s++)
                if (off + s < a.itemCap) {
                    a.itemBlock[off + s] = ((uint32_t)b << 10) | s;   // segment index < 1024
                    a.itemRow[off + s] = s > 0 ? sb + s - 1u : 0u;
                }
        for (uint64_t left = __ballot(mine && v > OWN); left; left &= left - 1) {
            const int src = __builtin_ctzll(left);
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)off, src);
            const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)v, src);
            const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)sb, src);
            const uint32_t bb = (uint32_t)(b - lane + src);
            for (uint32_t s = (uint32_t)lane; s < n; s += 64u)
                if (o + s < a.itemCap) {
                    a.itemBlock[o + s] = (bb << 10) | s;
                    a.itemRow[o + s] = s > 0 ? r + s - 1u : 0u;
                }
        }
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && part == 0) {
        a.counters[GS_CNT_ITEMS] = carry < a.itemCap ? carry : a.itemCap;
        a.counters[GS_CNT_QUEUE] = a.queueStart;      // (rounds 1-3: the one queue behind the waves' first items)
    }
    if (part == 0 && threadIdx.x < 8) a.bwdQueue[threadIdx.x * 32] = 0u;      // blend_bwd_v2_kernel: pops count from the static share on
}

// workgroup `part` of `parts`: its share of the accumulator clear
__device__ __forceinline__ void bwd_clear_part(const BwdPrepArgs& a, size_t part, size_t parts)
{
    for (size_t i = part * blockDim.x + threadIdx.x; i < a.clearCount; i += parts * blockDim.x)
        a.clearBuf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- the fused blend FORWARD's bookkeeping (one workgroup): segBase[b] = index of pixel block b's first saved state
// (exclusive scan of max(ceil(count / SEG) - 1, 0)), the launch order of its work items, the queue head, the per-block
// sweep lengths cleared.  Runs as seg_base_kernel (blend_v2.hip) or, after the one-pass tile sort, as a spare workgroup
// of wide_scatter_kernel (binning.hip), where the per-tile pair counts are already at hand.
struct SegBaseArgs {
    int nBlocks, blocksX, tileW, tileH, gridW;
    const uint32_t* tileRanges;     // [T,2], or nullptr: take the count of a tile from tileTotal
    const uint32_t* tileTotal;      // [T]
    uint32_t* segBase;
    uint32_t* blockWork;
    uint32_t* counters;
    const uint32_t* workHint;
    uint32_t* blockOrder;
    uint32_t queueStart;
    uint32_t* fwdQueue;             // [8][32]: the fused forward's eight work-queue heads, one per XCD (a cache line each)
    int nq;                         // queues in use (GS_TUNE_FWD_QUEUES)
    int spatial;                    // 1: queue x gets the x-th stripe of the image; 0: the blocks dealt to the queues in launch order
};

// Launch order of the forward's items, round 4: position p of the order belongs to queue / XCD p mod nq (blend_v2.hip).
// Default (spatial = 0): ONE list of all blocks, deepest first, dealt to the queues round-robin -- every XCD gets the same mix
// of deep and shallow blocks.  spatial = 1: queue x is given the blocks of the x-th STRIPE of the image -- the blocks
// [x per, (x + 1) per) in row-major order, per = ceil(nBlocks / nq) --, deepest first inside the stripe: neighbouring blocks
// share most of their records, and a stripe's Gaussians stay in its XCD's L2 (FETCH_SIZE of the forward 53 -> 18 MB per
// launch on the bench scene) -- but stripes of equal block count are not stripes of equal work, and stealing only starts when
// a whole XCD has run dry: same time on the bench scene, +25 % on the grown one (blend forward 0.90 -> 1.13 ms).  Kept for A/B.
// Positions a short last stripe leaves over hold 0xFFFFFFFF (no block).
// lds: 16 + 8 * 256 + 2 words
constexpr int GS_SEGBASE_LDS = 16 + 8 * 256 + 2;
template <int SEG>
__device__ __forceinline__ void seg_base_body(const SegBaseArgs& a, uint32_t* lds)
{
    uint32_t* sm = lds;
    uint32_t* bucket = lds + 16;              // [nq][256]
    uint32_t& carry = lds[16 + 8 * 256];
    uint32_t& wmax = lds[16 + 8 * 256 + 1];
    const int nT = (int)blockDim.x, nW = nT >> 6;
    // (not spatial: ONE stripe, sorted deepest first, whose positions go to the queues round-robin -- position p to queue p mod nq)
    const int nq = a.spatial ? a.nq : 1, per = (a.nBlocks + nq - 1) / nq;
    // every persistent wave's first item is fixed by its blockIdx.x (no pop: thousands of simultaneous pops on one counter
    // take ~6 ns each to resolve); the queues proper start behind those
    if (threadIdx.x == 0) { carry = 0; a.counters[GS_CNT_QUEUE_FWD] = a.queueStart; wmax = 0; }
    if (threadIdx.x < 8) a.fwdQueue[threadIdx.x * 32] = 0u;         // (blend_fwd_v2q_kernel: pops count from the static rows on)
    for (int i = threadIdx.x; i < nq * 256; i += nT) bucket[i] = 0;
    for (int i = threadIdx.x; i < a.nBlocks + 8; i += nT) a.blockOrder[i] = 0xFFFFFFFFu;
    __syncthreads();
    // Inside a stripe: the forward's time is set by its longest serial lists (where a block stops is not predictable from
    // its list length), so when the caller supplies the sweep lengths a previous forward of this view measured, the deepest
    // blocks start first: 256-bucket counting sort per stripe, heaviest bucket first.
    if (a.workHint) {
        uint32_t m = 0;
        for (int i = threadIdx.x; i < a.nBlocks; i += nT) m = max(m, a.workHint[i]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(&wmax, m);
        __syncthreads();
        const float scale = 255.0f / (float)(wmax + 1u);
        for (int i = threadIdx.x; i < a.nBlocks; i += nT)
            atomicAdd(&bucket[(i / per) * 256 + 255 - (int)((float)a.workHint[i] * scale)], 1u);
        __syncthreads();
        for (int st = (int)(threadIdx.x >> 6); st < nq; st += nW) {   // exclusive scan of a stripe's 256 counts by one wave (4 per lane)
            const int l = threadIdx.x & 63;
            uint32_t* bk = bucket + st * 256;
            uint32_t c[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) { c[k] = bk[l * 4 + k]; sum += c[k]; }
            uint32_t incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d, 64);
                if (l >= d) incl += t;
            }
            uint32_t run = incl - sum;
#pragma unroll
            for (int k = 0; k < 4; k++) { bk[l * 4 + k] = run; run += c[k]; }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < a.nBlocks; i += nT) {
            const int st = i / per;
            const uint32_t rank = atomicAdd(&bucket[st * 256 + 255 - (int)((float)a.workHint[i] * scale)], 1u);
            a.blockOrder[(uint32_t)nq * rank + (uint32_t)st] = (uint32_t)i;
        }
    } else {
        for (int i = threadIdx.x; i < a.nBlocks; i += nT) {
            const int st = i / per;
            a.blockOrder[nq * (i - st * per) + st] = (uint32_t)i;
        }
    }
    __syncthreads();          // the hint may BE the sweep-length buffer cleared below: every read of it is done
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int base = 0; base < a.nBlocks; base += nT) {
        const int b = base + (int)threadIdx.x;
        uint32_t v = 0;
        if (b < a.nBlocks) {
            const int by = b / a.blocksX, bx = b - by * a.blocksX;
            const int tile = ((by * 16) / a.tileH) * a.gridW + (bx * 16) / a.tileW;
            uint32_t cnt;
            if (a.tileRanges) {
                const uint32_t s = a.tileRanges[2 * tile], e = a.tileRanges[2 * tile + 1];
                cnt = e > s ? e - s : 0u;
            } else cnt = a.tileTotal[tile];
            v = cnt > SEG ? (cnt + SEG - 1) / SEG - 1 : 0u;
            a.blockWork[b] = 0;
        }
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) sm[w] = incl;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
        for (int i = 0; i < nW; i++) { const uint32_t s = sm[i]; if (i < w) wbase += s; tot += s; }
        const uint32_t c = carry;
        if (b < a.nBlocks) a.segBase[b] = c + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
}

// blend_v2.hip
void fill_seg_base(gs_ctx* c, SegBaseArgs& a);
void fill_bwd_prep(gs_ctx* c, int N, uint32_t queueStart, BwdPrepArgs& p);

}  // namespace gs
