// gs_bwd_prep.h -- what the fused blend backward needs before its first wave starts: the (block, segment) work-item
// list from the forward's per-block sweep lengths, the queue head, and a cleared accumulator.  Shared by the stand-alone
// bwd_items_kernel (blend_v2.hip) and by loss_fused_kernel (ssim.hip), which carries the same work along as extra
// blocks when the loss of a fused forward is taken through the library: the serial item scan (one block, ~12 us) and the
// clear then run UNDER the loss kernel instead of between it and the backward.
#pragma once
#include "gs_ctx.h"

namespace gs {

struct BwdPrepArgs {
    int nBlocks;
    const uint32_t* blockWork;
    uint32_t* itemBlock;
    uint32_t itemCap;
    uint32_t* counters;
    uint32_t queueStart;
    float4* clearBuf;        // gradAcc16 as float4s
    size_t clearCount;
};

// one workgroup of blockDim.x = 64 k threads (k <= 16): item list + queue head.  sm: >= 17 words of LDS.
template <int SEG>
__device__ __forceinline__ void bwd_items_scan(const BwdPrepArgs& a, uint32_t* sm)
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
        uint32_t off = c + wbase + incl - v;
        for (uint32_t s = 0; s < v; s++, off++)
            if (off < a.itemCap) a.itemBlock[off] = ((uint32_t)b << 10) | s;   // segment index < 1024
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.counters[GS_CNT_ITEMS] = carry < a.itemCap ? carry : a.itemCap;
        a.counters[GS_CNT_QUEUE] = a.queueStart;      // the waves' first items are their blockIdx.x
    }
}

// workgroup `part` of `parts`: its share of the accumulator clear
__device__ __forceinline__ void bwd_clear_part(const BwdPrepArgs& a, size_t part, size_t parts)
{
    for (size_t i = part * blockDim.x + threadIdx.x; i < a.clearCount; i += parts * blockDim.x)
        a.clearBuf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// blend_v2.hip
void fill_bwd_prep(gs_ctx* c, int N, uint32_t queueStart, BwdPrepArgs& p);

}  // namespace gs
