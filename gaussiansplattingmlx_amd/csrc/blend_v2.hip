// blend_v2.hip -- the fused path's alpha-blend kernels (gs_render_forward / gs_render_backward).
//
// Same arithmetic as blend.hip (which keeps serving the op-level entry points), different mapping:
//
//  * Every wavefront is autonomous: it gathers 64 records of its list at a time, one per lane (a coalesced
//    index burst + three 16-B loads per lane), keeps them in 12 VGPRs, and broadcasts record j to SGPRs with
//    v_readlane_b32; the record then rides as the scalar operand of the VALU ops.  No LDS staging, no workgroup
//    barriers, and none of the LDS broadcast reads that bounded the LDS version (12-24 LDS cycles per
//    wave-iteration on an LDS shared by four SIMDs).  The next 64 records are in flight while the current 64
//    are blended, so one memory latency is paid per 64 splats, off the critical path.  (Scalar loads of the
//    records were tried first: two dependent scalar-cache misses per splat group left the VALU 80 % idle.)
//  * Forward: one wavefront per 8x8 pixel quarter of a 16x16 block, four splats per trip, branch-free
//    termination (a finished pixel keeps blending with weight zero), wave-uniform exit.  Every SEG splats the
//    running state (T, C, D) is saved per pixel.
//  * Backward: the saved states make a tile's list SEGMENT-parallel.  The work items are (pixel block,
//    segment) pairs of at most SEG splats each, pulled from a device-side queue by persistent single-wave
//    workgroups: the heaviest tile no longer sets the kernel time.  Inside an item the sweep runs FORWARD
//    (T by multiplication, as the forward pass), with the cotangent of T in closed form,
//        c_i = (sum_{j>i} T_j a_j S_j + T_n cT_n) / T_{i+1},
//    the sum being (final colour - running colour) . cotColour.  Each lane owns 4 pixels, so the per-splat
//    wave reduction (10 sums, DPP) is paid once per 256 pixel-splats instead of once per 64.
//  * The reference rebuilds T from the rounded output alpha (T_n' = 1 - outAlpha) and divides its way back;
//    every T of a pixel is therefore off by the factor s = T_n' / T_n.  The same factor is applied here, so
//    the gradients match the reference's arithmetic, not just the exact calculus.
//
// Compiled with -ffp-contract=off and explicit fmaf so that forward and backward evaluate the running sums
// bit-identically (the closed form needs the backward's recomputed colour to meet the forward's).
#include "gs_ctx.h"

namespace gs {

constexpr int BLK = 16;

struct Rec {
    float mx, my, c00, c01, c10, c11, r, g, b, op, depth;
};

// one record per lane (lane j of a chunk holds splat chunkStart + j of the list)
struct RecV {
    float4 a, b, c;
};

__device__ __forceinline__ RecV load_chunk(const float4* __restrict__ packed12, const uint32_t* __restrict__ idx,
                                           uint32_t i0, uint32_t iEnd, int lane)
{
    RecV v;
    v.a = v.b = v.c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i0 + lane < iEnd) {
        const float4* p = packed12 + (size_t)idx[i0 + lane] * 3;
        v.a = p[0]; v.b = p[1]; v.c = p[2];
    }
    return v;
}

__device__ __forceinline__ float rl(float x, int j)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), j));
}

// broadcast lane j's record to scalar registers
__device__ __forceinline__ Rec bcast(const RecV& v, int j)
{
    Rec r;
    r.mx = rl(v.a.x, j); r.my = rl(v.a.y, j); r.c00 = rl(v.a.z, j); r.c01 = rl(v.a.w, j);
    r.c10 = rl(v.b.x, j); r.c11 = rl(v.b.y, j); r.r = rl(v.b.z, j); r.g = rl(v.b.w, j);
    r.b = rl(v.c.x, j); r.op = rl(v.c.y, j); r.depth = rl(v.c.z, j);
    return r;
}

// exponent in the reference's operation order (tileGlobalAlphaFromGaussian, kernels.slang:450-455; this file is
// compiled without FMA contraction).  Elongated splats make the four terms cancel by 2-3 orders of magnitude,
// so a re-factored exponent (pre-scaled conic, fused multiply-adds) is equally accurate but decorrelates its
// rounding from the reference's: 1.7e-4 L-inf on colours of magnitude 10-100.  Mirroring the order keeps the
// two renders within 1e-4.
__device__ __forceinline__ float splat_raw(const Rec& s, float px, float py, float& dx, float& dy, float& G)
{
    dx = px - s.mx; dy = py - s.my;
    const float dxdy = dx * dy;
    const float q = dx * dx * s.c00 + dy * dy * s.c11 + dxdy * s.c01 + dxdy * s.c10;
    // exp(-0.5 q) = 2^(q * (-0.5 log2 e)): the factor -0.5 is a power of two, so folding it into the constant
    // rounds exactly like (-0.5 q) * log2 e
    G = __builtin_amdgcn_exp2f(q * -0.72134752044448170368f);
    return s.op * G;
}

// ---------------------------------------------------------------------------------------------
// segment bookkeeping
// ---------------------------------------------------------------------------------------------
// segBase[b] = index of pixel block b's first saved state (state before splat SEG); exclusive scan of
// max(ceil(count/SEG) - 1, 0).  One workgroup.
template <int SEG>
__global__ __launch_bounds__(1024) void seg_base_kernel(int nBlocks, int blocksX, int tileW, int tileH, int gridW,
                                                        const uint32_t* __restrict__ tileRanges,
                                                        uint32_t* __restrict__ segBase, uint32_t* __restrict__ blockWork)
{
    __shared__ uint32_t sm[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int base = 0; base < nBlocks; base += 1024) {
        const int b = base + threadIdx.x;
        uint32_t v = 0;
        if (b < nBlocks) {
            const int by = b / blocksX, bx = b - by * blocksX;
            const int tile = ((by * BLK) / tileH) * gridW + (bx * BLK) / tileW;
            const uint32_t s = tileRanges[2 * tile], e = tileRanges[2 * tile + 1];
            const uint32_t cnt = e > s ? e - s : 0u;
            v = cnt > SEG ? (cnt + SEG - 1) / SEG - 1 : 0u;
            blockWork[b] = 0;
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
        for (int i = 0; i < 16; i++) { const uint32_t s = sm[i]; if (i < w) wbase += s; tot += s; }
        const uint32_t c = carry;
        if (b < nBlocks) segBase[b] = c + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
}

// work items of the backward: (block, segment) for segment < ceil(blockWork/SEG).  One workgroup.
template <int SEG>
__global__ __launch_bounds__(1024) void bwd_items_kernel(int nBlocks, const uint32_t* __restrict__ blockWork,
                                                         uint32_t* __restrict__ itemBlock, uint32_t itemCap,
                                                         uint32_t* __restrict__ counters)
{
    __shared__ uint32_t sm[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int base = 0; base < nBlocks; base += 1024) {
        const int b = base + threadIdx.x;
        const uint32_t v = b < nBlocks ? min((blockWork[b] + SEG - 1) / SEG, 1024u) : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) sm[w] = incl;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
        for (int i = 0; i < 16; i++) { const uint32_t s = sm[i]; if (i < w) wbase += s; tot += s; }
        const uint32_t c = carry;
        uint32_t off = c + wbase + incl - v;
        for (uint32_t s = 0; s < v; s++, off++)
            if (off < itemCap) itemBlock[off] = ((uint32_t)b << 10) | s;   // segment index < 1024
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counters[GS_CNT_ITEMS] = carry < itemCap ? carry : itemCap;
        counters[GS_CNT_QUEUE] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int SEG>
__global__ __launch_bounds__(256) void blend_fwd_v2_kernel(
    int W, int H, int tileW, int tileH, int gridW, int blocksX, int whiteBg, const float4* __restrict__ rec12,
    const uint32_t* __restrict__ sortedIdx, const uint32_t* __restrict__ tileRanges,
    const uint32_t* __restrict__ segBase, uint32_t segCap, float* __restrict__ outColor, float* __restrict__ outDepth,
    float* __restrict__ outAlpha, uint32_t* __restrict__ lastContrib, float* __restrict__ finalT,
    float* __restrict__ segState, uint32_t* __restrict__ blockWork)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x;
    const int by = b / blocksX, bx = b - by * blocksX;
    const int tile = ((by * BLK) / tileH) * gridW + (bx * BLK) / tileW;
    const uint32_t start = __builtin_amdgcn_readfirstlane(tileRanges[2 * tile]);
    const uint32_t end = __builtin_amdgcn_readfirstlane(tileRanges[2 * tile + 1]);
    const uint32_t count = end > start ? end - start : 0u;
    const uint32_t sbase = __builtin_amdgcn_readfirstlane(segBase[b]);

    const int x = bx * BLK + (wv & 1) * 8 + (lane & 7), y = by * BLK + (wv >> 1) * 8 + (lane >> 3);
    const bool inside = x < W && y < H;
    const float px = (float)x, py = (float)y;
    float T = 1.0f, Tact = inside ? 1.0f : 0.0f, cr = 0.f, cg = 0.f, cb = 0.f, dd = 0.f;
    bool active = inside;
    uint32_t nc = inside ? count : 0u;

    const uint32_t* __restrict__ idx = sortedIdx + start;
    static_assert(SEG % 64 == 0, "segment length must be a multiple of the 64-record chunk");
    auto save_state = [&](uint32_t i) {
        const uint32_t slot = sbase + i / SEG - 1;
        if (slot < segCap) {
            float* st = segState + (size_t)slot * (5 * 256) + wv * 64 + lane;
            st[0] = T; st[256] = cr; st[512] = cg; st[768] = cb; st[1024] = dd;
        }
    };
    // branch-free per-splat update: a finished pixel keeps "blending" with weight zero
    auto step = [&](const Rec& s, uint32_t i) {
        float dx, dy, G;
        const float raw = splat_raw(s, px, py, dx, dy, G);
        const float alpha = raw > 0.99f ? 0.99f : raw;
        const float w = Tact * alpha;
        cr = fmaf(w, s.r, cr); cg = fmaf(w, s.g, cg); cb = fmaf(w, s.b, cb); dd = fmaf(w, s.depth, dd);
        const float Tn = Tact * (1.0f - alpha);
        const bool fin = active && (Tn < 1e-4f);
        T = active ? Tn : T;
        nc = fin ? (i + 1) : nc;
        active = active && !fin;
        Tact = active ? Tn : 0.0f;
    };
    const float4* __restrict__ p12 = rec12;
    RecV nxt = load_chunk(p12, idx, 0, count, lane);
    for (uint32_t c0 = 0; c0 < count; c0 += 64) {
        const RecV cur = nxt;
        if (c0 + 64 < count) nxt = load_chunk(p12, idx, c0 + 64, count, lane);   // in flight during this chunk
        if (c0 != 0 && (c0 % SEG) == 0) save_state(c0);                          // SEG is a multiple of 64
        const uint32_t n = min(64u, count - c0);
        bool live = true;
        uint32_t j = 0;
        for (; j + 4 <= n; j += 4) {
            const Rec r0 = bcast(cur, j), r1 = bcast(cur, j + 1), r2 = bcast(cur, j + 2), r3 = bcast(cur, j + 3);
            step(r0, c0 + j); step(r1, c0 + j + 1); step(r2, c0 + j + 2); step(r3, c0 + j + 3);
            if (!__any(active)) { live = false; break; }
        }
        if (!live) break;
        for (; j < n; j++) step(bcast(cur, j), c0 + j);
        if (!__any(active)) break;
    }
    if (inside) {
        const size_t pix = (size_t)y * W + x;
        const float bg = whiteBg ? T : 0.0f;
        outColor[3 * pix] = cr + bg; outColor[3 * pix + 1] = cg + bg; outColor[3 * pix + 2] = cb + bg;
        outDepth[pix] = dd;
        outAlpha[pix] = 1.0f - T;
        lastContrib[pix] = nc;
        finalT[pix] = T;
    }
    // sweep length of this block for the backward's work items: max nContrib over its pixels
    uint32_t m = nc;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if (lane == 0 && m) atomicMax(&blockWork[b], m);
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
#define GS2_DPP_STEP(CTRL)                                     \
    "v_add_f32_dpp %0, %0, %0 " CTRL "\n\t"                    \
    "v_add_f32_dpp %1, %1, %1 " CTRL "\n\t"                    \
    "v_add_f32_dpp %2, %2, %2 " CTRL "\n\t"                    \
    "v_add_f32_dpp %3, %3, %3 " CTRL "\n\t"                    \
    "v_add_f32_dpp %4, %4, %4 " CTRL "\n\t"                    \
    "v_add_f32_dpp %5, %5, %5 " CTRL "\n\t"                    \
    "v_add_f32_dpp %6, %6, %6 " CTRL "\n\t"                    \
    "v_add_f32_dpp %7, %7, %7 " CTRL "\n\t"                    \
    "v_add_f32_dpp %8, %8, %8 " CTRL "\n\t"                    \
    "v_add_f32_dpp %9, %9, %9 " CTRL "\n\t"

// wave64 sums of 10 values; totals valid in lanes 48..63.  The 10 chains are interleaved step by step, so
// each DPP read is 10 instructions behind the write it depends on; the leading s_nop covers the hazard
// against the producers of the inputs.
__device__ __forceinline__ void wave_sum10(float (&v)[10])
{
    asm volatile(
        "s_nop 1\n\t"
        GS2_DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
        GS2_DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        GS2_DPP_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
        GS2_DPP_STEP("row_mirror row_mask:0xf bank_mask:0xf")
        GS2_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
        GS2_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
        "s_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
          "+v"(v[8]), "+v"(v[9]));
}

// 10 sums per splat: dmx dmy dc00 dc01(=dc10) dc11 dop dr dg db ddepth; flushed into the reference's packed
// row order (dmx dmy dc00 dc01 dc10 dc11 dr dg db dop ddepth) of gradAcc16
template <int SEG>
__global__ __launch_bounds__(64) void blend_bwd_v2_kernel(
    int W, int H, int tileW, int tileH, int gridW, int blocksX, int whiteBg, const float4* __restrict__ rec12,
    const uint32_t* __restrict__ sortedIdx, const uint32_t* __restrict__ tileRanges,
    const uint32_t* __restrict__ segBase, uint32_t segCap, const uint32_t* __restrict__ blockWork,
    const uint32_t* __restrict__ itemBlock, uint32_t* __restrict__ counters, const float* __restrict__ cotColor,
    const float* __restrict__ cotDepth, const float* __restrict__ cotAlpha, const float* __restrict__ outColor,
    const float* __restrict__ outDepth, const float* __restrict__ outAlpha, const uint32_t* __restrict__ lastContrib,
    const float* __restrict__ finalT, const float* __restrict__ segState, float* __restrict__ gradAcc16)
{
    __shared__ float part[SEG][12];
    const int lane = threadIdx.x;
    const uint32_t nItems = __builtin_amdgcn_readfirstlane(counters[GS_CNT_ITEMS]);
    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(&counters[GS_CNT_QUEUE], 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= nItems) break;     // the queue only grows: every wave reaches this exit
        const uint32_t packed = __builtin_amdgcn_readfirstlane(itemBlock[item]);
        const int b = (int)(packed >> 10);
        const uint32_t seg = packed & 1023u;
        const int by = b / blocksX, bx = b - by * blocksX;
        const int tile = ((by * BLK) / tileH) * gridW + (bx * BLK) / tileW;
        const uint32_t start = __builtin_amdgcn_readfirstlane(tileRanges[2 * tile]);
        const uint32_t work = __builtin_amdgcn_readfirstlane(blockWork[b]);
        const uint32_t i0 = seg * SEG, i1 = min(i0 + SEG, work);
        const uint32_t slot = __builtin_amdgcn_readfirstlane(segBase[b]) + seg - 1;

        float px[4], py[4], T[4], cr[4], cg[4], cb[4], dd[4];
        float cCx[4], cCy[4], cCz[4], cD[4], Cfx[4], Cfy[4], Cfz[4], Df[4], tail[4], sc[4];
        uint32_t nc[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = bx * BLK + (k & 1) * 8 + (lane & 7), y = by * BLK + (k >> 1) * 8 + (lane >> 3);
            px[k] = (float)x; py[k] = (float)y;
            nc[k] = 0; T[k] = 1.0f; cr[k] = cg[k] = cb[k] = dd[k] = 0.0f;
            cCx[k] = cCy[k] = cCz[k] = cD[k] = Cfx[k] = Cfy[k] = Cfz[k] = Df[k] = tail[k] = sc[k] = 0.0f;
            if (x < W && y < H) {
                const size_t pix = (size_t)y * W + x;
                const uint32_t n = lastContrib[pix];
                if (n > i0) {
                    nc[k] = n;
                    cCx[k] = cotColor[3 * pix]; cCy[k] = cotColor[3 * pix + 1]; cCz[k] = cotColor[3 * pix + 2];
                    cD[k] = cotDepth ? cotDepth[pix] : 0.0f;
                    const float cA = cotAlpha ? cotAlpha[pix] : 0.0f;
                    const float Tn = finalT[pix];
                    const float bg = whiteBg ? Tn : 0.0f;
                    Cfx[k] = outColor[3 * pix] - bg; Cfy[k] = outColor[3 * pix + 1] - bg;
                    Cfz[k] = outColor[3 * pix + 2] - bg;
                    Df[k] = outDepth[pix];
                    const float cTn = -cA + (whiteBg ? (cCx[k] + cCy[k] + cCz[k]) : 0.0f);
                    tail[k] = Tn * cTn;
                    sc[k] = (1.0f - outAlpha[pix]) / Tn;     // the reference's T = 1 - outAlpha anchor
                    if (seg != 0 && slot < segCap) {
                        const float* st = segState + (size_t)slot * (5 * 256) + k * 64 + lane;
                        T[k] = st[0]; cr[k] = st[256]; cg[k] = st[512]; cb[k] = st[768]; dd[k] = st[1024];
                    }
                }
            }
        }

        const uint32_t* __restrict__ idx = sortedIdx + start;
        RecV cur = load_chunk(rec12, idx, i0, i1, lane);
        for (uint32_t i = i0; i < i1; i++) {
            const uint32_t jl = (i - i0) & 63u;
            if (jl == 0 && i != i0) cur = load_chunk(rec12, idx, i, i1, lane);
            const Rec s = bcast(cur, (int)jl);
            float acc[10];
#pragma unroll
            for (int q = 0; q < 10; q++) acc[q] = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (i < nc[k]) {
                    float dx, dy, G;
                    const float raw = splat_raw(s, px[k], py[k], dx, dy, G);
                    const float alpha = raw > 0.99f ? 0.99f : raw;
                    const float w = T[k] * alpha;
                    cr[k] = fmaf(w, s.r, cr[k]); cg[k] = fmaf(w, s.g, cg[k]); cb[k] = fmaf(w, s.b, cb[k]);
                    dd[k] = fmaf(w, s.depth, dd[k]);
                    const float Tn = T[k] * (1.0f - alpha);
                    // cotangent of T_{i+1}: what the rest of the list and the background still owe
                    float rem = fmaf(cCx[k], Cfx[k] - cr[k], tail[k]);
                    rem = fmaf(cCy[k], Cfy[k] - cg[k], rem);
                    rem = fmaf(cCz[k], Cfz[k] - cb[k], rem);
                    rem = fmaf(cD[k], Df[k] - dd[k], rem);
                    const float c = rem * __builtin_amdgcn_rcpf(Tn);   // v_rcp_f32, 1 ulp
                    const float S = fmaf(cCx[k], s.r, fmaf(cCy[k], s.g, fmaf(cCz[k], s.b, cD[k] * s.depth)));
                    const float Ts = sc[k] * T[k];
                    const float dAlpha = Ts * (S - c);
                    const float contrib = Ts * alpha;
                    const float gate = raw > 0.99f ? 0.0f : dAlpha;
                    const float h = -0.5f * (gate * raw);           // d/d(exponent) times -1/2
                    const float hx = dx * (s.c00 * h), hy = dy * (s.c11 * h), hc = s.c10 * h + s.c01 * h;
                    acc[0] -= hx + hx + dy * hc;
                    acc[1] -= hy + hy + dx * hc;
                    acc[2] = fmaf(dx * dx, h, acc[2]);
                    acc[3] = fmaf(dx * dy, h, acc[3]);
                    acc[4] = fmaf(dy * dy, h, acc[4]);
                    acc[5] = fmaf(G, gate, acc[5]);
                    acc[6] = fmaf(contrib, cCx[k], acc[6]);
                    acc[7] = fmaf(contrib, cCy[k], acc[7]);
                    acc[8] = fmaf(contrib, cCz[k], acc[8]);
                    acc[9] = fmaf(contrib, cD[k], acc[9]);
                    T[k] = Tn;
                }
            }
            wave_sum10(acc);
            if (lane == 63) {
                float4* dst = reinterpret_cast<float4*>(&part[i - i0][0]);
                dst[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                dst[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
                dst[2] = make_float4(acc[8], acc[9], 0.0f, 0.0f);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): lane 63's LDS writes have landed (single wave)
        __builtin_amdgcn_wave_barrier();
        const uint32_t n = i1 - i0;
        for (uint32_t e = lane; e < n * 11; e += 64) {
            const uint32_t j = e / 11, q = e - j * 11;
            // packed column q <- reduced slot: 0 1 2 3 3 4 6 7 8 5 9
            const uint32_t src = q < 4 ? q : (q < 6 ? q - 1 : (q < 9 ? q : (q == 9 ? 5u : 9u)));
            const float v = part[j][src];
            if (v != 0.0f) atomicAdd(&gradAcc16[(size_t)idx[i0 + j] * 16 + q], v);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
constexpr int SEGLEN = GS_SEG_LEN;

int launch_blend_forward_v2(gs_ctx* c, float* outColor, float* outDepth, float* outAlpha)
{
    const int blocksX = gs_div_up(c->W, BLK), nBlocks = c->numPixBlocks;
    hipLaunchKernelGGL(seg_base_kernel<SEGLEN>, dim3(1), dim3(1024), 0, c->stream, nBlocks, blocksX, c->tileW, c->tileH,
                       c->gridW, c->tileRanges, c->segBase, c->blockWork);
    hipLaunchKernelGGL(blend_fwd_v2_kernel<SEGLEN>, dim3(nBlocks), dim3(256), 0, c->stream, c->W, c->H, c->tileW,
                       c->tileH, c->gridW, blocksX, c->whiteBg, reinterpret_cast<const float4*>(c->packed12), c->sortedIdx,
                       c->tileRanges, c->segBase, (uint32_t)c->segCap, outColor, outDepth, outAlpha, c->lastContrib,
                       c->finalT, c->segState, c->blockWork);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_blend_backward_v2(gs_ctx* c, int N, const float* cotColor, const float* cotDepth, const float* cotAlpha,
                             const float* outColor, const float* outDepth, const float* outAlpha)
{
    GS_HIP_CHECK(c, hipMemsetAsync(c->gradAcc16, 0, sizeof(float) * 16 * (size_t)N, c->stream));
    const int blocksX = gs_div_up(c->W, BLK), nBlocks = c->numPixBlocks;
    hipLaunchKernelGGL(bwd_items_kernel<SEGLEN>, dim3(1), dim3(1024), 0, c->stream, nBlocks, c->blockWork, c->itemBlock,
                       (uint32_t)c->itemCap, c->counters);
    int grid = c->numCUs * 16;
    if ((long long)grid > c->itemCap) grid = (int)c->itemCap;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(blend_bwd_v2_kernel<SEGLEN>, dim3(grid), dim3(64), 0, c->stream, c->W, c->H, c->tileW, c->tileH,
                       c->gridW, blocksX, c->whiteBg, reinterpret_cast<const float4*>(c->packed12), c->sortedIdx,
                       c->tileRanges, c->segBase, (uint32_t)c->segCap, c->blockWork, c->itemBlock, c->counters, cotColor,
                       cotDepth, cotAlpha, outColor, outDepth, outAlpha, c->lastContrib, c->finalT, c->segState,
                       c->gradAcc16);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

}  // namespace gs
