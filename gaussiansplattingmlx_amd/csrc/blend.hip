// blend.hip -- front-to-back alpha blend of the sorted per-tile splat lists, forward and backward.
//
// Replaces gaussian_tile_global_forward / gaussian_tile_global_backward
// (slang/gaussian_tile_global_kernels.slang:437-614, 648-881).  The reference runs the forward with
// one thread per pixel, every pixel re-reading its tile's whole list from device memory; here a
// workgroup owns one 16x16 pixel block, the block's waves stage the sorted list through LDS in
// coalesced index bursts + 48-B record gathers, every lane blends PPL pixels out of registers, and a
// block leaves as soon as all its pixels have saturated (T < 1e-4).
//   PPL = 4 : one wavefront per tile (64 lanes x 4 px)        -- least LDS traffic, longest critical path
//   PPL = 2 : two wavefronts per tile
//   PPL = 1 : four wavefronts per tile
// Backward walks the list in reverse, rebuilds T by division exactly as the reference's
// undoTileGlobalPixelState does, sums each splat's 11 gradients over the lane's pixels in registers, then
// over the wave with DPP row operations, and adds one 44-B row per (tile, splat) into a 64-B-aligned
// accumulator with native f32 atomics.
// VALU/transcendental-bound (about 24 flop per pixel-splat forward, 70 backward, against 48 B per splat).
#include "gs_ctx.h"
#include "gs_cull.h"
#include "gs_wavesum.h"

namespace gs {

constexpr int TILE = 16;

// Which 16x16 pixel block is work item `blk`, which tile's list does it sweep, and where do its pixels end?
// Tile sizes that are multiples of 16: the blocks of the image grid, row-major (blocksX per row); every block lies in
// one tile.  Any other tile size (the reference app's W/4 x H/4 = 200 x 200, Data/ColmapDataLoader.swift:495-498): the
// blocks are enumerated PER TILE -- bptX x bptY of them, clipped at the tile's right and bottom edge -- so that a
// block never straddles two tiles and the same LDS-staged kernels serve every tile size (the first builds ran these
// sizes one thread per pixel from global memory with per-pixel atomics: 425 ms per backward at 800x800 / 200x200).
struct BlockGeom {
    int W, H, tileW, tileH, gridW, blocksX, bptX, bptY;      // bptX == 0: image-grid enumeration
};
struct BlockRect {
    int tile, x0, y0, xEnd, yEnd;
};
__device__ __forceinline__ BlockRect block_rect(const BlockGeom& g, int blk)
{
    BlockRect r;
    if (g.bptX == 0) {
        const int by = blk / g.blocksX, bx = blk - by * g.blocksX;
        r.x0 = bx * TILE; r.y0 = by * TILE; r.xEnd = g.W; r.yEnd = g.H;
        r.tile = (r.y0 / g.tileH) * g.gridW + r.x0 / g.tileW;
    } else {
        const int per = g.bptX * g.bptY;
        r.tile = blk / per;
        const int rem = blk - r.tile * per, by = rem / g.bptX, bx = rem - by * g.bptX;
        const int ty = r.tile / g.gridW, tx = r.tile - ty * g.gridW;
        r.x0 = tx * g.tileW + bx * TILE; r.y0 = ty * g.tileH + by * TILE;
        r.xEnd = min(g.W, (tx + 1) * g.tileW); r.yEnd = min(g.H, (ty + 1) * g.tileH);
    }
    return r;
}

__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }

// exp(x), x <= 0, for the FORWARD kernels: v_exp_f32 on the rounded product x log2(e) is off by |x log2 e| 2^-24 relative
// (3.4e-4 on rendered colours of magnitude ~27); carrying the product's rounding error and log2(e)'s tail along brings it
// to ~1 ulp at four more instructions (blend_v2.hip, gauss_alpha_raw).
__device__ __forceinline__ float comp_exp(float x)
{
    constexpr float L2E = 1.44269502162933349609375f, L2E_TAIL_LN2 = 1.3349758e-08f, LN2 = 0.69314718055994531f;
    const float hi = x * L2E;
    const float lo = fmaf(x, L2E, -hi);
    const float d = fmaf(x, L2E_TAIL_LN2, lo * LN2);
    const float g = __builtin_amdgcn_exp2f(hi);
    return fmaf(g, d, g);
}

// ---- wave64 sum of 11 values via DPP, hand-placed ------------------------------------------------
// hipcc turns a builtin-DPP butterfly into v_mov_dpp + v_pk_add_f32 pairs (about 200 instructions for 11
// values); written out as v_add_f32_dpp the same reduction is 66 instructions.  The 11 chains are interleaved
// step by step, so every DPP read is at least 11 instructions behind the write it depends on; the leading
// s_nop covers the VALU-write -> DPP-read hazard against whatever produced the inputs.
// After the block, lanes 48..63 hold the wave totals (lane 63 is the one that is read).
#define GS_DPP_STEP(CTRL)                                      \
    "v_add_f32_dpp %0, %0, %0 " CTRL "\n\t"                    \
    "v_add_f32_dpp %1, %1, %1 " CTRL "\n\t"                    \
    "v_add_f32_dpp %2, %2, %2 " CTRL "\n\t"                    \
    "v_add_f32_dpp %3, %3, %3 " CTRL "\n\t"                    \
    "v_add_f32_dpp %4, %4, %4 " CTRL "\n\t"                    \
    "v_add_f32_dpp %5, %5, %5 " CTRL "\n\t"                    \
    "v_add_f32_dpp %6, %6, %6 " CTRL "\n\t"                    \
    "v_add_f32_dpp %7, %7, %7 " CTRL "\n\t"                    \
    "v_add_f32_dpp %8, %8, %8 " CTRL "\n\t"                    \
    "v_add_f32_dpp %9, %9, %9 " CTRL "\n\t"                    \
    "v_add_f32_dpp %10, %10, %10 " CTRL "\n\t"

__device__ __forceinline__ void wave_sum11(float (&v)[11])
{
    asm volatile(
        "s_nop 1\n\t"
        GS_DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
        GS_DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        GS_DPP_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
        GS_DPP_STEP("row_mirror row_mask:0xf bank_mask:0xf")
        GS_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
        GS_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
        "s_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
          "+v"(v[8]), "+v"(v[9]), "+v"(v[10]));
}

// -----------------------------------------------------------------------------------------------
// forward, 16x16 pixel blocks (tile sizes that are multiples of 16)
// -----------------------------------------------------------------------------------------------
template <int PPL>
__global__ __launch_bounds__(256 / PPL) void blend_fwd_kernel(
    BlockGeom geom, int whiteBg, const float4* __restrict__ packed12,
    const uint32_t* __restrict__ sortedIdx, const uint32_t* __restrict__ tileRanges, float* __restrict__ outColor,
    float* __restrict__ outDepth, float* __restrict__ outAlpha, uint32_t* __restrict__ lastContrib,
    const uint32_t* __restrict__ blockOrder)
{
    constexpr int NT = 256 / PPL;
    constexpr int CHUNK = NT;
    __shared__ float4 sg[CHUNK * 3];
    const int tid = threadIdx.x;
    const int blk = (int)blockOrder[blockIdx.x];      // heaviest pixel blocks are dispatched first
    const BlockRect br = block_rect(geom, blk);
    const int W = geom.W, tile = br.tile;
    const uint32_t start = tileRanges[2 * tile], end = tileRanges[2 * tile + 1];
    const uint32_t count = end > start ? end - start : 0u;

    float px[PPL], py[PPL], T[PPL], cr[PPL], cg[PPL], cb[PPL], dd[PPL];
    uint32_t nc[PPL];
    bool inside[PPL], done[PPL];
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        const int p = tid + k * NT;               // pixel index inside the block, row-major
        const int x = br.x0 + (p & 15), y = br.y0 + (p >> 4);
        inside[k] = x < br.xEnd && y < br.yEnd;
        done[k] = !inside[k];
        px[k] = (float)x; py[k] = (float)y;       // integer pixel coordinates (reference :555-556)
        T[k] = 1.0f; cr[k] = cg[k] = cb[k] = dd[k] = 0.0f;
        nc[k] = count;
    }

    for (uint32_t chunk = 0; chunk < count; chunk += CHUNK) {
        bool allDone = true;
#pragma unroll
        for (int k = 0; k < PPL; k++) allDone = allDone && done[k];
        if (__syncthreads_and(allDone)) break;    // also fences the previous chunk's LDS reads
        const uint32_t i = chunk + tid;
        if (i < count) {
            const uint32_t g = sortedIdx[start + i];
            const float4* src = packed12 + (size_t)g * 3;
            sg[tid * 3 + 0] = src[0];
            sg[tid * 3 + 1] = src[1];
            sg[tid * 3 + 2] = src[2];
        }
        __syncthreads();
        const uint32_t m = min((uint32_t)CHUNK, count - chunk);
        if (!allDone) {
            for (uint32_t j = 0; j < m; j++) {
                const float4 a = sg[j * 3], b = sg[j * 3 + 1], c = sg[j * 3 + 2];
                // a: mx my c00 c01 | b: c10 c11 r g | c: b opacity depth pad
#pragma unroll
                for (int k = 0; k < PPL; k++) {
                    if (!done[k]) {
                        const float dx = px[k] - a.x, dy = py[k] - a.y;
                        const float dxdy = dx * dy;
                        const float e = -0.5f * (dx * dx * a.z + dy * dy * b.y + dxdy * a.w + dxdy * b.x);
                        const float raw = comp_exp(e) * c.y;
                        const float alpha = raw > 0.99f ? 0.99f : raw;
                        const float contrib = T[k] * alpha;
                        cr[k] += contrib * b.z; cg[k] += contrib * b.w; cb[k] += contrib * c.x;
                        dd[k] += contrib * c.z;
                        T[k] = T[k] * (1.0f - alpha);
                        if (T[k] < 1e-4f) { nc[k] = chunk + j + 1; done[k] = true; }
                    }
                }
                if (PPL > 1) {   // leave the chunk early once this wave has nothing left to do
                    bool w = true;
#pragma unroll
                    for (int k = 0; k < PPL; k++) w = w && done[k];
                    if (__all(w)) break;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        if (inside[k]) {
            const int p = tid + k * NT;
            const int x = br.x0 + (p & 15), y = br.y0 + (p >> 4);
            const size_t pix = (size_t)y * W + x;
            const float bg = whiteBg ? T[k] : 0.0f;
            outColor[3 * pix] = cr[k] + bg; outColor[3 * pix + 1] = cg[k] + bg; outColor[3 * pix + 2] = cb[k] + bg;
            if (outDepth) outDepth[pix] = dd[k];
            outAlpha[pix] = 1.0f - T[k];
            lastContrib[pix] = nc[k];
        }
    }
}


// -----------------------------------------------------------------------------------------------
// tiles LARGER than a 16x16 block (the reference app builds its renderer with TILE_SIZE = (W/4, H/4): 200x200 at 800x800)
//
// A block sweeps its TILE's list -- 30 k to 54 k entries at 200x200 on the bench scene -- of which only the entries near
// the block can move its pixels: the reference's semantics make every pixel blend every Gaussian of its tile, but a
// Gaussian whose weight exp(-q/2) is below 2^-29 on every pixel of the block moves no state by more than ~5e-8 (the
// fused kernels' staging cull, gs_cull.h).  The kernels above blended all of them (4.5 ms forward, 15.8 ms backward); here
// every thread tests one entry per round against the block's pixel rectangle -- a cheap sufficient test first: q >=
// lambda_min(conic) x distance^2 to the rectangle, then the exact rectangle minimum of the quadratic form --, the kept
// entries are compacted IN LIST ORDER into LDS with their list positions, and the sweep visits only those.  A block then
// pays for scanning its tile's list, not for blending it.  nContrib keeps its meaning (positions in the tile's list).
// -----------------------------------------------------------------------------------------------
__device__ __forceinline__ bool block_reach(const float4& a, const float4& b, float X0, float X1, float Y0, float Y1)
{
    // distance^2 from the mean to the rectangle (0 inside), and the smaller eigenvalue of the conic
    const float ddx = fmaxf(fmaxf(X0, -X1), 0.0f), ddy = fmaxf(fmaxf(Y0, -Y1), 0.0f);
    const float d2 = ddx * ddx + ddy * ddy;
    const float bb = 0.5f * (a.w + b.x), mid = 0.5f * (a.z + b.y), dif = 0.5f * (a.z - b.y);
    const float lmin = mid - sqrtf(dif * dif + bb * bb);
    // (1 % of slack covers the rounding of this bound against rect_min_q's own)
    if (lmin > 0.0f && lmin * d2 > CULL_QMIN * 1.01f) return false;
    return !(rect_min_q(a.z, a.w, b.x, b.y, X0, X1, Y0, Y1) > CULL_QMIN);
}

// ordered block-wide compaction: rank of this thread's kept entry among the kept ones with a LOWER thread id, and the
// total.  waveCnt: NT / 64 words of LDS; contains a barrier.
template <int NT>
__device__ __forceinline__ uint32_t kept_rank(bool keep, uint32_t* waveCnt, uint32_t& total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) waveCnt[wv] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t off = 0; total = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) { const uint32_t c = waveCnt[i]; off += i < wv ? c : 0u; total += c; }
    return off + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

template <int PPL>
__global__ __launch_bounds__(256 / PPL) void blend_fwd_cull_kernel(
    BlockGeom geom, int whiteBg, const float4* __restrict__ packed12,
    const uint32_t* __restrict__ sortedIdx, const uint32_t* __restrict__ tileRanges, float* __restrict__ outColor,
    float* __restrict__ outDepth, float* __restrict__ outAlpha, uint32_t* __restrict__ lastContrib,
    const uint32_t* __restrict__ blockOrder)
{
    constexpr int NT = 256 / PPL;
    __shared__ float4 sg[NT * 3];
    __shared__ uint32_t waveCnt[NT / 64];
    const int tid = threadIdx.x;
    const int blk = (int)blockOrder[blockIdx.x];
    const BlockRect br = block_rect(geom, blk);
    const int W = geom.W, tile = br.tile;
    const uint32_t start = tileRanges[2 * tile], end = tileRanges[2 * tile + 1];
    const uint32_t count = end > start ? end - start : 0u;
    const float rx0 = (float)br.x0, rx1 = (float)(min(br.x0 + TILE, br.xEnd) - 1);
    const float ry0 = (float)br.y0, ry1 = (float)(min(br.y0 + TILE, br.yEnd) - 1);

    float px[PPL], py[PPL], T[PPL], cr[PPL], cg[PPL], cb[PPL], dd[PPL];
    uint32_t nc[PPL];
    bool inside[PPL], done[PPL];
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        const int p = tid + k * NT;
        const int x = br.x0 + (p & 15), y = br.y0 + (p >> 4);
        inside[k] = x < br.xEnd && y < br.yEnd;
        done[k] = !inside[k];
        px[k] = (float)x; py[k] = (float)y;
        T[k] = 1.0f; cr[k] = cg[k] = cb[k] = dd[k] = 0.0f;
        nc[k] = count;
    }
    // records one round ahead, indices two: each is a dependent memory latency
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc = ra;
    if ((uint32_t)tid < count) {
        const float4* src = packed12 + (size_t)sortedIdx[start + tid] * 3;
        ra = src[0]; rb = src[1]; rc = src[2];
    }
    uint32_t gAhead = (uint32_t)(NT + tid) < count ? sortedIdx[start + NT + tid] : 0u;
    for (uint32_t base = 0; base < count; base += NT) {
        bool allDone = true;
#pragma unroll
        for (int k = 0; k < PPL; k++) allDone = allDone && done[k];
        if (__syncthreads_and(allDone)) break;        // also fences the previous round's LDS reads
        const uint32_t i = base + tid;
        const float4 a = ra, b = rb, c = rc;
        const bool keep = i < count && block_reach(a, b, rx0 - a.x, rx1 - a.x, ry0 - a.y, ry1 - a.y);
        if (i + NT < count) {
            const float4* src = packed12 + (size_t)gAhead * 3;
            ra = src[0]; rb = src[1]; rc = src[2];
        }
        gAhead = i + 2u * NT < count ? sortedIdx[start + i + 2u * NT] : 0u;
        uint32_t total;
        const uint32_t pos = kept_rank<NT>(keep, waveCnt, total);
        if (keep) {
            sg[pos * 3 + 0] = a;
            sg[pos * 3 + 1] = b;
            sg[pos * 3 + 2] = make_float4(c.x, c.y, c.z, __uint_as_float(i + 1u));      // list position + 1
        }
        __syncthreads();
        if (!allDone) {
            for (uint32_t j = 0; j < total; j++) {
                const float4 ea = sg[j * 3], eb = sg[j * 3 + 1], ec = sg[j * 3 + 2];
#pragma unroll
                for (int k = 0; k < PPL; k++) {
                    if (!done[k]) {
                        const float dx = px[k] - ea.x, dy = py[k] - ea.y;
                        const float dxdy = dx * dy;
                        const float e = -0.5f * (dx * dx * ea.z + dy * dy * eb.y + dxdy * ea.w + dxdy * eb.x);
                        const float raw = comp_exp(e) * ec.y;
                        const float alpha = raw > 0.99f ? 0.99f : raw;
                        const float contrib = T[k] * alpha;
                        cr[k] += contrib * eb.z; cg[k] += contrib * eb.w; cb[k] += contrib * ec.x;
                        dd[k] += contrib * ec.z;
                        T[k] = T[k] * (1.0f - alpha);
                        if (T[k] < 1e-4f) { nc[k] = __float_as_uint(ec.w); done[k] = true; }
                    }
                }
                if (PPL > 1) {
                    bool w = true;
#pragma unroll
                    for (int k = 0; k < PPL; k++) w = w && done[k];
                    if (__all(w)) break;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        if (inside[k]) {
            const int p = tid + k * NT;
            const int x = br.x0 + (p & 15), y = br.y0 + (p >> 4);
            const size_t pix = (size_t)y * W + x;
            const float bg = whiteBg ? T[k] : 0.0f;
            outColor[3 * pix] = cr[k] + bg; outColor[3 * pix + 1] = cg[k] + bg; outColor[3 * pix + 2] = cb[k] + bg;
            if (outDepth) outDepth[pix] = dd[k];
            outAlpha[pix] = 1.0f - T[k];
            lastContrib[pix] = nc[k];
        }
    }
}

// -----------------------------------------------------------------------------------------------
// backward
// -----------------------------------------------------------------------------------------------
struct PixGrad {
    float v[11];   // mx my c00 c01 c10 c11 r g b opacity depth
};

// one (pixel, splat) step of the reverse sweep; updates T and cotT, accumulates into acc
__device__ __forceinline__ void bwd_step(const float4& a, const float4& b, const float4& c, float px, float py,
                                         float cCx, float cCy, float cCz, float cD, float& T, float& cT,
                                         PixGrad& acc)
{
    const float dx = px - a.x, dy = py - a.y, dxdy = dx * dy;
    const float e = -0.5f * (dx * dx * a.z + dy * dy * b.y + dxdy * a.w + dxdy * b.x);
    const float ex = fast_exp(e);
    const float raw = ex * c.y;
    const float alpha = raw > 0.99f ? 0.99f : raw;
    // undoTileGlobalPixelState (:501-521)
    float denom = 1.0f - alpha;
    if (denom < 1e-6f) denom = 1e-6f;
    const float Tprev = T * __builtin_amdgcn_rcpf(denom);   // v_rcp_f32 (1 ulp): within the 1e-3 gradient bar
    const float contrib = Tprev * alpha;
    // reverse of updateTileGlobalPixelState
    const float S13 = c.z * cD + c.x * cCz + b.w * cCy + b.z * cCx;
    const float dAlpha = -(Tprev * cT) + Tprev * S13;
    cT = (1.0f - alpha) * cT + alpha * S13;
    T = Tprev;
    // reverse of tileGlobalAlphaFromGaussian: the clamp passes iff raw <= 0.99
    const float S32 = raw > 0.99f ? 0.0f : dAlpha;
    const float dE = c.y * S32 * ex;
    const float S36 = -0.5f * dE;
    const float S39 = dy * (b.y * S36);
    const float S41 = dx * (a.z * S36);
    const float S42 = b.x * S36 + a.w * S36;
    acc.v[0] += -(S41 + S41 + dy * S42);
    acc.v[1] += -(S39 + S39 + dx * S42);
    acc.v[2] += dx * dx * S36;
    const float S37 = dxdy * S36;
    acc.v[3] += S37;
    acc.v[4] += S37;
    acc.v[5] += dy * dy * S36;
    acc.v[6] += contrib * cCx;
    acc.v[7] += contrib * cCy;
    acc.v[8] += contrib * cCz;
    acc.v[9] += ex * S32;
    acc.v[10] += contrib * cD;
}

template <int PPL>
__global__ __launch_bounds__(256 / PPL) void blend_bwd_kernel(
    BlockGeom geom, int whiteBg, const float4* __restrict__ packed12,
    const uint32_t* __restrict__ sortedIdx, const uint32_t* __restrict__ tileRanges,
    const float* __restrict__ cotColor, const float* __restrict__ cotDepth, const float* __restrict__ cotAlpha,
    const float* __restrict__ outAlpha, const uint32_t* __restrict__ lastContrib, float* __restrict__ gradAcc16,
    const uint32_t* __restrict__ blockOrder)
{
    constexpr int NT = 256 / PPL;
    constexpr int NW = NT / 64;
    constexpr int CHUNK = 64;
    __shared__ float4 sg[CHUNK * 3];
    __shared__ uint32_t sidx[CHUNK];
    __shared__ float part[NW][CHUNK][12];
    __shared__ uint32_t smax[NW > 1 ? NW : 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int blk = (int)blockOrder[blockIdx.x];
    const BlockRect br = block_rect(geom, blk);
    const int W = geom.W, tile = br.tile;
    const uint32_t start = tileRanges[2 * tile], end = tileRanges[2 * tile + 1];
    const uint32_t count = end > start ? end - start : 0u;
    if (count == 0) return;

    float px[PPL], py[PPL], T[PPL], cT[PPL], cCx[PPL], cCy[PPL], cCz[PPL], cD[PPL];
    uint32_t nc[PPL];
    uint32_t myMax = 0;
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        const int p = tid + k * NT;
        const int x = br.x0 + (p & 15), y = br.y0 + (p >> 4);
        px[k] = (float)x; py[k] = (float)y;
        nc[k] = 0; T[k] = 0.f; cT[k] = 0.f; cCx[k] = cCy[k] = cCz[k] = cD[k] = 0.f;
        if (x < br.xEnd && y < br.yEnd) {
            const size_t pix = (size_t)y * W + x;
            cCx[k] = cotColor[3 * pix]; cCy[k] = cotColor[3 * pix + 1]; cCz[k] = cotColor[3 * pix + 2];
            cD[k] = cotDepth ? cotDepth[pix] : 0.0f;
            const float cA = cotAlpha ? cotAlpha[pix] : 0.0f;
            T[k] = 1.0f - outAlpha[pix];
            cT[k] = -cA + (whiteBg ? (cCx[k] + cCy[k] + cCz[k]) : 0.0f);
            nc[k] = min(lastContrib[pix], count);
        }
        myMax = max(myMax, nc[k]);
    }
    // wave max and block max of nContrib: the sweep starts there, not at the end of the list
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) myMax = max(myMax, (uint32_t)__shfl_xor((int)myMax, d, 64));
    const uint32_t waveMax = myMax;
    uint32_t blockMax = waveMax;
    if (NW > 1) {
        if (lane == 0) smax[wv] = waveMax;
        __syncthreads();
        blockMax = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) blockMax = max(blockMax, smax[i]);
    }
    if (blockMax == 0) return;

    const int numChunks = (int)((blockMax + CHUNK - 1) / CHUNK);
    for (int ch = numChunks - 1; ch >= 0; ch--) {
        const uint32_t chunkStart = (uint32_t)ch * CHUNK;
        const uint32_t m = min((uint32_t)CHUNK, blockMax - chunkStart);
        __syncthreads();     // previous chunk's flush has finished reading part[] / sidx[]
        if ((uint32_t)tid < m) {
            const uint32_t g = sortedIdx[start + chunkStart + tid];
            sidx[tid] = g;
            const float4* src = packed12 + (size_t)g * 3;
            sg[tid * 3 + 0] = src[0];
            sg[tid * 3 + 1] = src[1];
            sg[tid * 3 + 2] = src[2];
        }
        __syncthreads();
        for (int j = (int)m - 1; j >= 0; j--) {
            const uint32_t ii = chunkStart + (uint32_t)j;
            PixGrad acc;
#pragma unroll
            for (int q = 0; q < 11; q++) acc.v[q] = 0.0f;
            if (ii < waveMax) {          // wave-uniform: some lane of this wave still has this splat
                const float4 a = sg[j * 3], b = sg[j * 3 + 1], c = sg[j * 3 + 2];
#pragma unroll
                for (int k = 0; k < PPL; k++)
                    if (ii < nc[k]) bwd_step(a, b, c, px[k], py[k], cCx[k], cCy[k], cCz[k], cD[k], T[k], cT[k], acc);
                wave_sum11(acc.v);
            }
            if (lane == 63) {
                float4* dst = reinterpret_cast<float4*>(&part[wv][j][0]);
                dst[0] = make_float4(acc.v[0], acc.v[1], acc.v[2], acc.v[3]);
                dst[1] = make_float4(acc.v[4], acc.v[5], acc.v[6], acc.v[7]);
                dst[2] = make_float4(acc.v[8], acc.v[9], acc.v[10], 0.0f);
            }
        }
        __syncthreads();
        // flush: one 44-B row per splat of the chunk, summed over the block's waves, f32 atomics
        for (uint32_t e = tid; e < m * 11; e += NT) {
            const uint32_t j = e / 11, q = e - j * 11;
            float v = part[0][j][q];
#pragma unroll
            for (int w2 = 1; w2 < NW; w2++) v += part[w2][j][q];
            if (v != 0.0f) atomicAdd(&gradAcc16[(size_t)sidx[j] * 16 + q], v);
        }
    }
}

// the reverse sweep over a large tile's list (blend_fwd_cull_kernel): NT list positions per round, highest first, the
// kept entries compacted in sweep order; per batch of 64 kept entries one reduction + flush round as in blend_bwd_kernel
template <int PPL>
__global__ __launch_bounds__(256 / PPL) void blend_bwd_cull_kernel(
    BlockGeom geom, int whiteBg, const float4* __restrict__ packed12,
    const uint32_t* __restrict__ sortedIdx, const uint32_t* __restrict__ tileRanges,
    const float* __restrict__ cotColor, const float* __restrict__ cotDepth, const float* __restrict__ cotAlpha,
    const float* __restrict__ outAlpha, const uint32_t* __restrict__ lastContrib, float* __restrict__ gradAcc16,
    const uint32_t* __restrict__ blockOrder)
{
    constexpr int NT = 256 / PPL;
    constexpr int NW = NT / 64;
    constexpr int BATCH = 64;
    __shared__ float4 sg[NT * 3];
    __shared__ uint32_t sidx[NT];
    __shared__ uint32_t spos[NT];
    __shared__ float part[NW][BATCH][12];
    __shared__ uint32_t smax[NW > 1 ? NW : 1];
    __shared__ uint32_t waveCnt[NW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int blk = (int)blockOrder[blockIdx.x];
    const BlockRect br = block_rect(geom, blk);
    const int W = geom.W, tile = br.tile;
    const uint32_t start = tileRanges[2 * tile], end = tileRanges[2 * tile + 1];
    const uint32_t count = end > start ? end - start : 0u;
    if (count == 0) return;
    const float rx0 = (float)br.x0, rx1 = (float)(min(br.x0 + TILE, br.xEnd) - 1);
    const float ry0 = (float)br.y0, ry1 = (float)(min(br.y0 + TILE, br.yEnd) - 1);

    float px[PPL], py[PPL], T[PPL], cT[PPL], cCx[PPL], cCy[PPL], cCz[PPL], cD[PPL];
    uint32_t nc[PPL];
    uint32_t myMax = 0;
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        const int p = tid + k * NT;
        const int x = br.x0 + (p & 15), y = br.y0 + (p >> 4);
        px[k] = (float)x; py[k] = (float)y;
        nc[k] = 0; T[k] = 0.f; cT[k] = 0.f; cCx[k] = cCy[k] = cCz[k] = cD[k] = 0.f;
        if (x < br.xEnd && y < br.yEnd) {
            const size_t pix = (size_t)y * W + x;
            cCx[k] = cotColor[3 * pix]; cCy[k] = cotColor[3 * pix + 1]; cCz[k] = cotColor[3 * pix + 2];
            cD[k] = cotDepth ? cotDepth[pix] : 0.0f;
            const float cA = cotAlpha ? cotAlpha[pix] : 0.0f;
            T[k] = 1.0f - outAlpha[pix];
            cT[k] = -cA + (whiteBg ? (cCx[k] + cCy[k] + cCz[k]) : 0.0f);
            nc[k] = min(lastContrib[pix], count);
        }
        myMax = max(myMax, nc[k]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) myMax = max(myMax, (uint32_t)__shfl_xor((int)myMax, d, 64));
    const uint32_t waveMax = myMax;
    uint32_t blockMax = waveMax;
    if (NW > 1) {
        if (lane == 0) smax[wv] = waveMax;
        __syncthreads();
        blockMax = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) blockMax = max(blockMax, smax[i]);
    }
    if (blockMax == 0) return;

    const int rounds = (int)((blockMax + NT - 1) / NT);
    // thread t of a round takes position base + NT - 1 - t: ascending thread id = descending list position = sweep order
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc = ra;
    uint32_t gCur = 0u, gAhead = 0u;
    {
        const uint32_t i0 = (uint32_t)(rounds - 1) * NT + (uint32_t)(NT - 1 - tid);
        if (i0 < blockMax) {
            gCur = sortedIdx[start + i0];
            const float4* src = packed12 + (size_t)gCur * 3;
            ra = src[0]; rb = src[1]; rc = src[2];
        }
        if (rounds >= 2) gAhead = sortedIdx[start + (uint32_t)(rounds - 2) * NT + (uint32_t)(NT - 1 - tid)];
    }
    for (int r = rounds - 1; r >= 0; r--) {
        const uint32_t i = (uint32_t)r * NT + (uint32_t)(NT - 1 - tid);
        __syncthreads();         // the previous round's flush has finished reading sidx[] / part[]
        const float4 a = ra, b = rb, c = rc;
        const uint32_t g = gCur;
        const bool keep = i < blockMax && block_reach(a, b, rx0 - a.x, rx1 - a.x, ry0 - a.y, ry1 - a.y);
        if (r >= 1) {            // (every position of the rounds below is inside the sweep)
            gCur = gAhead;
            const float4* src = packed12 + (size_t)gCur * 3;
            ra = src[0]; rb = src[1]; rc = src[2];
            gAhead = r >= 2 ? sortedIdx[start + (uint32_t)(r - 2) * NT + (uint32_t)(NT - 1 - tid)] : 0u;
        }
        uint32_t total;
        const uint32_t pos = kept_rank<NT>(keep, waveCnt, total);
        if (keep) {
            sg[pos * 3 + 0] = a; sg[pos * 3 + 1] = b; sg[pos * 3 + 2] = c;
            sidx[pos] = g; spos[pos] = i;
        }
        __syncthreads();
        for (uint32_t batch = 0; batch < total; batch += BATCH) {
            const uint32_t mB = min((uint32_t)BATCH, total - batch);
            if (batch) __syncthreads();          // the previous batch's flush has finished reading part[]
            for (uint32_t e = 0; e < mB; e++) {
                const uint32_t j = batch + e;
                const uint32_t ii = spos[j];
                PixGrad acc;
#pragma unroll
                for (int q = 0; q < 11; q++) acc.v[q] = 0.0f;
                float w = 0.0f;
                if (ii < waveMax) {          // wave-uniform: some lane of this wave still has this splat
                    const float4 ea = sg[j * 3], eb = sg[j * 3 + 1], ec = sg[j * 3 + 2];
#pragma unroll
                    for (int k = 0; k < PPL; k++)
                        if (ii < nc[k]) bwd_step(ea, eb, ec, px[k], py[k], cCx[k], cCy[k], cCz[k], cD[k], T[k], cT[k], acc);
                    // ten sums (dc10 is dc01 again) through the transposed reduction of the fused backward (gs_wavesum.h: 23
                    // instructions against wave_sum11's 66); the totals come out at lanes 4 s of rows 0..2:
                    //   slot s:  0 dmx  1 dc00  2 dmy  3 dc01 | 4 5 dop  6 7 ddepth | 8 dc11  9 dg  10 dr  11 db
                    const float u[10] = {acc.v[0], acc.v[1], acc.v[2], acc.v[3], acc.v[5], acc.v[6], acc.v[7], acc.v[8],
                                         acc.v[9], acc.v[10]};
                    w = wave_sum10_transposed<false>(u);
                }
                if ((lane & 3) == 0 && lane < 48) part[wv][e][lane >> 2] = w;
            }
            __syncthreads();
            for (uint32_t x = tid; x < mB * 11; x += NT) {
                const uint32_t e = x / 11, q = x - e * 11;
                // packed column q (dmx dmy dc00 dc01 dc10 dc11 dr dg db dop ddepth) <- slot
                const uint32_t slot = (0x64b9a833120ull >> (q * 4)) & 15u;       // 0 2 1 3 3 8 10 9 11 4 6
                float v = part[0][e][slot];
#pragma unroll
                for (int w2 = 1; w2 < NW; w2++) v += part[w2][e][slot];
                if (v != 0.0f) atomicAdd(&gradAcc16[(size_t)sidx[batch + e] * 16 + q], v);
            }
        }
    }
}

__global__ void gradacc_to_packed11_kernel(int N, const float* __restrict__ acc16, float* __restrict__ out11)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * 11) return;
    const int g = i / 11, q = i - g * 11;
    out11[i] = acc16[(size_t)g * 16 + q];
}

// -----------------------------------------------------------------------------------------------
// heaviest-first block order.  All ~2500 workgroups of an 800x800 frame are resident at once, so a
// workgroup's CU is fixed at launch and per-CU work is whatever the round-robin deal happens to sum to
// (tile lists run from 0 to thousands of splats).  Dealing the blocks in descending-work order gives every
// CU a stratified sample of the distribution: per-CU sums differ by at most about one heavy tile.
// -----------------------------------------------------------------------------------------------
// forward estimate: length of the block's tile list
__global__ void block_work_counts_kernel(int nBlocks, int blocksX, int tileW, int tileH, int gridW,
                                         const uint32_t* __restrict__ tileRanges, uint32_t* __restrict__ work)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    const int by = b / blocksX, bx = b - by * blocksX;
    const int tile = ((by * TILE) / tileH) * gridW + (bx * TILE) / tileW;
    const uint32_t s = tileRanges[2 * tile], e = tileRanges[2 * tile + 1];
    work[b] = e > s ? e - s : 0u;
}

// backward: exact sweep length = max nContrib over the block's pixels (one wave per block)
__global__ __launch_bounds__(64) void block_work_contrib_kernel(BlockGeom geom,
                                                                const uint32_t* __restrict__ lastContrib,
                                                                uint32_t* __restrict__ work)
{
    const int b = blockIdx.x;
    const BlockRect br = block_rect(geom, b);
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int p = threadIdx.x + k * 64;
        const int x = br.x0 + (p & 15), y = br.y0 + (p >> 4);
        if (x < br.xEnd && y < br.yEnd) m = max(m, lastContrib[(size_t)y * geom.W + x]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if (threadIdx.x == 0) work[b] = m;
}

// single workgroup: approximate heaviest-first order by a 256-bucket counting sort on work / max(work)
// (only the rank strata matter, not the exact order)
__global__ __launch_bounds__(1024) void order_blocks_kernel(int n, const uint32_t* __restrict__ work,
                                                            uint32_t* __restrict__ order)
{
    __shared__ uint32_t bucket[256];
    __shared__ uint32_t wmax;
    if (threadIdx.x < 256) bucket[threadIdx.x] = 0;
    if (threadIdx.x == 0) wmax = 0;
    __syncthreads();
    uint32_t m = 0;
    for (int i = threadIdx.x; i < n; i += 1024) m = max(m, work[i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&wmax, m);
    __syncthreads();
    const float scale = 255.0f / (float)(wmax + 1u);
    // bucket 0 = heaviest
    for (int i = threadIdx.x; i < n; i += 1024) atomicAdd(&bucket[255 - (int)((float)work[i] * scale)], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {   // exclusive scan of 256 counts by one wave (4 per lane)
        uint32_t c[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { c[k] = bucket[threadIdx.x * 4 + k]; sum += c[k]; }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d, 64);
            if ((int)threadIdx.x >= d) incl += t;
        }
        uint32_t run = incl - sum;
#pragma unroll
        for (int k = 0; k < 4; k++) { bucket[threadIdx.x * 4 + k] = run; run += c[k]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        const uint32_t pos = atomicAdd(&bucket[255 - (int)((float)work[i] * scale)], 1u);
        order[pos] = (uint32_t)i;
    }
}

__global__ void order_identity_kernel(int n, uint32_t* __restrict__ order)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[i] = (uint32_t)i;
}

static BlockGeom op_geom(const gs_ctx* c)
{
    BlockGeom g;
    g.W = c->W; g.H = c->H; g.tileW = c->tileW; g.tileH = c->tileH; g.gridW = c->gridW; g.blocksX = gs_div_up(c->W, TILE);
    g.bptX = c->fast16 ? 0 : gs_div_up(c->tileW, TILE);
    g.bptY = c->fast16 ? 0 : gs_div_up(c->tileH, TILE);
    return g;
}

static int launch_block_order(gs_ctx* c, const uint32_t* work)
{
    const int n = c->opBlocks;
    hipLaunchKernelGGL(order_blocks_kernel, dim3(1), dim3(1024), 0, c->stream, n, work, c->blockOrder);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

// -----------------------------------------------------------------------------------------------
// launchers
// -----------------------------------------------------------------------------------------------
int launch_blend_forward(gs_ctx* c, float* outColor, float* outDepth, float* outAlpha, uint32_t* lastContrib)
{
    const float4* p12 = reinterpret_cast<const float4*>(c->packed12);
    const BlockGeom geom = op_geom(c);
    const int nBlocks = c->opBlocks;
    // no reordering here: the list length says little about where a saturating tile stops
    // (correlation 0.07 with the measured sweep length on the bench scene), so the natural order stays
    hipLaunchKernelGGL(order_identity_kernel, dim3(gs_div_up(nBlocks, 256)), dim3(256), 0, c->stream, nBlocks,
                       c->blockOrder);
    const dim3 grid(nBlocks);
#define GS_FWD(P)                                                                                                  \
    hipLaunchKernelGGL(blend_fwd_kernel<P>, grid, dim3(256 / P), 0, c->stream, geom, c->whiteBg, p12, c->sortedIdx, \
                       c->tileRanges, outColor, outDepth, outAlpha, lastContrib, c->blockOrder)
#define GS_FWDC(P)                                                                                                      \
    hipLaunchKernelGGL(blend_fwd_cull_kernel<P>, grid, dim3(256 / P), 0, c->stream, geom, c->whiteBg, p12, c->sortedIdx, \
                       c->tileRanges, outColor, outDepth, outAlpha, lastContrib, c->blockOrder)
    if (c->tileW > TILE || c->tileH > TILE) {       // a tile is more than one block: scan, cull, sweep the rest
        if (c->opFwdPpl == 4) GS_FWDC(4);
        else if (c->opFwdPpl == 2) GS_FWDC(2);
        else GS_FWDC(1);
    } else if (c->opFwdPpl == 4) GS_FWD(4);
    else if (c->opFwdPpl == 2) GS_FWD(2);
    else GS_FWD(1);
#undef GS_FWDC
#undef GS_FWD
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

// accumulates d(packed) into ctx->gradAcc16 (zeroed here)
int launch_blend_backward(gs_ctx* c, int N, const float* cotColor, const float* cotDepth, const float* cotAlpha,
                          const float* outAlpha, const uint32_t* lastContrib)
{
    GS_HIP_CHECK(c, hipMemsetAsync(c->gradAcc16, 0, sizeof(float) * 16 * (size_t)N, c->stream));
    const float4* p12 = reinterpret_cast<const float4*>(c->packed12);
    const BlockGeom geom = op_geom(c);
    const int nBlocks = c->opBlocks;
    // (the ctx's own scratch: a caller's view-hint buffer holds one word per block of the IMAGE grid, which a tile size
    // that is not a multiple of 16 exceeds)
    uint32_t* work = c->fast16 ? c->blockWork : c->blockWorkOwn;
    hipLaunchKernelGGL(block_work_contrib_kernel, dim3(nBlocks), dim3(64), 0, c->stream, geom, lastContrib, work);
    const int rc = launch_block_order(c, work);
    if (rc) return rc;
    const dim3 grid(nBlocks);
#define GS_BWD(P)                                                                                                  \
    hipLaunchKernelGGL(blend_bwd_kernel<P>, grid, dim3(256 / P), 0, c->stream, geom, c->whiteBg, p12, c->sortedIdx, \
                       c->tileRanges, cotColor, cotDepth, cotAlpha, outAlpha, lastContrib, c->gradAcc16, c->blockOrder)
#define GS_BWDC(P)                                                                                                      \
    hipLaunchKernelGGL(blend_bwd_cull_kernel<P>, grid, dim3(256 / P), 0, c->stream, geom, c->whiteBg, p12, c->sortedIdx, \
                       c->tileRanges, cotColor, cotDepth, cotAlpha, outAlpha, lastContrib, c->gradAcc16, c->blockOrder)
    if (c->tileW > TILE || c->tileH > TILE) {       // (two pixels per lane unless the knob says four: 1.54 -> 1.44 ms at 200x200)
        if (c->opBwdPpl == 4) GS_BWDC(4);
        else GS_BWDC(2);
    } else if (c->opBwdPpl == 4) GS_BWD(4);
    else if (c->opBwdPpl == 2) GS_BWD(2);
    else GS_BWD(1);
#undef GS_BWDC
#undef GS_BWD
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_gradacc_to_packed11(gs_ctx* c, int N, float* gradPacked11)
{
    if (N == 0) return GS_OK;
    hipLaunchKernelGGL(gradacc_to_packed11_kernel, dim3(gs_div_up((long long)N * 11, 256)), dim3(256), 0, c->stream,
                       N, c->gradAcc16, gradPacked11);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

}  // namespace gs
