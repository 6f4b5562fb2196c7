// projection.hip -- 3D->2D projection, EWA covariance, SH->RGB: forward + backward.
// Compiled with -ffp-contract=off (see gs_math.h).
//
// Two flavours of each direction:
//  * op-level kernels mirror the reference's custom-function boundary
//    (gaussian_projection_screen_fused_forward/backward,
//    slang/gaussian_projection_kernels.slang:36-173, 205-398): activated inputs,
//    8 output tensors / 5 gradient tensors;
//  * fused kernels start from the six RAW parameter tensors, fold the
//    activations (GaussianRenderer.swift:936-963) and the packing (:85-99) in,
//    and emit what binning needs (tile rectangle, tiles touched, depth key), so
//    the per-Gaussian data crosses HBM once.
// One lane per Gaussian; HBM-bound (408 B/Gaussian forward, 728 B backward at K=25).
#include "gs_ctx.h"
#include "gs_rider.h"

namespace gs {

constexpr int PROJ_THREADS = 256;

CamParams make_cam(const gs_camera* cam, int W, int H)
{
    CamParams p;
    for (int i = 0; i < 16; i++) { p.V[i] = cam->view[i]; p.P[i] = cam->proj[i]; }
    for (int i = 0; i < 3; i++) p.cam[i] = cam->cam_center[i];
    p.fovX = cam->fov_x; p.fovY = cam->fov_y; p.focalX = cam->focal_x; p.focalY = cam->focal_y;
    p.limX = tanf(cam->fov_x * 0.5f) * 1.3f;
    p.limY = tanf(cam->fov_y * 0.5f) * 1.3f;
    p.W = (float)W; p.H = (float)H;
    return p;
}

// ---------------------------------------------------------------------------------------------
// op-level forward
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PROJ_THREADS) void proj_fwd_op_kernel(
    int N, int K, int degree, CamParams cam, const float* __restrict__ scales, const float* __restrict__ rot,
    const float* __restrict__ means3d, const float* __restrict__ shs, float* __restrict__ means2d,
    float* __restrict__ depths, float* __restrict__ color, float* __restrict__ cov2d, float* __restrict__ conic,
    float* __restrict__ radii, float* __restrict__ rectMin, float* __restrict__ rectMax)
{
    const int p = blockIdx.x * PROJ_THREADS + threadIdx.x;
    if (p >= N) return;
    const float m[3] = {means3d[3 * p], means3d[3 * p + 1], means3d[3 * p + 2]};
    const float s[3] = {scales[3 * p], scales[3 * p + 1], scales[3 * p + 2]};
    const float q[4] = {rot[4 * p], rot[4 * p + 1], rot[4 * p + 2], rot[4 * p + 3]};
    ProjOut o;
    project_geometry(m, s, q, cam, o);

    const float x = m[0] - cam.cam[0], y = m[1] - cam.cam[1], z = m[2] - cam.cam[2];
    const float* sh = shs + (size_t)p * K * 3;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    sh_foreach(degree, x, y, z, [&](int k, float b, float, float, float) {
        if (k == 0) { c0 = b * sh[0]; c1 = b * sh[1]; c2 = b * sh[2]; }
        else { c0 += b * sh[k * 3]; c1 += b * sh[k * 3 + 1]; c2 += b * sh[k * 3 + 2]; }
    });
    c0 += 0.5f; c1 += 0.5f; c2 += 0.5f;
    color[3 * p] = c0 > 0.f ? c0 : 0.f;
    color[3 * p + 1] = c1 > 0.f ? c1 : 0.f;
    color[3 * p + 2] = c2 > 0.f ? c2 : 0.f;

    means2d[2 * p] = o.sx; means2d[2 * p + 1] = o.sy;
    depths[p] = o.depth;
#pragma unroll
    for (int k = 0; k < 4; k++) { cov2d[4 * p + k] = o.cov2d[k]; conic[4 * p + k] = o.conic[k]; }
    radii[p] = o.radius;
    rectMin[2 * p] = o.rect[0]; rectMin[2 * p + 1] = o.rect[1];
    rectMax[2 * p] = o.rect[2]; rectMax[2 * p + 1] = o.rect[3];
}

// ---------------------------------------------------------------------------------------------
// colour reverse mode shared by both backward kernels.  SHLOAD(k, ch) reads a coefficient,
// SHSTORE(k, ch, v) writes its gradient.  Returns d(x,y,z).
// ---------------------------------------------------------------------------------------------
template <class Load, class Store>
__device__ __forceinline__ void color_backward(int degree, int K, float x, float y, float z, const float cotCol[3],
                                               Load&& shload, Store&& shstore, float dxyz[3], float* mgOut = nullptr)
{
    float acc[3] = {0.f, 0.f, 0.f};
    sh_foreach(degree, x, y, z, [&](int k, float b, float, float, float) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) acc[ch] = (k == 0) ? b * shload(k, ch) : acc[ch] + b * shload(k, ch);
    });
    float mg[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) mg[ch] = d_max_left(acc[ch] + 0.5f, 0.0f, cotCol[ch]);
    if (mgOut) { mgOut[0] = mg[0]; mgOut[1] = mg[1]; mgOut[2] = mg[2]; }
    float dx = 0.f, dy = 0.f, dz = 0.f;
    int written = 0;
    sh_foreach(degree, x, y, z, [&](int k, float b, float gx, float gy, float gz) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float coef = shload(k, ch);      // read first: the fused kernel stores the gradient in place
            shstore(k, ch, b * mg[ch]);
            const float w = coef * mg[ch];
            dx += gx * w; dy += gy * w; dz += gz * w;
        }
        written = k + 1;
    });
    for (int k = written; k < K; k++)
        for (int ch = 0; ch < 3; ch++) shstore(k, ch, 0.0f);
    dxyz[0] = dx; dxyz[1] = dy; dxyz[2] = dz;
}

// ---------------------------------------------------------------------------------------------
// op-level backward
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PROJ_THREADS) void proj_bwd_op_kernel(
    int N, int K, int degree, CamParams cam, const float* __restrict__ scales, const float* __restrict__ rot,
    const float* __restrict__ means3d, const float* __restrict__ shs, const float* __restrict__ cotDepths,
    const float* __restrict__ cotMeans2d, const float* __restrict__ cotCov2d, const float* __restrict__ cotColor,
    const float* __restrict__ cotConic, float* __restrict__ gScales, float* __restrict__ gRot,
    float* __restrict__ gMeans, float* __restrict__ gShs, float* __restrict__ gCam)
{
    const int p = blockIdx.x * PROJ_THREADS + threadIdx.x;
    if (p >= N) return;
    const float m[3] = {means3d[3 * p], means3d[3 * p + 1], means3d[3 * p + 2]};
    const float s[3] = {scales[3 * p], scales[3 * p + 1], scales[3 * p + 2]};
    const float q[4] = {rot[4 * p], rot[4 * p + 1], rot[4 * p + 2], rot[4 * p + 3]};
    const float cm[2] = {cotMeans2d[2 * p], cotMeans2d[2 * p + 1]};
    const float ccov[4] = {cotCov2d[4 * p], cotCov2d[4 * p + 1], cotCov2d[4 * p + 2], cotCov2d[4 * p + 3]};
    const float ccon[4] = {cotConic[4 * p], cotConic[4 * p + 1], cotConic[4 * p + 2], cotConic[4 * p + 3]};
    GeomGrads g;
    project_geometry_bwd(m, s, q, cam, cm, cotDepths[p], ccov, ccon, g);

    const float x = m[0] - cam.cam[0], y = m[1] - cam.cam[1], z = m[2] - cam.cam[2];
    const float* sh = shs + (size_t)p * K * 3;
    float* gsh = gShs + (size_t)p * K * 3;
    const float cc[3] = {cotColor[3 * p], cotColor[3 * p + 1], cotColor[3 * p + 2]};
    float d[3];
    color_backward(degree, K, x, y, z, cc, [&](int k, int ch) { return sh[k * 3 + ch]; },
                   [&](int k, int ch, float v) { gsh[k * 3 + ch] = v; }, d);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        gMeans[3 * p + a] = g.dm[a] + d[a];
        gCam[3 * p + a] = -d[a];
        gScales[3 * p + a] = g.ds[a];
    }
#pragma unroll
    for (int a = 0; a < 4; a++) gRot[4 * p + a] = g.dq[a];
}

// ---------------------------------------------------------------------------------------------
// SH-rest staging.  features_rest is [N][K-1][3]: 288 B per Gaussian at K = 25.  Read straight from a
// lane-per-Gaussian kernel every load instruction touches 64 different cache lines; instead each wavefront
// moves its 64 rows (one contiguous 18 KB span) through LDS with fully coalesced dwordx4 accesses and then
// reads / writes its own row at stride L+1 words (odd, so bank-conflict free).
// ---------------------------------------------------------------------------------------------
constexpr int PROJ_FUSED_THREADS = 128;   // 2 waves x 64 rows x 73 words = 37 KB of LDS per workgroup at K = 25

// float4 per lane of one wave's rows at K = 25: 64 rows x 72 floats / 64 lanes / 4
constexpr int SH_ROWS_MAX4 = 18;
#ifndef GS_PROJ_LATE_MOMENTS
#define GS_PROJ_LATE_MOMENTS 1
#endif

// The loads of a whole batch are issued before the first LDS store, unconditionally and from clamped addresses: as a
// plain loop (`v = g4[e]; store to LDS`) every iteration waited for its own load -- 18 memory latencies in a row at
// the head of every wave of an HBM-bound kernel.  `keep` returns the batch to the caller (the fused Adam update needs
// the parameter values again after their LDS rows have been overwritten with gradients).
__device__ __forceinline__ void sh_rows_load4(const float* __restrict__ g, int total4, int e0, int lane,
                                              float4 (&regs)[SH_ROWS_MAX4])
{
    const float4* g4 = reinterpret_cast<const float4*>(g);     // wave spans start 16-B aligned when L % 4 == 0
#pragma unroll
    for (int i = 0; i < SH_ROWS_MAX4; i++) regs[i] = g4[min(e0 + lane + 64 * i, total4 - 1)];
}
__device__ __forceinline__ void sh_rows_to_lds4(float* __restrict__ lds, int total4, int L, int e0, int lane,
                                                const float4 (&regs)[SH_ROWS_MAX4])
{
#pragma unroll
    for (int i = 0; i < SH_ROWS_MAX4; i++) {
        const int e = e0 + lane + 64 * i;
        if (e < total4) {
            const int r = (e * 4) / L, c = e * 4 - r * L;
            float* d = lds + r * (L + 1) + c;
            d[0] = regs[i].x; d[1] = regs[i].y; d[2] = regs[i].z; d[3] = regs[i].w;
        }
    }
}

__device__ __forceinline__ void sh_rows_in(float* __restrict__ lds, const float* __restrict__ g, int rows, int L,
                                           int lane)
{
    const int total = rows * L;
    if ((L & 3) == 0) {
        const int total4 = total >> 2;
        for (int e0 = 0; e0 < total4; e0 += 64 * SH_ROWS_MAX4) {
            float4 regs[SH_ROWS_MAX4];
            sh_rows_load4(g, total4, e0, lane, regs);
            sh_rows_to_lds4(lds, total4, L, e0, lane, regs);
        }
    } else {
        for (int e = lane; e < total; e += 64) {
            const int r = e / L, c = e - r * L;
            lds[r * (L + 1) + c] = g[e];
        }
    }
}

// (half-row staging for the fused forward: sh_half_load / sh_half_to_lds, gs_rider.h)
__device__ __forceinline__ void sh_rows_out(const float* __restrict__ lds, float* __restrict__ g, int rows, int L,
                                            int lane)
{
    const int total = rows * L;
    if ((L & 3) == 0) {
        float4* g4 = reinterpret_cast<float4*>(g);
        for (int e = lane; e * 4 < total; e += 64) {
            const int r = (e * 4) / L, c = e * 4 - r * L;
            const float* d = lds + r * (L + 1) + c;
            g4[e] = make_float4(d[0], d[1], d[2], d[3]);
        }
    } else {
        for (int e = lane; e < total; e += 64) {
            const int r = e / L, c = e - r * L;
            g[e] = lds[r * (L + 1) + c];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fused forward: raw parameters -> packed12 + binning inputs
// ---------------------------------------------------------------------------------------------
// COLOUR = false: geometry only -- the colour floats of the packed record are left to the riders of the binning kernels
// (gs_rider.h), no SH row is touched here, no LDS.
// SELF (with COLOUR = false): the wave computes its own Gaussians' colours right behind the geometry, as a rider unit would
// (colour_rider_wave: SH rows of Gaussians that touch no tile are not fetched) -- for the sizes where no binning kernel has
// room for riders and a large part of the Gaussians is outside the view (the 2 M garden scene: 48 %).
#ifndef GS_DEGENERATE_INVISIBLE
#define GS_DEGENERATE_INVISIBLE 1
#endif
template <bool TWO_PHASE, bool COLOUR, bool SELF = false>
__global__ __launch_bounds__(PROJ_FUSED_THREADS) void proj_fwd_fused_kernel(
    int N, int K, int degree, CamParams cam, int tileW, int tileH, int gridW, int gridH,
    const float* __restrict__ xyz, const float* __restrict__ fdc, const float* __restrict__ frest,
    const float* __restrict__ scalesRaw, const float* __restrict__ rotRaw, const float* __restrict__ opacityRaw,
    float* __restrict__ packed12, float* __restrict__ radiiOut, ushort4* __restrict__ tileRect,
    uint32_t* __restrict__ tilesTouched, uint32_t* __restrict__ depthKey, uint32_t* __restrict__ depthVal,
    uint32_t* __restrict__ visPerBlock, uint32_t* __restrict__ counters, int flags, ColourRider self,
    GsVirtGeom vg, GsCutCoarse cc, uint4* __restrict__ tilePieces)
{
    const int noKeyForUntouched = flags & 1;
    const bool trimRects = (flags & 2) && !vg.nbx && tileW == 16 && tileH == 16;
    extern __shared__ float shLds[];
    __shared__ uint32_t sDropped;
    if (cc.superCut) {           // (uniform)
        if (threadIdx.x == 0) sDropped = 0u;
        __syncthreads();
    }
    // first kernel of a forward: clears the ctx counters for the kernels behind it (no memset launch)
    if (blockIdx.x == 0 && threadIdx.x < GS_CNT_COUNT) counters[threadIdx.x] = 0;
    const int p = blockIdx.x * PROJ_FUSED_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int L = (K - 1) * 3;
    // K = 25 (L = 72): the rows go through LDS in two halves of 12 coefficients (see sh_half_load); otherwise whole
    const bool twoPhase = TWO_PHASE;
    const int LH = L >> 1, kSplit = 1 + (K - 1) / 2;          // second half starts at coefficient kSplit
    const int rowW = twoPhase ? LH + 1 : L + 1;
    float* myRows = shLds + wv * 64 * rowW;
    const int row0 = blockIdx.x * PROJ_FUSED_THREADS + wv * 64;
    const int rows = min(64, N - row0);
    float4 halfB[SH_HALF_MAX4];
    if (COLOUR && rows > 0 && L > 0) {
        if (twoPhase) {
            float4 halfA[SH_HALF_MAX4];
            sh_half_load(frest + (size_t)row0 * L, rows, L, 0, LH, lane, halfA);
            sh_half_load(frest + (size_t)row0 * L, rows, L, LH, LH, lane, halfB);
            sh_half_to_lds(myRows, rows, LH, lane, halfA);
        } else {
            sh_rows_in(myRows, frest + (size_t)row0 * L, rows, L, lane);
        }
    }
    // each wave reads back only what it staged itself: DS operations of one wave complete in order
    bool visible = false;
    uint32_t myTouched = 0;       // (SELF)
    ProjOut o;
    float opacity = 0.f, colA[3] = {0.f, 0.f, 0.f}, dirv[3] = {0.f, 0.f, 0.f};
    if (p < N) {
        const float m[3] = {xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2]};
        const float s[3] = {expf(scalesRaw[3 * p]), expf(scalesRaw[3 * p + 1]), expf(scalesRaw[3 * p + 2])};
        const float r0 = rotRaw[4 * p], r1 = rotRaw[4 * p + 1], r2 = rotRaw[4 * p + 2], r3 = rotRaw[4 * p + 3];
        const float den = sqrtf(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3) + 1e-8f;
        const float q[4] = {r0 / den, r1 / den, r2 / den, r3 / den};
        opacity = 1.0f / (1.0f + expf(-opacityRaw[p]));
        project_geometry(m, s, q, cam, o);
        // A 2-D covariance whose float32 determinant is not positive -- the true one always is (the +0.3 blur), so this is
        // cancellation: a needle tens of thousands of pixels long, cov2d = (4.2e7, -3.8e7; -3.8e7, 3.5e7) -- has a conic that is not
        // positive definite: q < 0 without bound, "weights" above 1, a 0.99-alpha blob over its whole 3-sigma square.  The reference's
        // training kernels have no guard (slang/gaussian_projection_screen_shared.slang:248-254; its viewer's shaders do:
        // Metal/GaussianRender.metal:153-154), and there the splat's gradients are 0 x inf = NaN, which takes it out of the picture for
        // good.  Here it is out of the picture while it is degenerate: radius 0, not binned, zero gradient (the fused path's second
        // deliberate deviation; the op-level gs_projection_forward keeps the 1:1 arithmetic).
        if (GS_DEGENERATE_INVISIBLE && !(o.cov2d[0] * o.cov2d[3] - o.cov2d[1] * o.cov2d[2] > 0.0f)) o.radius = 0.0f;

        const float x = m[0] - cam.cam[0], y = m[1] - cam.cam[1], z = m[2] - cam.cam[2];
        const float* d0 = fdc + (size_t)p * 3;
        const float* rest = myRows + lane * rowW;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (!COLOUR) {
        } else if (!twoPhase) {
            sh_foreach(degree, x, y, z, [&](int k, float b, float, float, float) {
                if (k == 0) { c0 = b * d0[0]; c1 = b * d0[1]; c2 = b * d0[2]; }
                else {
                    const float* r = rest + (k - 1) * 3;
                    c0 += b * r[0]; c1 += b * r[1]; c2 += b * r[2];
                }
            });
        } else {
            // same sum, same order (k ascending); coefficients 1 .. kSplit-1 are staged now
            sh_foreach(degree, x, y, z, [&](int k, float b, float, float, float) {
                if (k == 0) { c0 = b * d0[0]; c1 = b * d0[1]; c2 = b * d0[2]; }
                else if (k < kSplit) {
                    const float* r = rest + (k - 1) * 3;
                    c0 += b * r[0]; c1 += b * r[1]; c2 += b * r[2];
                }
            });
        }
        colA[0] = c0; colA[1] = c1; colA[2] = c2;
        dirv[0] = x; dirv[1] = y; dirv[2] = z;
        }
        if (COLOUR && twoPhase && rows > 0 && L > 0) sh_half_to_lds(myRows, rows, LH, lane, halfB);     // the wave's first half is consumed
        if (p < N) {
        float c0 = colA[0], c1 = colA[1], c2 = colA[2];
        if (COLOUR && twoPhase) {
            const float* rest = myRows + lane * rowW;
            sh_foreach(degree, dirv[0], dirv[1], dirv[2], [&](int k, float b, float, float, float) {
                if (k >= kSplit) {
                    const float* r = rest + (k - kSplit) * 3;
                    c0 += b * r[0]; c1 += b * r[1]; c2 += b * r[2];
                }
            });
        }
        // 12th float: per channel, which side of the max(., 0) the colour fell on (2 bits: 0 below, 1 tie, 2 above), so
        // that the colour cotangent can be gated right after the blend backward (color_cot_kernel)
        uint32_t gate = 0u;
        if (COLOUR) {
            c0 += 0.5f; c1 += 0.5f; c2 += 0.5f;
            gate = colour_gate(c0, c1, c2);
        }

        float4* out = reinterpret_cast<float4*>(packed12 + (size_t)p * 12);
        out[0] = make_float4(o.sx, o.sy, o.conic[0], o.conic[1]);
        out[1] = make_float4(o.conic[2], o.conic[3], c0, c1);
        out[2] = make_float4(c2, opacity, o.depth, __uint_as_float(gate));
        if (radiiOut) radiiOut[p] = o.radius;

        uint32_t touched = 0;
        ushort4 tr = make_ushort4(0, 0, 0, 0);
        uint32_t pc[4] = {0u, 0u, 0u, 0u};      // (trimmed rects: the rect's four row groups, first column | columns << 16)
        if (o.radius > 0.0f) {
            int x0, y0, x1, y1;
            if (vg.nbx)        // block lists: tileW .. gridH describe the grid of 16 x 16 blocks enumerated per tile
                block_rect_of_splat(o.rect, o.sx, o.sy, o.cov2d[0], o.cov2d[3], vg.nbx, vg.nby, vg.tw, vg.th, gridW / vg.nbx,
                                    gridH / vg.nby, (int)cam.W, (int)cam.H, x0, y0, x1, y1);
            else if (trimRects)   // 16 x 16 tiles, GS_TUNE_TRIM_RECTS: the reference's 3-sigma square cut by the box of q <= 40.3, beyond
                                  // which the blend's staging drops the entry for every quadrant anyway (block_rect_of_splat)
                block_rect_of_splat(o.rect, o.sx, o.sy, o.cov2d[0], o.cov2d[3], 1, 1, 16, 16, gridW, gridH, (int)cam.W, (int)cam.H,
                                    x0, y0, x1, y1);
            else
                tile_rect(o.rect[0], o.rect[1], o.rect[2], o.rect[3], tileW, tileH, gridW, gridH, x0, y0, x1, y1);
            touched = (uint32_t)((x1 - x0) * (y1 - y0));
            // ... and inside that box only the columns the ellipse reaches, row group by row group (rect_row_groups4)
            if (trimRects && tilePieces && touched)
                touched = rect_row_groups4(o.sx, o.sy, o.cov2d[0], o.cov2d[1], o.cov2d[3], x0, y0, x1, y1, (int)cam.H, pc);
            tr = make_ushort4((unsigned short)x0, (unsigned short)y0, (unsigned short)x1, (unsigned short)y1);
            visible = true;
            // A view under depth cuts: a Gaussian that lies beyond the deepest cut of every 4 x 4 tiles its rect touches would
            // lose all its pairs in the cut expansion one by one (binning.hip, cut_super_kernel).  Dropped here it touches no
            // tile at all: no SH rows fetched for its colour, no candidates enumerated, its depth key sorted behind the rest.
            // Exact as the cuts are: a forward that needed more is detected and repeated without them.
            if (cc.superCut && touched) {
                const uint32_t key = __float_as_uint(o.depth);
                bool reach = false;
                const int sx1 = (x1 - 1) / GS_CUT_SUPER, sy1 = (y1 - 1) / GS_CUT_SUPER;
                for (int sy = y0 / GS_CUT_SUPER; sy <= sy1 && !reach; sy++)
                    for (int sx = x0 / GS_CUT_SUPER; sx <= sx1; sx++)
                        if (key <= 0xFFFFFFFFu - cc.superCut[sy * cc.superW + sx]) { reach = true; break; }
                if (!reach) { atomicAdd(&sDropped, touched); touched = 0; tr = make_ushort4(0, 0, 0, 0); pc[0] = pc[1] = pc[2] = pc[3] = 0u; }
            }
        }
        tileRect[p] = tr;
        if (trimRects && tilePieces) tilePieces[p] = make_uint4(pc[0], pc[1], pc[2], pc[3]);
        tilesTouched[p] = touched;
        myTouched = touched;
        depthKey[p] = (touched || !noKeyForUntouched) ? __float_as_uint(o.depth) : GS_SORT_NO_KEY;     // binning.hip, bin_prep_kernel
        depthVal[p] = (uint32_t)p;
    }
    // visible count: one plain store per block, summed when somebody asks (gs_last_stats).  A same-address atomic per
    // wave here cost a third of the kernel (4700 atomics on one counter: 82 -> 55 us).
    const int nvis = __syncthreads_count(visible);
    if (threadIdx.x == 0) visPerBlock[blockIdx.x] = (uint32_t)nvis;
    if (cc.superCut && threadIdx.x == 0) cc.dropPerBlock[blockIdx.x] = sDropped;       // (behind the barrier of the count above)
    // (every lane reads tilesTouched / writes the colour floats of ITS OWN record: program order is all that is needed)
    if (SELF) colour_rider_wave(self, blockIdx.x * (PROJ_FUSED_THREADS / 64) + wv, shLds + wv * 64 * GS_RIDER_ROW, lane, (int)myTouched);
}

// ---------------------------------------------------------------------------------------------
// fused backward: gradAcc16 (d packed) + raw parameters -> raw-parameter gradients
// ---------------------------------------------------------------------------------------------
// EMIT_MG: data-parallel variant.  The SH gradient of one view is the outer product basis_k(xyz - cam) x mg with
// mg[3] the colour cotangent after the max(., 0) gate, so a rank only has to publish mg (12 B per Gaussian instead
// of 12 K); every rank rebuilds and sums the SH gradients of all views itself (sh_grad_from_views_kernel).
//
// MODE 2 (ADAM): no gradient leaves the kernel; every parameter element is updated in place with its Adam step
// (optim.hip's arithmetic) against the moment arenas that mirror the parameter arena.  It saves the gradient
// arena's round trip (written here, read by adam_kernel): 344 B of the ~1 KB per Gaussian the two kernels move.
struct AdamFuse {
    const float* pBase;   // parameter arena base: every tensor of the forward lies inside it
    float* mBase;
    float* vBase;
    float lr[6];          // xyz, f_dc, f_rest, scales, rotation, opacity
    float b1, b2, eps, gscale;
    const uint32_t* gate; // device word: non-zero = leave parameters and moments alone (the forward overflowed its
                          // reserved pair capacity and rendered nothing: gs_ctx.h adamGate)
    // data-parallel steps (round 5: the step's gate rides in its first collective, dp.hip):
    const uint32_t* ovf;  // the forward's overflow word and
    float* rider;         // where the kernel's first thread stores it as 0.0f / 1.0f (gs_set_overflow_rider), or nullptr
    // the gate as the OR of `gwCount` words `gwStride` apart (the ranks' riders behind their gathered colour cotangents:
    // gs_set_gathered_gate), stored to gwOut for the optimizer kernels queued behind; gwWords == nullptr: *gate
    const uint32_t* gwWords;
    long long gwStride;
    int gwCount;
    uint32_t* gwOut;
    uint32_t* seen;       // set to 1 by a kernel that finds its gate raised (gs_set_gate_seen), or nullptr
};

// the gate of an optimizer kernel: its one word, or the OR of the gathered words (every thread reads the same few words)
__device__ __forceinline__ uint32_t adam_gate_word(const AdamFuse& A)
{
    if (!A.gwWords) return A.gate ? *A.gate : 0u;
    uint32_t g = 0;
    for (int r = 0; r < A.gwCount; r++) g |= A.gwWords[(long long)r * A.gwStride];
    return g;
}
// ... published by the kernel's first thread
__device__ __forceinline__ void adam_gate_publish(const AdamFuse& A, uint32_t g)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (A.gwWords && A.gwOut) *A.gwOut = g;
        if (g && A.seen) *A.seen = 1u;
    }
}

// one element's Adam step on values already in registers (the loads are issued long before, the stores after)
__device__ __forceinline__ void adam_step(const AdamFuse& A, float g, float lr, float& p, float& m, float& v)
{
    const float gr = g * A.gscale;
    m = A.b1 * m + (1.0f - A.b1) * gr;
    v = A.b2 * v + (1.0f - A.b2) * gr * gr;
    p = p - gs_adam_delta(lr, m, v, A.eps);
}

// Adam over the f_rest rows of one wave (64 Gaussians), gradients in the wave's LDS rows: walked in memory order
// against p, m, v, four consecutive elements per lane, the loads of a whole batch in flight before the first update
// (nothing may wait on a store).  16-B accesses need 16-B aligned addresses: scalar head up to the first aligned
// element, float4 body, scalar tail.
__device__ __forceinline__ void adam_rows(const AdamFuse& adam, const float* frest, int row0, int rows, int L,
                                          const float* myRows, int lane, float lr)
{
    const size_t off0 = (size_t)(frest - adam.pBase) + (size_t)row0 * L;
    float* P = const_cast<float*>(adam.pBase) + off0;
    float* M = adam.mBase + off0;
    float* V = adam.vBase + off0;
    const int total = rows * L;
    auto grad_at = [&](int e) { const int r = e / L; return myRows[r * (L + 1) + (e - r * L)]; };
    const int head = min(total, (int)((4 - (off0 & 3)) & 3));
    const int n4 = (total - head) >> 2;
    const int tail0 = head + 4 * n4;
    if (lane < head + (total - tail0)) {
        const int e = lane < head ? lane : tail0 + (lane - head);
        float pv = P[e], mv = M[e], vv = V[e];
        adam_step(adam, grad_at(e), lr, pv, mv, vv);
        P[e] = pv; M[e] = mv; V[e] = vv;
    }
    float4* P4 = reinterpret_cast<float4*>(P + head);
    float4* M4 = reinterpret_cast<float4*>(M + head);
    float4* V4 = reinterpret_cast<float4*>(V + head);
    constexpr int B = 6;
    for (int e0 = lane; e0 < n4; e0 += 64 * B) {
        float4 pp[B], mm[B], vv[B];
#pragma unroll
        for (int b = 0; b < B; b++) {      // unconditional, clamped: every load of the batch in flight at once
            const int e = min(e0 + 64 * b, n4 - 1);
            pp[b] = P4[e]; mm[b] = M4[e]; vv[b] = V4[e];
        }
#pragma unroll
        for (int b = 0; b < B; b++) {
            const int e = e0 + 64 * b;
            if (e < n4) {
                const int f = head + 4 * e;
                adam_step(adam, grad_at(f), lr, pp[b].x, mm[b].x, vv[b].x);
                adam_step(adam, grad_at(f + 1), lr, pp[b].y, mm[b].y, vv[b].y);
                adam_step(adam, grad_at(f + 2), lr, pp[b].z, mm[b].z, vv[b].z);
                adam_step(adam, grad_at(f + 3), lr, pp[b].w, mm[b].w, vv[b].w);
                P4[e] = pp[b]; M4[e] = mm[b]; V4[e] = vv[b];
            }
        }
    }
}

// The same update when the wave's parameter rows are still in registers from the staging load (sh_rows_load4: float4
// `lane + 64 i` of the span, 16-B aligned): f_rest is read from HBM once, not twice -- its LDS copy has been overwritten
// with the gradients by now, and a second read of the 18 KB span came from HBM again (the PMC fetch bytes of the
// kernel were 1.25x its algorithmic bytes, the excess = N x 288 B).  Moments in batches of 3 float4, one batch ahead
// (the 18 kept float4 leave room for no more under the 256 registers of two waves per SIMD).
__device__ __forceinline__ void adam_rows_kept(const AdamFuse& adam, const float* frest, int row0, int total4, int L,
                                               const float* myRows, int lane, float lr,
                                               float4 (&pp)[SH_ROWS_MAX4])
{
    const size_t off0 = (size_t)(frest - adam.pBase) + (size_t)row0 * L;
    float4* P4 = reinterpret_cast<float4*>(const_cast<float*>(adam.pBase) + off0);
    float4* M4 = reinterpret_cast<float4*>(adam.mBase + off0);
    float4* V4 = reinterpret_cast<float4*>(adam.vBase + off0);
    auto grad_at = [&](int e) { const int r = e / L; return myRows[r * (L + 1) + (e - r * L)]; };
    constexpr int B = 3, NB = SH_ROWS_MAX4 / B;
    static_assert(NB * B == SH_ROWS_MAX4, "whole batches");
#ifndef GS_ADAM_ROWS_DEPTH
#define GS_ADAM_ROWS_DEPTH 2
#endif
    constexpr int D = GS_ADAM_ROWS_DEPTH;      // batches of moments in flight ahead of the update
    float4 mm[D][B], vv[D][B];
    auto load = [&](int k, float4 (&m)[B], float4 (&v)[B]) {
#pragma unroll
        for (int b = 0; b < B; b++) {
            const int e = min(lane + 64 * (k * B + b), total4 - 1);
            m[b] = M4[e]; v[b] = V4[e];
        }
    };
    auto update = [&](int k, float4 (&m)[B], float4 (&v)[B]) {
#pragma unroll
        for (int b = 0; b < B; b++) {
            const int e = lane + 64 * (k * B + b);
            if (e < total4) {
                float4& p = pp[k * B + b];
                adam_step(adam, grad_at(4 * e), lr, p.x, m[b].x, v[b].x);
                adam_step(adam, grad_at(4 * e + 1), lr, p.y, m[b].y, v[b].y);
                adam_step(adam, grad_at(4 * e + 2), lr, p.z, m[b].z, v[b].z);
                adam_step(adam, grad_at(4 * e + 3), lr, p.w, m[b].w, v[b].w);
                P4[e] = p; M4[e] = m[b]; V4[e] = v[b];
            }
        }
    };
#pragma unroll
    for (int k = 0; k < D - 1 && k < NB; k++) load(k, mm[k % D], vv[k % D]);
#pragma unroll
    for (int k = 0; k < NB; k++) {
        if (k + D - 1 < NB) load(k + D - 1, mm[(k + D - 1) % D], vv[(k + D - 1) % D]);
        update(k, mm[k % D], vv[k % D]);
    }
}

// A cotangent row that is not finite.  The blend backward gates a pixel's contribution to a splat branch-free (alpha and its
// gradient times 0 past the pixel's nContrib or under the 0.99 clamp), which is exact while the splat's weight exp(-q/2) is finite.
// A conic that is not positive definite -- float32 cancellation in the determinant of a needle tens of thousands of pixels long:
// the soak's Gaussian of scales (8.2, 0.007, 0.006), cov2d = (4.2e7, -3.8e7; -3.8e7, 3.5e7) -- has q < 0 without bound, the weight
// overflows and 0 x inf = NaN lands in the accumulator row of THAT splat (no other row, no pixel state: the NaN never enters T or
// the owed colour).  The reference's loop does not evaluate a pixel past its nContrib at all.  Such a row is dropped whole -- a zero
// gradient for this Gaussian in this step -- instead of poisoning its parameters and moments for good (one Gaussian of a million
// after ~8500 iterations of tools/soak.py, on the reference's lists and on trimmed rects alike).
__device__ __forceinline__ void drop_nonfinite_row(float4& g0, float4& g1, float4& g2)
{
#ifdef GS_KEEP_NONFINITE_ROWS      // (experiment build: the behaviour before the fix, to see that the test would have caught it)
    return;
#endif
    auto fin = [](float v) { return (v - v) == 0.0f; };
    if (!(fin(g0.x) && fin(g0.y) && fin(g0.z) && fin(g0.w) && fin(g1.x) && fin(g1.y) && fin(g1.z) && fin(g1.w) && fin(g2.x) &&
          fin(g2.y) && fin(g2.z))) {
        g0 = make_float4(0.f, 0.f, 0.f, 0.f); g1 = g0; g2 = g0;
    }
}

template <int MODE>
__global__ __launch_bounds__(PROJ_FUSED_THREADS) void proj_bwd_fused_kernel(
    int N, int K, int degree, CamParams cam, const float* xyz, const float* fdc,
    const float* frest, const float* scalesRaw, const float* rotRaw,
    const float* opacityRaw, const float* __restrict__ gradAcc16, float* gXyz,
    float* gFdc, float* gFrest, float* gScales, float* gRot,
    float* gOpacity, float* __restrict__ gradNormAccum, AdamFuse adam)
{
    constexpr bool EMIT_MG = MODE == 1, ADAM = MODE == 2;
    extern __shared__ float shLds[];
    // no update from a forward that did not render (gs_ctx.h adamGate).  The word is requested here and looked at only
    // in front of the first store: a branch on it up here put a dependent scalar-load latency in front of every
    // block's loads (2 M Gaussians: 0.77 -> 0.88 ms)
    const uint32_t gateWord = ADAM ? *adam.gate : 0u;
    const int p = blockIdx.x * PROJ_FUSED_THREADS + threadIdx.x;
    if (MODE == 0 && adam.rider && p == 0) *adam.rider = *adam.ovf ? 1.0f : 0.0f;      // (data-parallel all-reduce: the gate rides behind the gradients)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int L = (K - 1) * 3;
    float* myRows = shLds + wv * 64 * (L + 1);
    const int row0 = blockIdx.x * PROJ_FUSED_THREADS + wv * 64;
    const int rows = min(64, N - row0);
    // fused Adam: the wave's f_rest span stays in registers for the update at the bottom (adam_rows_kept) when it is
    // float4-addressable in all three arenas and fits the 18 registers (K <= 25)
    float4 keptRows[SH_ROWS_MAX4];
    const int total4 = (rows * L) >> 2;
    const bool kept = ADAM && rows > 0 && L > 0 && (L & 3) == 0 && total4 <= 64 * SH_ROWS_MAX4 &&
                      (((size_t)(frest - adam.pBase) + (size_t)row0 * L) & 3) == 0;
    if (kept) {
        sh_rows_load4(frest + (size_t)row0 * L, total4, 0, lane, keptRows);
        sh_rows_to_lds4(myRows, total4, L, 0, lane, keptRows);
    } else if (rows > 0 && L > 0) sh_rows_in(myRows, frest + (size_t)row0 * L, rows, L, lane);
    if (p < N) {
    const float4* ga = reinterpret_cast<const float4*>(gradAcc16 + (size_t)p * 16);
    float4 g0 = ga[0], g1 = ga[1], g2 = ga[2];
    drop_nonfinite_row(g0, g1, g2);
    const float cm[2] = {g0.x, g0.y};
    // row: dmx dmy dc00 dc01 | dc10 dc11 dr dg | db dop ddepth
    const float ccon[4] = {g0.z, g0.w, g1.x, g1.y};
    const float ccol[3] = {g1.z, g1.w, g2.x};
    const float cotOpacity = g2.y, cotDepth = g2.z;
    const float ccov[4] = {0.f, 0.f, 0.f, 0.f};   // cov2d is unused downstream (GaussianRenderer.swift:796-802)

    const float m[3] = {xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2]};
    const float sr[3] = {scalesRaw[3 * p], scalesRaw[3 * p + 1], scalesRaw[3 * p + 2]};
    const float s[3] = {expf(sr[0]), expf(sr[1]), expf(sr[2])};
    const float rr[4] = {rotRaw[4 * p], rotRaw[4 * p + 1], rotRaw[4 * p + 2], rotRaw[4 * p + 3]};
    const float n2 = rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2] + rr[3] * rr[3];
    const float nrm = sqrtf(n2);
    const float den = nrm + 1e-8f;
    const float q[4] = {rr[0] / den, rr[1] / den, rr[2] / den, rr[3] / den};

    GeomGrads g;
    project_geometry_bwd(m, s, q, cam, cm, cotDepth, ccov, ccon, g);
    // A Gaussian no pixel blended (not visible, or off every tile) arrives with an all-zero cotangent row and its
    // gradient is exactly zero.  The reference's arithmetic evaluates J^T 0 term by term, which is 0 * inf = NaN when
    // the point sits within ~1e-3 of the camera plane (1 / t_z^2 overflows); one such NaN poisons Adam for good.
    // The fused path returns the exact value instead (the op-level gs_projection_backward keeps the 1:1 arithmetic).
    if (g0.x == 0.f && g0.y == 0.f && g0.z == 0.f && g0.w == 0.f && g1.x == 0.f && g1.y == 0.f && g1.z == 0.f &&
        g1.w == 0.f && g2.x == 0.f && g2.y == 0.f && g2.z == 0.f) {
#pragma unroll
        for (int a = 0; a < 3; a++) { g.dm[a] = 0.f; g.ds[a] = 0.f; }
#pragma unroll
        for (int a = 0; a < 4; a++) g.dq[a] = 0.f;
    }

    const float x = m[0] - cam.cam[0], y = m[1] - cam.cam[1], z = m[2] - cam.cam[2];
    const float* d0 = fdc + (size_t)p * 3;
    float* rest = myRows + lane * (L + 1);     // coefficients in, gradients out, in place (own row only)
    float* gd0 = gFdc + (size_t)p * 3;
    float d[3];
    const float opr = opacityRaw[p];
    // fused Adam: arena offsets of this Gaussian's 14 small elements, their moments loaded now (independent loads,
    // in flight under the arithmetic below) -- gradients sg_, values from the registers that already hold them
    float sm_[14], sv_[14], sg_[14], d0v[3] = {0.f, 0.f, 0.f};
    // (the offsets are rebuilt in front of the stores rather than kept: 28 registers of a kernel at its 256)
    auto small_off = [&](int i) -> size_t {
        const float* ptr[14] = {xyz + 3 * p, xyz + 3 * p + 1, xyz + 3 * p + 2, scalesRaw + 3 * p, scalesRaw + 3 * p + 1,
                                scalesRaw + 3 * p + 2, rotRaw + 4 * p, rotRaw + 4 * p + 1, rotRaw + 4 * p + 2,
                                rotRaw + 4 * p + 3, opacityRaw + p, d0, d0 + 1, d0 + 2};
        return (size_t)(ptr[i] - adam.pBase);
    };
    if (ADAM) { d0v[0] = d0[0]; d0v[1] = d0[1]; d0v[2] = d0[2]; }
    // one 12- or 16-byte access per tensor and lane instead of one per element: the 14 elements' moments in and their
    // parameters and moments out were 70 four-byte memory instructions per lane, 12 % of the kernel's bytes for 19 % of
    // its time (a build without them: 0.138 -> 0.112 ms); 4-byte aligned is all the arena promises for [N,3] tensors
    struct __attribute__((packed, aligned(4))) V3 { float x, y, z; };
    struct __attribute__((packed, aligned(4))) V4 { float x, y, z, w; };
    auto small_moments = [&]() {
        auto in3 = [&](const float* base, int i0) {
            const size_t o = small_off(i0);
            const V3 a = *reinterpret_cast<const V3*>(adam.mBase + o), b = *reinterpret_cast<const V3*>(adam.vBase + o);
            sm_[i0] = a.x; sm_[i0 + 1] = a.y; sm_[i0 + 2] = a.z; sv_[i0] = b.x; sv_[i0 + 1] = b.y; sv_[i0 + 2] = b.z;
            (void)base;
        };
        in3(xyz, 0); in3(scalesRaw, 3); in3(fdc, 11);
        {
            const size_t o = small_off(6);
            const V4 a = *reinterpret_cast<const V4*>(adam.mBase + o), b = *reinterpret_cast<const V4*>(adam.vBase + o);
            sm_[6] = a.x; sm_[7] = a.y; sm_[8] = a.z; sm_[9] = a.w; sv_[6] = b.x; sv_[7] = b.y; sv_[8] = b.z; sv_[9] = b.w;
        }
        { const size_t o = small_off(10); sm_[10] = adam.mBase[o]; sv_[10] = adam.vBase[o]; }
    };
    if (ADAM && !GS_PROJ_LATE_MOMENTS) small_moments();
    if (EMIT_MG)      // no SH gradient is written; gFdc, when given, receives the [N,3] gated colour cotangent
        color_backward(degree, K, x, y, z, ccol,
                       [&](int k, int ch) { return k == 0 ? d0[ch] : rest[(k - 1) * 3 + ch]; },
                       [&](int, int, float) {}, d, gFdc ? gd0 : nullptr);
    else {
        float gdc[3] = {0.f, 0.f, 0.f};          // the DC gradient stays in registers for the fused Adam step
        color_backward(degree, K, x, y, z, ccol,
                       [&](int k, int ch) { return k == 0 ? d0[ch] : rest[(k - 1) * 3 + ch]; },
                       [&](int k, int ch, float v) {
                           if (k == 0) { if (ADAM) gdc[ch] = v; else gd0[ch] = v; }
                           else rest[(k - 1) * 3 + ch] = v;
                       }, d);
        if (ADAM) { sg_[11] = gdc[0]; sg_[12] = gdc[1]; sg_[13] = gdc[2]; }
    }
    if (ADAM && GS_PROJ_LATE_MOMENTS) small_moments();
    const float gx = g.dm[0] + d[0], gy = g.dm[1] + d[1], gz = g.dm[2] + d[2];
    if (ADAM) { sg_[0] = gx; sg_[1] = gy; sg_[2] = gz; }
    else { gXyz[3 * p] = gx; gXyz[3 * p + 1] = gy; gXyz[3 * p + 2] = gz; }
    if (gradNormAccum && !gateWord) gradNormAccum[p] += sqrtf(gx * gx + gy * gy + gz * gz);   // accum_grad_norm (densify.hip), fused
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float v = g.ds[a] * s[a];          // d exp
        if (ADAM) sg_[3 + a] = v; else gScales[3 * p + a] = v;
    }
    // rotation normalisation VJP: y = q / (|q| + 1e-8)
    const float dot = g.dq[0] * rr[0] + g.dq[1] * rr[1] + g.dq[2] * rr[2] + g.dq[3] * rr[3];
    const float dn = -dot / (den * den);
    const float dn2 = dn * 0.5f / nrm;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const float v = g.dq[a] / den + 2.0f * rr[a] * dn2;
        if (ADAM) sg_[6 + a] = v; else gRot[4 * p + a] = v;
    }
    const float sg = 1.0f / (1.0f + expf(-opr));
    const float gop = cotOpacity * sg * (1.0f - sg);
    if (ADAM) sg_[10] = gop; else gOpacity[p] = gop;
    if (ADAM && !gateWord) {
        // the 14 small elements: values and moments were loaded at the top; one burst of stores here
        float sp[14] = {m[0], m[1], m[2], sr[0], sr[1], sr[2], rr[0], rr[1], rr[2], rr[3], opr, d0v[0], d0v[1], d0v[2]};
        const float slr[14] = {adam.lr[0], adam.lr[0], adam.lr[0], adam.lr[3], adam.lr[3], adam.lr[3], adam.lr[4],
                               adam.lr[4], adam.lr[4], adam.lr[4], adam.lr[5], adam.lr[1], adam.lr[1], adam.lr[1]};
#pragma unroll
        for (int i = 0; i < 14; i++) adam_step(adam, sg_[i], slr[i], sp[i], sm_[i], sv_[i]);
        float* pB = const_cast<float*>(adam.pBase);
        auto out3 = [&](int i0) {
            const size_t o = small_off(i0);
            *reinterpret_cast<V3*>(pB + o) = V3{sp[i0], sp[i0 + 1], sp[i0 + 2]};
            *reinterpret_cast<V3*>(adam.mBase + o) = V3{sm_[i0], sm_[i0 + 1], sm_[i0 + 2]};
            *reinterpret_cast<V3*>(adam.vBase + o) = V3{sv_[i0], sv_[i0 + 1], sv_[i0 + 2]};
        };
        out3(0); out3(3); out3(11);
        {
            const size_t o = small_off(6);
            *reinterpret_cast<V4*>(pB + o) = V4{sp[6], sp[7], sp[8], sp[9]};
            *reinterpret_cast<V4*>(adam.mBase + o) = V4{sm_[6], sm_[7], sm_[8], sm_[9]};
            *reinterpret_cast<V4*>(adam.vBase + o) = V4{sv_[6], sv_[7], sv_[8], sv_[9]};
        }
        { const size_t o = small_off(10); pB[o] = sp[10]; adam.mBase[o] = sm_[10]; adam.vBase[o] = sv_[10]; }
    }
    }
    if (MODE == 0 && rows > 0 && L > 0) sh_rows_out(myRows, gFrest + (size_t)row0 * L, rows, L, lane);
    if (ADAM && !gateWord && rows > 0 && L > 0) {
        if (kept) adam_rows_kept(adam, frest, row0, total4, L, myRows, lane, adam.lr[2], keptRows);
        else adam_rows(adam, frest, row0, rows, L, myRows, lane, adam.lr[2]);
    }
}

// ---------------------------------------------------------------------------------------------
// data-parallel: the gated colour cotangent straight from the blend backward's accumulator (columns dr dg db of
// gradAcc16) and the gate bits the forward left in packed12 -- ready before the projection backward runs, so its
// all-gather overlaps with that kernel
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void color_cot_kernel(int N, const float* __restrict__ gradAcc16,
                                                        const float* __restrict__ packed12, float* __restrict__ out,
                                                        const uint32_t* __restrict__ ovf, float* __restrict__ rider)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p == 0 && rider) *rider = *ovf ? 1.0f : 0.0f;      // this rank's word of the step's gate, gathered with the cotangents
    if (p >= N) return;
    const float* ga = gradAcc16 + (size_t)p * 16;
    const uint32_t gate = __float_as_uint(packed12[(size_t)p * 12 + 11]);
    float4 r0 = *reinterpret_cast<const float4*>(ga), r1 = *reinterpret_cast<const float4*>(ga + 4), r2 = *reinterpret_cast<const float4*>(ga + 8);
    drop_nonfinite_row(r0, r1, r2);        // (a row the blend backward left non-finite is dropped whole: proj_bwd_fused_kernel)
    const float col[3] = {r1.z, r1.w, r2.x};
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const uint32_t side = (gate >> (2 * ch)) & 3u;
        const float g = col[ch];
        out[(size_t)p * 3 + ch] = side == 2u ? g : (side == 1u ? 0.5f * g : 0.0f);      // d max(a, 0): tie -> 1/2
    }
}

// ---------------------------------------------------------------------------------------------
// data-parallel SH gradient: grad_shs[k] = sum over views r of basis_k(xyz - cam_r) * mg_r
// ---------------------------------------------------------------------------------------------
struct ViewCenters {
    float c[16][3];
    int n;
};

// ADAM: instead of writing the summed SH gradients, apply their Adam step to features_dc / features_rest in place
// (fdcParam / frestParam inside the parameter arena adam.pBase; lr[1], lr[2]; gscale = 1 / world).
template <bool ADAM>
__global__ __launch_bounds__(PROJ_FUSED_THREADS) void sh_grad_from_views_kernel(
    int N, int K, int degree, ViewCenters views, const float* __restrict__ xyz, const float* __restrict__ mgAll,
    long long mgStride, float* gFdc, float* gFrest, const float* fdcParam, const float* frestParam, AdamFuse adam)
{
    extern __shared__ float shLds[];
    // (looked at in front of the first store, proj_bwd_fused_kernel.)  With gathered words (round 5) this kernel is the
    // first to know the step's gate: it ORs the ranks' words for itself and leaves the result for the kernels behind it
    const uint32_t gateWord = (ADAM || adam.gwWords) ? adam_gate_word(adam) : 0u;
    adam_gate_publish(adam, gateWord);
    const int p = blockIdx.x * PROJ_FUSED_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int L = (K - 1) * 3;
    float* rest = shLds + wv * 64 * (L + 1) + lane * (L + 1);
    const int row0 = blockIdx.x * PROJ_FUSED_THREADS + wv * 64;
    const int rows = min(64, N - row0);
    if (p < N) {
        for (int i = 0; i < L; i++) rest[i] = 0.0f;
        float dc[3] = {0.f, 0.f, 0.f};
        const float m0 = xyz[3 * p], m1 = xyz[3 * p + 1], m2 = xyz[3 * p + 2];
        for (int r = 0; r < views.n; r++) {
            const float* mg = mgAll + (size_t)r * (size_t)mgStride + (size_t)p * 3;
            const float g0 = mg[0], g1 = mg[1], g2 = mg[2];
            if (g0 == 0.0f && g1 == 0.0f && g2 == 0.0f) continue;
            sh_foreach(degree, m0 - views.c[r][0], m1 - views.c[r][1], m2 - views.c[r][2],
                       [&](int k, float b, float, float, float) {
                           if (k == 0) { dc[0] += b * g0; dc[1] += b * g1; dc[2] += b * g2; }
                           else { rest[(k - 1) * 3] += b * g0; rest[(k - 1) * 3 + 1] += b * g1; rest[(k - 1) * 3 + 2] += b * g2; }
                       });
        }
        if (ADAM && gateWord) {
            // the step's forward overflowed on some rank: no update
        } else if (ADAM) {
            const size_t off = (size_t)(fdcParam - adam.pBase) + 3 * (size_t)p;
            float pv[3], mv[3], vv[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) { pv[ch] = adam.pBase[off + ch]; mv[ch] = adam.mBase[off + ch]; vv[ch] = adam.vBase[off + ch]; }
#pragma unroll
            for (int ch = 0; ch < 3; ch++) adam_step(adam, dc[ch], adam.lr[1], pv[ch], mv[ch], vv[ch]);
            // (one 12-byte store per arena instead of three 4-byte ones: proj_bwd_fused_kernel, small_moments)
            struct __attribute__((packed, aligned(4))) V3 { float x, y, z; };
            *reinterpret_cast<V3*>(const_cast<float*>(adam.pBase) + off) = V3{pv[0], pv[1], pv[2]};
            *reinterpret_cast<V3*>(adam.mBase + off) = V3{mv[0], mv[1], mv[2]};
            *reinterpret_cast<V3*>(adam.vBase + off) = V3{vv[0], vv[1], vv[2]};
        } else {
            gFdc[3 * p] = dc[0]; gFdc[3 * p + 1] = dc[1]; gFdc[3 * p + 2] = dc[2];
        }
    }
    if (rows > 0 && L > 0) {
        if (ADAM) { if (!gateWord) adam_rows(adam, frestParam, row0, rows, L, shLds + wv * 64 * (L + 1), lane, adam.lr[2]); }
        else sh_rows_out(shLds + wv * 64 * (L + 1), gFrest + (size_t)row0 * L, rows, L, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// data-parallel step, round 6: the SH rows are read ONCE per step.
//
// Rounds 2-5 split the fused projection backward + Adam of the single-device step into proj_bwd_fused_kernel<1> (the four
// geometry gradients; it staged every Gaussian's 288 B of SH rows through LDS only for the view-direction term of the xyz
// gradient, d_r = sum_k grad basis_k(xyz - cam_r) (SH_k . cc_r)) and sh_grad_from_views_kernel<true> (the SH gradients rebuilt
// from the gathered colour cotangents + their Adam step, which reads the same rows again): 46 + 93 us on one rank against 128
// for the fused kernel.  But d_r is, like the SH gradient itself, a function of replicated values (xyz, SH, camera centre)
// and of the view's gathered colour cotangent alone: every rank can rebuild sum_r d_r while it has the SH rows in hand
// for their Adam step.  So:
//   proj_bwd_geom_kernel       geometry only (56 + 64 B in, 44 + 12 B out per Gaussian; no LDS, no SH rows): the xyz gradient
//                              WITHOUT the view-direction term goes to the gradient arena (the all-reduce sums it) and, a
//                              copy, to xyzOwn -- the densify statistic needs this view's FULL xyz gradient;
//   sh_views_dir_adam_kernel   rows to registers + LDS once; pass 1 over the views: d_r from the OLD coefficients, their sum
//                              to xyzAdd, |xyzOwn_r + d_r| of this rank's own view(s) into the densify statistic; pass 2: the SH
//                              gradients into the same LDS rows; Adam from the kept registers (adam_rows_kept);
//   adam_kernel                the geometry slice after the all-reduce, gradient = reduced + xyzAdd on the xyz segment.
// Sum over views of (geometry_r + d_r) becomes (sum geometry_r) + (sum d_r): the same value up to float association.
// ---------------------------------------------------------------------------------------------
struct ViewCentersOwn {
    float c[16][3];
    const float* own[16];      // xyzOwn of the views THIS rank rendered (their full xyz gradient = own + d_r), else nullptr
    int n;
};

__global__ __launch_bounds__(256) void proj_bwd_geom_kernel(
    int N, CamParams cam, const float* __restrict__ xyz, const float* __restrict__ scalesRaw, const float* __restrict__ rotRaw,
    const float* __restrict__ opacityRaw, const float* __restrict__ gradAcc16, float* __restrict__ gXyz,
    float* __restrict__ gScales, float* __restrict__ gRot, float* __restrict__ gOpacity, float* __restrict__ xyzOwn,
    const float* __restrict__ packed12, float* __restrict__ ccOut, const uint32_t* __restrict__ ovf, float* __restrict__ rider)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    // ccOut != nullptr: color_cot_kernel's work rides here (the gated colour cotangents + this rank's word of the step's gate):
    // with the SH rows gone this kernel is ~10 us, too short for the all-gather to hide under -- one launch and one fork less
    if (ccOut && p == 0 && rider) *rider = *ovf ? 1.0f : 0.0f;
    if (p >= N) return;
    const float4* ga = reinterpret_cast<const float4*>(gradAcc16 + (size_t)p * 16);
    float4 g0 = ga[0], g1 = ga[1], g2 = ga[2];
    drop_nonfinite_row(g0, g1, g2);
    if (ccOut) {
        const uint32_t gate = __float_as_uint(packed12[(size_t)p * 12 + 11]);
        const float col[3] = {g1.z, g1.w, g2.x};
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const uint32_t side = (gate >> (2 * ch)) & 3u;
            ccOut[(size_t)p * 3 + ch] = side == 2u ? col[ch] : (side == 1u ? 0.5f * col[ch] : 0.0f);      // d max(a, 0): tie -> 1/2
        }
    }
    // row: dmx dmy dc00 dc01 | dc10 dc11 dr dg | db dop ddepth
    const float cm[2] = {g0.x, g0.y};
    const float ccon[4] = {g0.z, g0.w, g1.x, g1.y};
    const float cotOpacity = g2.y, cotDepth = g2.z;
    const float ccov[4] = {0.f, 0.f, 0.f, 0.f};   // cov2d is unused downstream (GaussianRenderer.swift:796-802)
    const float m[3] = {xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2]};
    const float sr[3] = {scalesRaw[3 * p], scalesRaw[3 * p + 1], scalesRaw[3 * p + 2]};
    const float s[3] = {expf(sr[0]), expf(sr[1]), expf(sr[2])};
    const float rr[4] = {rotRaw[4 * p], rotRaw[4 * p + 1], rotRaw[4 * p + 2], rotRaw[4 * p + 3]};
    const float n2 = rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2] + rr[3] * rr[3];
    const float nrm = sqrtf(n2);
    const float den = nrm + 1e-8f;
    const float q[4] = {rr[0] / den, rr[1] / den, rr[2] / den, rr[3] / den};
    GeomGrads g;
    project_geometry_bwd(m, s, q, cam, cm, cotDepth, ccov, ccon, g);
    // (a Gaussian no pixel blended: the exact zero instead of the reference's 0 * inf, as proj_bwd_fused_kernel)
    if (g0.x == 0.f && g0.y == 0.f && g0.z == 0.f && g0.w == 0.f && g1.x == 0.f && g1.y == 0.f && g1.z == 0.f &&
        g1.w == 0.f && g2.x == 0.f && g2.y == 0.f && g2.z == 0.f) {
#pragma unroll
        for (int a = 0; a < 3; a++) { g.dm[a] = 0.f; g.ds[a] = 0.f; }
#pragma unroll
        for (int a = 0; a < 4; a++) g.dq[a] = 0.f;
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        gXyz[3 * p + a] = g.dm[a];
        xyzOwn[3 * p + a] = g.dm[a];
        gScales[3 * p + a] = g.ds[a] * s[a];          // d exp
    }
    // rotation normalisation VJP: y = q / (|q| + 1e-8)
    const float dot = g.dq[0] * rr[0] + g.dq[1] * rr[1] + g.dq[2] * rr[2] + g.dq[3] * rr[3];
    const float dn = -dot / (den * den);
    const float dn2 = dn * 0.5f / nrm;
#pragma unroll
    for (int a = 0; a < 4; a++) gRot[4 * p + a] = g.dq[a] / den + 2.0f * rr[a] * dn2;
    const float sg = 1.0f / (1.0f + expf(-opacityRaw[p]));
    gOpacity[p] = cotOpacity * sg * (1.0f - sg);
}

__global__ __launch_bounds__(PROJ_FUSED_THREADS) void sh_views_dir_adam_kernel(
    int N, int K, int degree, ViewCentersOwn views, const float* __restrict__ xyz, const float* __restrict__ mgAll,
    long long mgStride, const float* fdcParam, const float* frestParam, float* __restrict__ xyzAdd,
    float* __restrict__ gradNormAccum, AdamFuse adam)
{
    extern __shared__ float shLds[];
    const uint32_t gateWord = adam_gate_word(adam);      // (the ranks' gathered words ORed; looked at in front of the first store)
    adam_gate_publish(adam, gateWord);
    const int p = blockIdx.x * PROJ_FUSED_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int L = (K - 1) * 3;
    float* myRows = shLds + wv * 64 * (L + 1);
    float* rest = myRows + lane * (L + 1);
    const int row0 = blockIdx.x * PROJ_FUSED_THREADS + wv * 64;
    const int rows = min(64, N - row0);
    // the wave's f_rest span: HBM -> registers (kept for the Adam update at the bottom) -> LDS rows (read per lane below)
    float4 keptRows[SH_ROWS_MAX4];
    const int total4 = (rows * L) >> 2;
    const bool kept = rows > 0 && L > 0 && (L & 3) == 0 && total4 <= 64 * SH_ROWS_MAX4 &&
                      (((size_t)(frestParam - adam.pBase) + (size_t)row0 * L) & 3) == 0;
    if (kept) {
        sh_rows_load4(frestParam + (size_t)row0 * L, total4, 0, lane, keptRows);
        sh_rows_to_lds4(myRows, total4, L, 0, lane, keptRows);
    } else if (rows > 0 && L > 0) sh_rows_in(myRows, frestParam + (size_t)row0 * L, rows, L, lane);
    if (p < N) {
        const float m0 = xyz[3 * p], m1 = xyz[3 * p + 1], m2 = xyz[3 * p + 2];
        // pass 1: the view-direction terms of the xyz gradient, from the coefficients as the forwards saw them
        float ds0 = 0.f, ds1 = 0.f, ds2 = 0.f, stat = 0.f;
        bool mine = false;
        for (int r = 0; r < views.n; r++) {
            const float* mg = mgAll + (size_t)r * (size_t)mgStride + (size_t)p * 3;
            const float g0 = mg[0], g1 = mg[1], g2 = mg[2];
            float dx = 0.f, dy = 0.f, dz = 0.f;
            if (!(g0 == 0.0f && g1 == 0.0f && g2 == 0.0f))
                sh_foreach(degree, m0 - views.c[r][0], m1 - views.c[r][1], m2 - views.c[r][2],
                           [&](int k, float, float gx, float gy, float gz) {
                               if (k == 0) return;          // (the constant basis function has no gradient)
                               const float w0 = rest[(k - 1) * 3] * g0, w1 = rest[(k - 1) * 3 + 1] * g1, w2 = rest[(k - 1) * 3 + 2] * g2;
                               dx += gx * w0; dy += gy * w0; dz += gz * w0;
                               dx += gx * w1; dy += gy * w1; dz += gz * w1;
                               dx += gx * w2; dy += gy * w2; dz += gz * w2;
                           });
            ds0 += dx; ds1 += dy; ds2 += dz;
            if (views.own[r]) {        // a view this rank rendered: its full xyz gradient, for accum_grad_norm (densify.hip), per VIEW
                const float* o = views.own[r] + (size_t)p * 3;
                const float gx = o[0] + dx, gy = o[1] + dy, gz = o[2] + dz;
                stat += sqrtf(gx * gx + gy * gy + gz * gz);
                mine = true;
            }
        }
        xyzAdd[3 * p] = ds0; xyzAdd[3 * p + 1] = ds1; xyzAdd[3 * p + 2] = ds2;
        if (gradNormAccum && mine && !gateWord) gradNormAccum[p] += stat;
        // pass 2: the SH gradients, summed over the views in view order, in place of the coefficients
        for (int i = 0; i < L; i++) rest[i] = 0.0f;
        float dc[3] = {0.f, 0.f, 0.f};
        for (int r = 0; r < views.n; r++) {
            const float* mg = mgAll + (size_t)r * (size_t)mgStride + (size_t)p * 3;
            const float g0 = mg[0], g1 = mg[1], g2 = mg[2];
            if (g0 == 0.0f && g1 == 0.0f && g2 == 0.0f) continue;
            sh_foreach(degree, m0 - views.c[r][0], m1 - views.c[r][1], m2 - views.c[r][2],
                       [&](int k, float b, float, float, float) {
                           if (k == 0) { dc[0] += b * g0; dc[1] += b * g1; dc[2] += b * g2; }
                           else { rest[(k - 1) * 3] += b * g0; rest[(k - 1) * 3 + 1] += b * g1; rest[(k - 1) * 3 + 2] += b * g2; }
                       });
        }
        if (!gateWord) {
            const size_t off = (size_t)(fdcParam - adam.pBase) + 3 * (size_t)p;
            float pv[3], mv[3], vv[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) { pv[ch] = adam.pBase[off + ch]; mv[ch] = adam.mBase[off + ch]; vv[ch] = adam.vBase[off + ch]; }
#pragma unroll
            for (int ch = 0; ch < 3; ch++) adam_step(adam, dc[ch], adam.lr[1], pv[ch], mv[ch], vv[ch]);
            struct __attribute__((packed, aligned(4))) V3 { float x, y, z; };
            *reinterpret_cast<V3*>(const_cast<float*>(adam.pBase) + off) = V3{pv[0], pv[1], pv[2]};
            *reinterpret_cast<V3*>(adam.mBase + off) = V3{mv[0], mv[1], mv[2]};
            *reinterpret_cast<V3*>(adam.vBase + off) = V3{vv[0], vv[1], vv[2]};
        }
    }
    if (rows > 0 && L > 0 && !gateWord) {
        if (kept) adam_rows_kept(adam, frestParam, row0, total4, L, myRows, lane, adam.lr[2], keptRows);
        else adam_rows(adam, frestParam, row0, rows, L, myRows, lane, adam.lr[2]);
    }
}

// ---------------------------------------------------------------------------------------------
// packing helpers
// ---------------------------------------------------------------------------------------------
__global__ void pack11_to_12_kernel(int N, const float* __restrict__ p11, float* __restrict__ p12)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * 12) return;
    const int g = i / 12, k = i - g * 12;
    p12[i] = k < 11 ? p11[(size_t)g * 11 + k] : 0.0f;
}

__global__ void pack_gaussians_kernel(int N, const float* __restrict__ means2d, const float* __restrict__ conic,
                                      const float* __restrict__ color, const float* __restrict__ opacity,
                                      const float* __restrict__ depths, float* __restrict__ packed)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float* o = packed + (size_t)i * 11;
    o[0] = means2d[2 * i]; o[1] = means2d[2 * i + 1];
#pragma unroll
    for (int k = 0; k < 4; k++) o[2 + k] = conic[4 * i + k];
#pragma unroll
    for (int k = 0; k < 3; k++) o[6 + k] = color[3 * i + k];
    o[9] = opacity[i];
    o[10] = depths[i];
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
int launch_projection_forward(gs_ctx* c, int N, int K, const float* scales, const float* rot, const float* means3d,
                              const float* shs, const CamParams& cam, float* means2d, float* depths, float* color,
                              float* cov2d, float* conic, float* radii, float* rectMin, float* rectMax)
{
    if (N == 0) return GS_OK;
    hipLaunchKernelGGL(proj_fwd_op_kernel, dim3(gs_div_up(N, PROJ_THREADS)), dim3(PROJ_THREADS), 0, c->stream, N, K,
                       c->degree, cam, scales, rot, means3d, shs, means2d, depths, color, cov2d, conic, radii,
                       rectMin, rectMax);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_projection_backward(gs_ctx* c, int N, int K, const float* scales, const float* rot, const float* means3d,
                               const float* shs, const CamParams& cam, const float* cotDepths,
                               const float* cotMeans2d, const float* cotCov2d, const float* cotColor,
                               const float* cotConic, float* gScales, float* gRot, float* gMeans, float* gShs,
                               float* gCam)
{
    if (N == 0) return GS_OK;
    hipLaunchKernelGGL(proj_bwd_op_kernel, dim3(gs_div_up(N, PROJ_THREADS)), dim3(PROJ_THREADS), 0, c->stream, N, K,
                       c->degree, cam, scales, rot, means3d, shs, cotDepths, cotMeans2d, cotCov2d, cotColor,
                       cotConic, gScales, gRot, gMeans, gShs, gCam);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_projection_fused_forward(gs_ctx* c, int N, int K, const float* xyz, const float* fdc, const float* frest,
                                    const float* scales, const float* rot, const float* opacity,
                                    const CamParams& cam, float* radii)
{
    if (N == 0) return GS_OK;
    const int L = (K - 1) * 3;
    const bool twoPhase = (L % 8) == 0 && L >= 48 && L / 2 <= 4 * SH_HALF_MAX4;     // K = 25: two halves of 36 floats
    const size_t lds = sizeof(float) * (PROJ_FUSED_THREADS / 64) * 64 * ((twoPhase ? L / 2 : L) + 1);
    // K = 25: geometry here, the colours as riders of the binning kernels behind this launch (gs_rider.h)
    // -- where those kernels have room for them: the three launches of the splitter depth sort and the tile sort's prefix
    // kernel (16385 .. 655 k Gaussians, not the context's first forward).  Elsewhere the split costs more than it hides
    // (measured): one projection kernel as before.  GS_TUNE_COLOUR_RIDERS = 2 forces the split (A/B).
    c->rider.on = K == GS_RIDER_K && (c->colourRiders == 2 || (c->colourRiders == 1 && depth_sort_takes_splitters(c, N)));
    // where no rider travels (the LSD depth sort's and the one-workgroup sort's kernels are no hosts): geometry first, then
    // the wave's own colours with the rows of unseen Gaussians left out -- faster than the interleaved one-kernel form at
    // every size measured (2 M garden 0.168 -> 0.129 ms, 300 k 0.035 -> 0.031, 10 k 0.0116 -> 0.0098; same bits).
    // GS_TUNE_COLOUR_RIDERS = 3 forces this form, 0 the interleaved one
    const bool selfColour = K == GS_RIDER_K && !c->rider.on && (c->colourRiders == 3 || c->colourRiders == 1);
    // a view under depth cuts: its cuts in coarse form first (binning.hip), for the kernel to drop the Gaussians that lie beyond
    // all of theirs (block lists and 16 x 16 tiles alike: the context's grid is the grid the cuts are kept on)
    GsCutCoarse cc;
    c->dropBlocks = 0;
    if (c->fwd.cutsActive && c->fwd.cutStore && c->cutSuper && c->superCut && c->dropPerBlock) {
        const int src = launch_cut_super(c, c->fwd.cutStore);
        if (src) return src;
        cc.superCut = c->superCut; cc.superW = cut_super_width(c); cc.dropPerBlock = c->dropPerBlock;
        c->dropBlocks = gs_div_up(N, PROJ_FUSED_THREADS);
    }
    // the rects' row groups (rect_row_groups4) where the kernel trims: 16 x 16 tiles, no block lists, the knob on -- the expansion
    // of this forward then enumerates the groups (binning.hip, expand_kernel<.., true>)
    c->piecesValid = c->trimRects && c->rowGroups && !c->virt.nbx && c->tileW == 16 && c->tileH == 16 && c->tilePieces != nullptr;
    uint4* pieces = c->piecesValid ? c->tilePieces : nullptr;
    const int pflags = (gs_small_depth_sort(N) ? 1 : 0) | (c->trimRects ? 2 : 0);      // bit 0: no depth key for a Gaussian that touches
                                                                                      // no tile; bit 1: GS_TUNE_TRIM_RECTS
    ColourRider a = {};
    a.xyz = xyz; a.fdc = fdc; a.frest = frest; a.packed12 = c->packed12; a.tilesTouched = c->tilesTouched;
    a.cam[0] = cam.cam[0]; a.cam[1] = cam.cam[1]; a.cam[2] = cam.cam[2];
    a.N = N; a.degree = c->degree; a.unit0 = 0; a.units = gs_div_up(N, 64);
    if (c->rider.on) {
        c->rider.args = a;
        c->rider.args.units = 0;
        c->rider.next = 0;
        c->rider.total = gs_div_up(N, 64);
        hipLaunchKernelGGL((proj_fwd_fused_kernel<true, false>), dim3(gs_div_up(N, PROJ_FUSED_THREADS)), dim3(PROJ_FUSED_THREADS),
                           0, c->stream, N, K, c->degree, cam, c->tileW, c->tileH, c->gridW, c->gridH, xyz, fdc, frest,
                           scales, rot, opacity, c->packed12, radii, c->tileRect, c->tilesTouched, c->depthKey[0],
                           c->depthVal[0], c->visPerBlock, c->counters, pflags, a, c->virt, cc, pieces);
    } else if (selfColour)
        hipLaunchKernelGGL((proj_fwd_fused_kernel<true, false, true>), dim3(gs_div_up(N, PROJ_FUSED_THREADS)), dim3(PROJ_FUSED_THREADS),
                           sizeof(float) * (PROJ_FUSED_THREADS / 64) * 64 * GS_RIDER_ROW, c->stream, N, K, c->degree, cam, c->tileW,
                           c->tileH, c->gridW, c->gridH, xyz, fdc, frest, scales, rot, opacity, c->packed12, radii, c->tileRect,
                           c->tilesTouched, c->depthKey[0], c->depthVal[0], c->visPerBlock, c->counters,
                           pflags, a, c->virt, cc, pieces);
    else if (twoPhase)
        hipLaunchKernelGGL((proj_fwd_fused_kernel<true, true>), dim3(gs_div_up(N, PROJ_FUSED_THREADS)), dim3(PROJ_FUSED_THREADS),
                           lds, c->stream, N, K, c->degree, cam, c->tileW, c->tileH, c->gridW, c->gridH, xyz, fdc, frest,
                           scales, rot, opacity, c->packed12, radii, c->tileRect, c->tilesTouched, c->depthKey[0],
                           c->depthVal[0], c->visPerBlock, c->counters, pflags, a, c->virt, cc, pieces);
    else
        hipLaunchKernelGGL((proj_fwd_fused_kernel<false, true>), dim3(gs_div_up(N, PROJ_FUSED_THREADS)), dim3(PROJ_FUSED_THREADS),
                           lds, c->stream, N, K, c->degree, cam, c->tileW, c->tileH, c->gridW, c->gridH, xyz, fdc, frest,
                           scales, rot, opacity, c->packed12, radii, c->tileRect, c->tilesTouched, c->depthKey[0],
                           c->depthVal[0], c->visPerBlock, c->counters, pflags, a, c->virt, cc, pieces);
    c->visBlocks = gs_div_up(N, PROJ_FUSED_THREADS);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

// the colour units no binning kernel has taken along (gs_rider.h): in front of the blend, on the whole chip
__global__ __launch_bounds__(GS_RIDER_WAVES * 64) void colour_rest_kernel(ColourRider r)
{
    colour_rider_block(r, (int)blockIdx.x);
}

int launch_colour_rest(gs_ctx* c)
{
    if (!c->rider.on) return GS_OK;
    c->rider.on = false;
    const int left = c->rider.total - c->rider.next;
    if (left <= 0) return GS_OK;
    ColourRider a = c->rider.args;
    a.unit0 = c->rider.next; a.units = left;
    c->rider.next = c->rider.total;
    hipLaunchKernelGGL(colour_rest_kernel, dim3(rider_blocks(left, GS_RIDER_WAVES * 64)), dim3(GS_RIDER_WAVES * 64), 0, c->stream, a);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_projection_fused_backward(gs_ctx* c, int N, int K, const float* xyz, const float* fdc,
                                     const float* frest, const float* scales, const float* rot,
                                     const float* opacity, const CamParams& cam, float* gXyz, float* gFdc,
                                     float* gFrest, float* gScales, float* gRot, float* gOpacity, bool emitColorCot)
{
    if (N == 0) return GS_OK;
    const size_t lds = sizeof(float) * (PROJ_FUSED_THREADS / 64) * 64 * ((K - 1) * 3 + 1);
    AdamFuse none = {};
    if (!emitColorCot) { none.ovf = c->counters + GS_CNT_OVERFLOW; none.rider = c->overflowRider; }
    if (!emitColorCot)
        hipLaunchKernelGGL(proj_bwd_fused_kernel<0>, dim3(gs_div_up(N, PROJ_FUSED_THREADS)), dim3(PROJ_FUSED_THREADS),
                           lds, c->stream, N, K, c->degree, cam, xyz, fdc, frest, scales, rot, opacity, c->gradAcc16, gXyz,
                           gFdc, gFrest, gScales, gRot, gOpacity, c->gradNormAccum, none);
    else   // data-parallel variant: gFdc receives mg[N,3]
        hipLaunchKernelGGL(proj_bwd_fused_kernel<1>, dim3(gs_div_up(N, PROJ_FUSED_THREADS)), dim3(PROJ_FUSED_THREADS),
                           lds, c->stream, N, K, c->degree, cam, xyz, fdc, frest, scales, rot, opacity, c->gradAcc16, gXyz,
                           gFdc, nullptr, gScales, gRot, gOpacity, c->gradNormAccum, none);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_projection_fused_backward_adam(gs_ctx* c, int N, int K, const float* xyz, const float* fdc,
                                          const float* frest, const float* scales, const float* rot,
                                          const float* opacity, const CamParams& cam, const float* pBase, float* mBase,
                                          float* vBase, const float lr[6], float b1, float b2, float eps, float gscale)
{
    if (N == 0) return GS_OK;
    const size_t lds = sizeof(float) * (PROJ_FUSED_THREADS / 64) * 64 * ((K - 1) * 3 + 1);
    AdamFuse a;
    a.pBase = pBase; a.mBase = mBase; a.vBase = vBase;
    for (int i = 0; i < 6; i++) a.lr[i] = lr[i];
    a.b1 = b1; a.b2 = b2; a.eps = eps; a.gscale = gscale;
    a.gate = c->adamGate;
    hipLaunchKernelGGL(proj_bwd_fused_kernel<2>, dim3(gs_div_up(N, PROJ_FUSED_THREADS)), dim3(PROJ_FUSED_THREADS), lds,
                       c->stream, N, K, c->degree, cam, xyz, fdc, frest, scales, rot, opacity, c->gradAcc16, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, c->gradNormAccum, a);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

// gs_set_gathered_gate: the gathered colour cotangents come as `count` blocks of ccBlockFloats floats, every block's word
// [3 N] = that rank's overflow rider
static long long cc_block_floats(const gs_ctx* c, int N) { return c->ccBlockFloats > 0 ? c->ccBlockFloats : 3LL * N; }
static void fill_gathered_gate(const gs_ctx* c, int N, const float* mgAll, AdamFuse& a)
{
    a.seen = c->gateSeen;
    if (c->ccBlockFloats <= 0) return;
    a.gwWords = reinterpret_cast<const uint32_t*>(mgAll + 3LL * N);
    a.gwStride = c->ccBlockFloats; a.gwCount = c->ccBlockCount; a.gwOut = c->gatheredGateOut;
}

int launch_color_cot(gs_ctx* c, int N, float* out)
{
    if (N == 0) return GS_OK;
    hipLaunchKernelGGL(color_cot_kernel, dim3(gs_div_up(N, 256)), dim3(256), 0, c->stream, N, c->gradAcc16, c->packed12,
                       out, c->counters + GS_CNT_OVERFLOW, c->overflowRider);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_sh_grad_from_views(gs_ctx* c, int N, int K, int R, const float* xyz, const float* mgAll,
                              const float* camCentersHost, float* gFdc, float* gFrest)
{
    if (N == 0) return GS_OK;
    ViewCenters v;
    v.n = R;
    for (int r = 0; r < R; r++) for (int k = 0; k < 3; k++) v.c[r][k] = camCentersHost[r * 3 + k];
    const size_t lds = sizeof(float) * (PROJ_FUSED_THREADS / 64) * 64 * ((K - 1) * 3 + 1);
    AdamFuse none = {};
    fill_gathered_gate(c, N, mgAll, none);
    hipLaunchKernelGGL(sh_grad_from_views_kernel<false>, dim3(gs_div_up(N, PROJ_FUSED_THREADS)), dim3(PROJ_FUSED_THREADS),
                       lds, c->stream, N, K, c->degree, v, xyz, mgAll, cc_block_floats(c, N), gFdc, gFrest, nullptr, nullptr, none);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_sh_grad_from_views_adam(gs_ctx* c, int N, int K, int R, const float* xyz, const float* mgAll,
                                   const float* camCentersHost, const float* fdcParam, const float* frestParam,
                                   const float* pBase, float* mBase, float* vBase, float lrDc, float lrRest, float b1,
                                   float b2, float eps, float gscale)
{
    if (N == 0) return GS_OK;
    ViewCenters v;
    v.n = R;
    for (int r = 0; r < R; r++) for (int k = 0; k < 3; k++) v.c[r][k] = camCentersHost[r * 3 + k];
    const size_t lds = sizeof(float) * (PROJ_FUSED_THREADS / 64) * 64 * ((K - 1) * 3 + 1);
    AdamFuse a = {};
    a.pBase = pBase; a.mBase = mBase; a.vBase = vBase;
    a.lr[1] = lrDc; a.lr[2] = lrRest;
    a.b1 = b1; a.b2 = b2; a.eps = eps; a.gscale = gscale;
    a.gate = c->adamGate;
    fill_gathered_gate(c, N, mgAll, a);
    hipLaunchKernelGGL(sh_grad_from_views_kernel<true>, dim3(gs_div_up(N, PROJ_FUSED_THREADS)), dim3(PROJ_FUSED_THREADS),
                       lds, c->stream, N, K, c->degree, v, xyz, mgAll, cc_block_floats(c, N), nullptr, nullptr, fdcParam, frestParam, a);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_projection_geom_backward(gs_ctx* c, int N, const float* xyz, const float* scales, const float* rot,
                                    const float* opacity, const CamParams& cam, float* gXyz, float* gScales, float* gRot,
                                    float* gOpacity, float* xyzOwn, float* colorCot)
{
    if (N == 0) return GS_OK;
    hipLaunchKernelGGL(proj_bwd_geom_kernel, dim3(gs_div_up(N, 256)), dim3(256), 0, c->stream, N, cam, xyz, scales, rot, opacity,
                       c->gradAcc16, gXyz, gScales, gRot, gOpacity, xyzOwn, c->packed12, colorCot, c->counters + GS_CNT_OVERFLOW,
                       c->overflowRider);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_sh_views_dir_adam(gs_ctx* c, int N, int K, int R, const float* xyz, const float* mgAll, const float* camCentersHost,
                             const float* const* ownXyzHost, const float* fdcParam, const float* frestParam, const float* pBase,
                             float* mBase, float* vBase, float lrDc, float lrRest, float b1, float b2, float eps, float gscale,
                             float* xyzAdd)
{
    if (N == 0) return GS_OK;
    ViewCentersOwn v;
    v.n = R;
    for (int r = 0; r < 16; r++) v.own[r] = nullptr;
    for (int r = 0; r < R; r++) {
        for (int k = 0; k < 3; k++) v.c[r][k] = camCentersHost[r * 3 + k];
        v.own[r] = ownXyzHost ? ownXyzHost[r] : nullptr;
    }
    const size_t lds = sizeof(float) * (PROJ_FUSED_THREADS / 64) * 64 * ((K - 1) * 3 + 1);
    AdamFuse a = {};
    a.pBase = pBase; a.mBase = mBase; a.vBase = vBase;
    a.lr[1] = lrDc; a.lr[2] = lrRest;
    a.b1 = b1; a.b2 = b2; a.eps = eps; a.gscale = gscale;
    a.gate = c->adamGate;
    fill_gathered_gate(c, N, mgAll, a);
    hipLaunchKernelGGL(sh_views_dir_adam_kernel, dim3(gs_div_up(N, PROJ_FUSED_THREADS)), dim3(PROJ_FUSED_THREADS), lds, c->stream, N,
                       K, c->degree, v, xyz, mgAll, cc_block_floats(c, N), fdcParam, frestParam, xyzAdd, c->gradNormAccum, a);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_pack11_to_12(gs_ctx* c, int N, const float* packed11)
{
    if (N == 0) return GS_OK;
    hipLaunchKernelGGL(pack11_to_12_kernel, dim3(gs_div_up((long long)N * 12, 256)), dim3(256), 0, c->stream, N,
                       packed11, c->packed12);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_pack_gaussians(gs_ctx* c, int N, const float* means2d, const float* conic, const float* color,
                          const float* opacity, const float* depths, float* packed11)
{
    if (N == 0) return GS_OK;
    hipLaunchKernelGGL(pack_gaussians_kernel, dim3(gs_div_up(N, 256)), dim3(256), 0, c->stream, N, means2d, conic,
                       color, opacity, depths, packed11);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

}  // namespace gs
