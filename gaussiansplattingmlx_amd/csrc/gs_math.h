// gs_math.h -- per-Gaussian device math shared by the projection kernels.
// Restates slang/gaussian_projection_screen_shared.slang (reference) in f32;
// the tie conventions of the reverse mode follow the reference's expanded
// autodiff (max tie -> 1/2, clamp passes on the closed interval,
// d sqrt(x) = 1/2 / sqrt(max(1e-7, x))).
//
// This translation unit is compiled with -ffp-contract=off so that the
// mean/rect arithmetic (mul/add/div/sqrt only) is bit-identical to an IEEE
// CPU evaluation: tile membership is integer work and must not flip on a
// fused-multiply-add rounding difference.
#pragma once
#include <hip/hip_runtime.h>

namespace gs {

struct CamParams {
    float V[16];
    float P[16];
    float cam[3];
    float fovX, fovY, focalX, focalY;
    float limX, limY;  // 1.3 * tan(fov/2), tan evaluated on the host in f32
    float W, H;
};

__device__ __forceinline__ float d_max_left(float a, float b, float g)
{
    return a > b ? g : (a < b ? 0.0f : 0.5f * g);
}

// ---- spherical harmonics, reference constants (shared.slang:269-311) -------
#define GS_C0 0.28209479177387814f
#define GS_C1 0.4886025119029199f
#define GS_C2A 1.0925484305920792f
#define GS_C2C 0.31539156525252005f
#define GS_C2E 0.5462742152960396f
#define GS_C3A 0.5900435899266435f
#define GS_C3B 2.890611442640554f
#define GS_C3C 0.4570457994644658f
#define GS_C3D 0.3731763325901154f
#define GS_C3E 1.445305721320277f
#define GS_C4A 2.5033429417967046f
#define GS_C4B 1.7701307697799304f
#define GS_C4C 0.9461746957575601f
#define GS_C4D 0.6690465435572892f
#define GS_C4E 0.10578554691520431f
#define GS_C4F 0.47308734787878004f
#define GS_C4G 0.6258357354491761f

// Calls f(k, basis_k, d basis_k/dx, d/dy, d/dz) for k < (degree+1)^2 in index
// order; the gradient expressions are dead code in callers that ignore them.
template <class F>
__device__ __forceinline__ void sh_foreach(int degree, float x, float y, float z, F&& f)
{
    f(0, GS_C0, 0.f, 0.f, 0.f);
    if (degree <= 0) return;
    f(1, -GS_C1 * y, 0.f, -GS_C1, 0.f);
    f(2, GS_C1 * z, 0.f, 0.f, GS_C1);
    f(3, -GS_C1 * x, -GS_C1, 0.f, 0.f);
    if (degree <= 1) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    f(4, GS_C2A * xy, GS_C2A * y, GS_C2A * x, 0.f);
    f(5, -GS_C2A * yz, 0.f, -GS_C2A * z, -GS_C2A * y);
    f(6, GS_C2C * (2.0f * zz - xx - yy), GS_C2C * (-2.0f * x), GS_C2C * (-2.0f * y), GS_C2C * (4.0f * z));
    f(7, -GS_C2A * xz, -GS_C2A * z, 0.f, -GS_C2A * x);
    f(8, GS_C2E * (xx - yy), GS_C2E * 2.0f * x, -GS_C2E * 2.0f * y, 0.f);
    if (degree <= 2) return;
    f(9, -GS_C3A * y * (3.0f * xx - yy), -GS_C3A * 6.0f * xy, -GS_C3A * (3.0f * xx - 3.0f * yy), 0.f);
    f(10, GS_C3B * xy * z, GS_C3B * yz, GS_C3B * xz, GS_C3B * xy);
    f(11, -GS_C3C * y * (4.0f * zz - xx - yy), -GS_C3C * (-2.0f * xy), -GS_C3C * (4.0f * zz - xx - 3.0f * yy),
      -GS_C3C * (8.0f * yz));
    f(12, GS_C3D * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), GS_C3D * (-6.0f * xz), GS_C3D * (-6.0f * yz),
      GS_C3D * (6.0f * zz - 3.0f * xx - 3.0f * yy));
    f(13, -GS_C3C * x * (4.0f * zz - xx - yy), -GS_C3C * (4.0f * zz - 3.0f * xx - yy), -GS_C3C * (-2.0f * xy),
      -GS_C3C * (8.0f * xz));
    f(14, GS_C3E * z * (xx - yy), GS_C3E * 2.0f * xz, -GS_C3E * 2.0f * yz, GS_C3E * (xx - yy));
    f(15, -GS_C3A * x * (xx - 3.0f * yy), -GS_C3A * (3.0f * xx - 3.0f * yy), -GS_C3A * (-6.0f * xy), 0.f);
    if (degree <= 3) return;
    const float s7 = 7.0f * zz;
    f(16, GS_C4A * xy * (xx - yy), GS_C4A * (3.0f * xx * y - yy * y), GS_C4A * (xx * x - 3.0f * x * yy), 0.f);
    f(17, -GS_C4B * yz * (3.0f * xx - yy), -GS_C4B * (6.0f * xy * z), -GS_C4B * (3.0f * xx * z - 3.0f * yy * z),
      -GS_C4B * (3.0f * xx * y - yy * y));
    f(18, GS_C4C * xy * (s7 - 1.0f), GS_C4C * y * (s7 - 1.0f), GS_C4C * x * (s7 - 1.0f), GS_C4C * 14.0f * xy * z);
    f(19, -GS_C4D * yz * (s7 - 3.0f), 0.f, -GS_C4D * z * (s7 - 3.0f), -GS_C4D * y * (21.0f * zz - 3.0f));
    f(20, GS_C4E * (zz * (35.0f * zz - 30.0f) + 3.0f), 0.f, 0.f, GS_C4E * (140.0f * zz * z - 60.0f * z));
    f(21, -GS_C4D * xz * (s7 - 3.0f), -GS_C4D * z * (s7 - 3.0f), 0.f, -GS_C4D * x * (21.0f * zz - 3.0f));
    f(22, GS_C4F * (xx - yy) * (s7 - 1.0f), GS_C4F * 2.0f * x * (s7 - 1.0f), -GS_C4F * 2.0f * y * (s7 - 1.0f),
      GS_C4F * 14.0f * z * (xx - yy));
    f(23, -GS_C4B * xz * (xx - 3.0f * yy), -GS_C4B * (3.0f * xx * z - 3.0f * yy * z), -GS_C4B * (-6.0f * xy * z),
      -GS_C4B * (xx * x - 3.0f * x * yy));
    f(24, GS_C4G * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy)), GS_C4G * (4.0f * xx * x - 12.0f * x * yy),
      GS_C4G * (-12.0f * xx * y + 4.0f * yy * y), 0.f);
}

// ---- 3-D covariance (shared.slang:117-168) ----------------------------------
struct RotCtx {
    float qw, qx, qy, qz, safeNorm, norm, n2;
    float r[9];
};

__device__ __forceinline__ void build_cov3d(const float s[3], const float rq[4], float c[9], RotCtx& rc)
{
    // Sigma = A A^T with A = R(q / max(|q|, 1e-8)) diag(s).  The order of every sum is the reference's (left to right over the
    // axis index): radius and tile rect are derived from this covariance and are held bit-exact against the oracle.
    const float n2 = rq[0] * rq[0] + rq[1] * rq[1] + rq[2] * rq[2] + rq[3] * rq[3];
    const float norm = sqrtf(n2);
    const float safeNorm = norm > 1e-8f ? norm : 1e-8f;
    const float qw = rq[0] / safeNorm, qx = rq[1] / safeNorm, qy = rq[2] / safeNorm, qz = rq[3] / safeNorm;
    float R[3][3];
    R[0][0] = 1.0f - 2.0f * (qy * qy + qz * qz);
    R[0][1] = 2.0f * (qx * qy - qw * qz);
    R[0][2] = 2.0f * (qx * qz + qw * qy);
    R[1][0] = 2.0f * (qx * qy + qw * qz);
    R[1][1] = 1.0f - 2.0f * (qx * qx + qz * qz);
    R[1][2] = 2.0f * (qy * qz - qw * qx);
    R[2][0] = 2.0f * (qx * qz - qw * qy);
    R[2][1] = 2.0f * (qy * qz + qw * qx);
    R[2][2] = 1.0f - 2.0f * (qx * qx + qy * qy);
    float A[3][3];          // the scaled axes: column k is the k-th principal axis times its extent
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) A[i][k] = R[i][k] * s[k];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float acc = A[i][0] * A[j][0];
            acc = acc + A[i][1] * A[j][1];
            acc = acc + A[i][2] * A[j][2];
            c[3 * i + j] = acc;
        }
    rc.qw = qw; rc.qx = qx; rc.qy = qy; rc.qz = qz;
    rc.safeNorm = safeNorm; rc.norm = norm; rc.n2 = n2;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) rc.r[3 * i + k] = R[i][k];
}

// ---- EWA 2-D covariance (shared.slang:170-243), z-clamp quirk included ------
struct Cov2Ctx {
    float t0, t1, t2, clipX, clipY, tx, ty;
    float b[6];
    float t[6];
};

__device__ __forceinline__ void build_cov2d(const float m[3], const float c[9], const CamParams& cam, float out[4],
                                            Cov2Ctx& cc)
{
    // Sigma' = (J W) Sigma (J W)^T + 0.3 I, J the perspective Jacobian at the view-space point with the reference's clamp of the
    // depth against the frustum limits (its "x / clamp(z) * z" quirk included), W the view rotation (row-vector convention:
    // W[r][c] = V[c][r]).  Sums in the reference's order, as in build_cov3d.
    const float* V = cam.V;
    float pv[3];            // the mean in view space
#pragma unroll
    for (int a = 0; a < 3; a++) pv[a] = m[0] * V[a] + m[1] * V[4 + a] + m[2] * V[8 + a] + V[12 + a];
    const float depth = pv[2];
    const float zLimX = depth < -cam.limX ? -cam.limX : (depth > cam.limX ? cam.limX : depth);
    const float zLimY = depth < -cam.limY ? -cam.limY : (depth > cam.limY ? cam.limY : depth);
    const float xs = pv[0] / zLimX * depth;
    const float ys = pv[1] / zLimY * depth;
    const float jxx = cam.focalX / depth;
    const float jxz = -xs * cam.focalX / (depth * depth);
    const float jyy = cam.focalY / depth;
    const float jyz = -ys * cam.focalY / (depth * depth);
    float JW[2][3];         // J W: row 0 from (jxx, 0, jxz), row 1 from (0, jyy, jyz)
#pragma unroll
    for (int k = 0; k < 3; k++) {
        JW[0][k] = jxx * V[4 * k] + jxz * V[4 * k + 2];
        JW[1][k] = jyy * V[4 * k + 1] + jyz * V[4 * k + 2];
    }
    float M[2][3];          // (J W) Sigma
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float acc = JW[r][0] * c[k];
            acc = acc + JW[r][1] * c[3 + k];
            acc = acc + JW[r][2] * c[6 + k];
            M[r][k] = acc;
        }
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            float acc = M[r][0] * JW[q][0];
            acc = acc + M[r][1] * JW[q][1];
            acc = acc + M[r][2] * JW[q][2];
            out[2 * r + q] = r == q ? acc + 0.3f : acc;
        }
    cc.t0 = pv[0]; cc.t1 = pv[1]; cc.t2 = depth; cc.clipX = zLimX; cc.clipY = zLimY; cc.tx = xs; cc.ty = ys;
#pragma unroll
    for (int k = 0; k < 3; k++) { cc.b[k] = JW[0][k]; cc.b[3 + k] = JW[1][k]; cc.t[k] = M[0][k]; cc.t[3 + k] = M[1][k]; }
}

struct ProjOut {
    float sx, sy, depth;
    float cov2d[4], conic[4];
    float radius;  // already multiplied by the visibility mask
    float rect[4]; // minX minY maxX maxY
};

// Everything of the forward except colour (kernels.slang:47-63, 91-172).
__device__ __forceinline__ void project_geometry(const float m[3], const float s[3], const float rq[4],
                                                 const CamParams& cam, ProjOut& o)
{
    const float* V = cam.V;
    const float* P = cam.P;
    const float pv0 = m[0] * V[0] + m[1] * V[4] + m[2] * V[8] + V[12];
    const float pv1 = m[0] * V[1] + m[1] * V[5] + m[2] * V[9] + V[13];
    const float pv2 = m[0] * V[2] + m[1] * V[6] + m[2] * V[10] + V[14];
    const float pv3 = m[0] * V[3] + m[1] * V[7] + m[2] * V[11] + V[15];
    const float pc0 = pv0 * P[0] + pv1 * P[4] + pv2 * P[8] + pv3 * P[12];
    const float pc1 = pv0 * P[1] + pv1 * P[5] + pv2 * P[9] + pv3 * P[13];
    const float pc3 = pv0 * P[3] + pv1 * P[7] + pv2 * P[11] + pv3 * P[15];
    const float wInv = 1.0f / (pc3 + 0.000001f);
    const float ndcX = pc0 * wInv, ndcY = pc1 * wInv;
    o.sx = ((ndcX + 1.0f) * cam.W - 1.0f) * 0.5f;
    o.sy = ((ndcY + 1.0f) * cam.H - 1.0f) * 0.5f;
    o.depth = pv2;
    const float visibleMask = (pv2 >= 0.2f) ? 1.0f : 0.0f;
    float c3[9];
    RotCtx rc;
    build_cov3d(s, rq, c3, rc);
    Cov2Ctx cc;
    build_cov2d(m, c3, cam, o.cov2d, cc);
    const float det = o.cov2d[0] * o.cov2d[3] - o.cov2d[1] * o.cov2d[2];
    o.conic[0] = o.cov2d[3] / det;
    o.conic[1] = -o.cov2d[1] / det;
    o.conic[2] = -o.cov2d[2] / det;
    o.conic[3] = o.cov2d[0] / det;
    const float mid = 0.5f * (o.cov2d[0] + o.cov2d[3]);
    float delta = mid * mid - det;
    if (!(delta > 1e-5f)) delta = 1e-5f;
    const float lambdaMax = mid + sqrtf(delta);
    o.radius = 3.0f * ceilf(sqrtf(lambdaMax)) * visibleMask;
    const float maxX = cam.W - 1.0f, maxY = cam.H - 1.0f;
    float minX = o.sx - o.radius, minY = o.sy - o.radius, maxRX = o.sx + o.radius, maxRY = o.sy + o.radius;
    if (minX < 0.0f) minX = 0.0f;
    if (minY < 0.0f) minY = 0.0f;
    if (maxRX > maxX) maxRX = maxX;
    if (maxRY > maxY) maxRY = maxY;
    o.rect[0] = minX; o.rect[1] = minY; o.rect[2] = maxRX; o.rect[3] = maxRY;
}

// Tile rectangle of a splat (gaussian_tile_global_kernels.slang:39-55).
__device__ __forceinline__ void tile_rect(float rMinX, float rMinY, float rMaxX, float rMaxY, int tileW, int tileH,
                                          int gridW, int gridH, int& x0, int& y0, int& x1, int& y1)
{
    x0 = (int)floorf(rMinX / (float)tileW);
    y0 = (int)floorf(rMinY / (float)tileH);
    x1 = (int)floorf(rMaxX / (float)tileW) + 1;
    y1 = (int)floorf(rMaxY / (float)tileH) + 1;
    x0 = max(0, min(x0, gridW)); y0 = max(0, min(y0, gridH));
    x1 = max(0, min(x1, gridW)); y1 = max(0, min(y1, gridH));
}

// Block rectangle of a splat under block lists (gs_ctx.h, GsVirtGeom): the blocks, in the grid of 16 x 16 blocks enumerated per
// tile, that lie in a tile of the splat's tile rectangle (the reference's membership rule, tile_rect above) AND hold a pixel
// within reach -- a pixel where the weight exp(-q/2) can be 2^-29 or more, the bound below which the blend kernels drop an
// entry for a whole quadrant anyway (GS_CULL_QMIN, gs_cull.h).  Reach is the axis-aligned box of the ellipse q <= QC,
// half-widths sqrt(QC cov_xx), sqrt(QC cov_yy), with a margin of a pixel and a part in a thousand (q is evaluated from the
// conic in float32); a covariance that is not finite reaches everything.  The blocks of a tile are counted through its
// nbx x nby grid: pixel column x lies in block column (x / tw) nbx + (x mod tw) / 16.
__device__ __forceinline__ void block_rect_of_splat(const float rect[4], float sx, float sy, float covxx, float covyy,
                                                    int nbx, int nby, int tw, int th, int gridWr, int gridHr, int W, int H,
                                                    int& x0, int& y0, int& x1, int& y1)
{
    constexpr float QC = 40.3f;                 // 2 * 29 * ln 2 = 40.20
    int tx0, ty0, tx1, ty1;
    tile_rect(rect[0], rect[1], rect[2], rect[3], tw, th, gridWr, gridHr, tx0, ty0, tx1, ty1);
    x0 = y0 = x1 = y1 = 0;
    if (tx1 <= tx0 || ty1 <= ty0) return;
    float hx = sqrtf(QC * covxx) * 1.001f + 1.0f, hy = sqrtf(QC * covyy) * 1.001f + 1.0f;
    if (!(hx < 1e8f)) hx = 1e8f;
    if (!(hy < 1e8f)) hy = 1e8f;
    const float fxl = fminf(fmaxf(ceilf(sx - hx), (float)(tx0 * tw)), 1e9f), fxh = fmaxf(fminf(floorf(sx + hx), (float)(min(tx1 * tw, W) - 1)), -1e9f);
    const float fyl = fminf(fmaxf(ceilf(sy - hy), (float)(ty0 * th)), 1e9f), fyh = fmaxf(fminf(floorf(sy + hy), (float)(min(ty1 * th, H) - 1)), -1e9f);
    if (!(fxl <= fxh) || !(fyl <= fyh)) return;
    const int pxl = (int)fxl, pxh = (int)fxh, pyl = (int)fyl, pyh = (int)fyh;
    const int txl = pxl / tw, txh = pxh / tw, tyl = pyl / th, tyh = pyh / th;
    x0 = txl * nbx + (pxl - txl * tw) / 16; x1 = txh * nbx + (pxh - txh * tw) / 16 + 1;
    y0 = tyl * nby + (pyl - tyl * th) / 16; y1 = tyh * nby + (pyh - tyh * th) / 16 + 1;
}

// Row groups of a trimmed rect (round 6; GS_TUNE_TRIM_RECTS at 16 x 16 tiles).  The box of the ellipse q <= QC still holds the
// corners an elongated, tilted splat never reaches.  The rect's tile rows [y0, y1) are cut into FOUR groups -- group p = rows
// y0 + (h p >> 2) .. y0 + (h (p + 1) >> 2), h = y1 - y0 -- and each group keeps only the tile columns the ellipse reaches on the
// group's pixel rows: with the covariance (sxx, sxy, syy) the ellipse's slice at dy has its centre at (sxy / syy) dy and the
// half-width sqrt((sxx - sxy^2 / syy)(QC - dy^2 / syy)) -- concave in dy, so its extreme over an interval of dy is at the
// ellipse's right- / leftmost point (dy = +-(sxy / sxx) hx) when that lies inside and at an end of the interval otherwise.
// Margins as in block_rect_of_splat (a pixel and a part in a thousand; the interval of dy a pixel longer at both ends).
// out[p] = first column | columns << 16; returns the pairs of the four groups together (c3: 14 % fewer than the box's).
// Anything not finite keeps the whole box.
__device__ __forceinline__ uint32_t rect_row_groups4(float sx, float sy, float sxx, float sxy, float syy, int x0, int y0, int x1,
                                                     int y1, int H, uint32_t out[4])
{
    constexpr float QC = 40.3f;
    const int h = y1 - y0, wFull = x1 - x0;
    if (h < 2 || wFull < 3) {       // (nothing to cut: most rects of a trained scene are this small, and the groups cost ~100 instructions)
#pragma unroll
        for (int p = 0; p < 4; p++) out[p] = (uint32_t)x0 | ((uint32_t)wFull << 16);
        return (uint32_t)(wFull * h);
    }
    // (v_sqrt_f32 / v_rcp_f32: a unit in the last place each, inside the margins below)
    const float hxe = __builtin_amdgcn_sqrtf(QC * sxx), hye = __builtin_amdgcn_sqrtf(QC * syy), isyy = __builtin_amdgcn_rcpf(syy);
    const float slope = sxy * isyy, vc = fmaxf(sxx - sxy * slope, 0.0f), dyR = sxy * __builtin_amdgcn_rcpf(sxx) * hxe;
    const float mar = 1.0f + 0.001f * hxe;
    const bool finite = (hxe < 1e8f) && (hye < 1e8f) && (slope == slope) && (dyR == dyR) && fabsf(slope) < 1e8f;
    // the box's pixel rows (block_rect_of_splat)
    float hyM = hye * 1.001f + 1.0f;
    if (!(hyM < 1e8f)) hyM = 1e8f;
    const float fyl = fmaxf(ceilf(sy - hyM), (float)(y0 * 16)), fyh = fminf(floorf(sy + hyM), (float)(min(y1 * 16, H) - 1));
    uint32_t total = 0;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int r0 = y0 + ((h * p) >> 2), r1 = y0 + ((h * (p + 1)) >> 2);
        int ta = x0, tb = x1 - 1;
        if (finite && r1 > r0) {
            const float ya = fmaxf((float)(16 * r0), fyl), yb = fminf((float)(16 * r1 - 1), fyh);
            const float d0 = fmaxf(ya - sy - 1.0f, -hye), d1 = fminf(yb - sy + 1.0f, hye);
            if (d0 <= d1) {
                const float s0 = __builtin_amdgcn_sqrtf(fmaxf(vc * (QC - d0 * d0 * isyy), 0.0f));
                const float s1 = __builtin_amdgcn_sqrtf(fmaxf(vc * (QC - d1 * d1 * isyy), 0.0f));
                const float xmax = (d0 <= dyR && dyR <= d1) ? hxe : fmaxf(slope * d0 + s0, slope * d1 + s1);
                const float xmin = (d0 <= -dyR && -dyR <= d1) ? -hxe : fminf(slope * d0 - s0, slope * d1 - s1);
                const float xa = fminf(fmaxf(floorf((sx + xmin - mar) * 0.0625f), -1e9f), 1e9f);
                const float xb = fminf(fmaxf(floorf((sx + xmax + mar) * 0.0625f), -1e9f), 1e9f);
                if (xa == xa && xb == xb) { ta = max(x0, (int)xa); tb = min(x1 - 1, (int)xb); }
            } else tb = ta - 1;          // (rows the ellipse does not reach at all)
        }
        const int w = max(tb - ta + 1, 0);
        out[p] = (uint32_t)(w > 0 ? ta : x0) | ((uint32_t)min(w, wFull) << 16);
        total += (uint32_t)(w * (r1 - r0));
    }
    return total;
}

struct GeomGrads {
    float dm[3], ds[3], dq[4];
};

// Reverse mode of the geometry part: cotangents of means2d, depth, cov2d, conic
// -> gradients of means3d, scales, rotations (activated values).
__device__ __forceinline__ void project_geometry_bwd(const float m[3], const float s[3], const float rq[4],
                                                     const CamParams& cam, const float cotM2d[2], float cotDepth,
                                                     const float cotCov[4], const float cotCon[4], GeomGrads& g)
{
    const float* V = cam.V;
    const float* P = cam.P;
    float c3[9];
    RotCtx rc;
    build_cov3d(s, rq, c3, rc);
    float c2[4];
    Cov2Ctx cc;
    build_cov2d(m, c3, cam, c2, cc);

    // inverse 2x2 (four independent entries)
    const float det = c2[0] * c2[3] - c2[1] * c2[2];
    const float det2 = det * det;
    const float S29 = cotCon[3] / det2, S30 = cotCon[2] / det2, S31 = cotCon[1] / det2, S32 = cotCon[0] / det2;
    const float S33 = c2[0] * -S29 + -c2[2] * -S30 + -c2[1] * -S31 + c2[3] * -S32;
    const float S34 = -S33;
    float dC[4];
    dC[2] = -(det * S30) + c2[1] * S34;
    dC[1] = -(det * S31) + c2[2] * S34;
    dC[3] = det * S32 + c2[0] * S33;
    dC[0] = det * S29 + c2[3] * S33;
#pragma unroll
    for (int k = 0; k < 4; k++) dC[k] += cotCov[k];

    const float* b = cc.b;
    const float* t = cc.t;
    float dt[6], db[6];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        dt[k] = dC[0] * b[k] + dC[1] * b[3 + k];
        dt[3 + k] = dC[2] * b[k] + dC[3] * b[3 + k];
        db[k] = dC[0] * t[k] + dC[2] * t[3 + k];
        db[3 + k] = dC[1] * t[k] + dC[3] * t[3 + k];
    }
    float dS[9];
#pragma unroll
    for (int l = 0; l < 3; l++) {
#pragma unroll
        for (int k = 0; k < 3; k++) dS[l * 3 + k] = b[l] * dt[k] + b[3 + l] * dt[3 + k];
        db[l] += dt[0] * c3[l * 3 + 0] + dt[1] * c3[l * 3 + 1] + dt[2] * c3[l * 3 + 2];
        db[3 + l] += dt[3] * c3[l * 3 + 0] + dt[4] * c3[l * 3 + 1] + dt[5] * c3[l * 3 + 2];
    }
    float dj00 = 0.f, dj02 = 0.f, dj11 = 0.f, dj12 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        dj00 += db[k] * V[k * 4 + 0];
        dj02 += db[k] * V[k * 4 + 2];
        dj11 += db[3 + k] * V[k * 4 + 1];
        dj12 += db[3 + k] * V[k * 4 + 2];
    }
    const float tz = cc.t2, tz2 = tz * tz;
    float dtz = -cam.focalX / tz2 * dj00 - cam.focalY / tz2 * dj11;
    const float dtx = -cam.focalX / tz2 * dj02;
    const float dty = -cam.focalY / tz2 * dj12;
    const float dtz2 = cc.tx * cam.focalX / (tz2 * tz2) * dj02 + cc.ty * cam.focalY / (tz2 * tz2) * dj12;
    dtz += 2.0f * tz * dtz2;
    const float ux = cc.t0 / cc.clipX, uy = cc.t1 / cc.clipY;
    float dt2 = dtz + ux * dtx + uy * dty;
    const float dux = cc.t2 * dtx, duy = cc.t2 * dty;
    const float dt0 = dux / cc.clipX, dt1 = duy / cc.clipY;
    const float dclipX = -cc.t0 / (cc.clipX * cc.clipX) * dux;
    const float dclipY = -cc.t1 / (cc.clipY * cc.clipY) * duy;
    if (cc.t2 >= -cam.limX && cc.t2 <= cam.limX) dt2 += dclipX;
    if (cc.t2 >= -cam.limY && cc.t2 <= cam.limY) dt2 += dclipY;
#pragma unroll
    for (int a = 0; a < 3; a++) g.dm[a] = V[a * 4 + 0] * dt0 + V[a * 4 + 1] * dt1 + V[a * 4 + 2] * dt2;

    // screen -> ndc -> clip -> view -> point
    const float pv0 = m[0] * V[0] + m[1] * V[4] + m[2] * V[8] + V[12];
    const float pv1 = m[0] * V[1] + m[1] * V[5] + m[2] * V[9] + V[13];
    const float pv2 = m[0] * V[2] + m[1] * V[6] + m[2] * V[10] + V[14];
    const float pv3 = m[0] * V[3] + m[1] * V[7] + m[2] * V[11] + V[15];
    const float pc0 = pv0 * P[0] + pv1 * P[4] + pv2 * P[8] + pv3 * P[12];
    const float pc1 = pv0 * P[1] + pv1 * P[5] + pv2 * P[9] + pv3 * P[13];
    const float pc3 = pv0 * P[3] + pv1 * P[7] + pv2 * P[11] + pv3 * P[15];
    const float wInv = 1.0f / (pc3 + 0.000001f);
    const float dndcX = cotM2d[0] * 0.5f * cam.W, dndcY = cotM2d[1] * 0.5f * cam.H;
    const float dpc0 = dndcX * wInv, dpc1 = dndcY * wInv;
    const float dwInv = pc0 * dndcX + pc1 * dndcY;
    const float dpc3 = -dwInv * wInv * wInv;
    float dpv[4];
#pragma unroll
    for (int i = 0; i < 4; i++) dpv[i] = P[i * 4 + 0] * dpc0 + P[i * 4 + 1] * dpc1 + P[i * 4 + 3] * dpc3;
    dpv[2] += cotDepth;
#pragma unroll
    for (int a = 0; a < 3; a++)
        g.dm[a] += V[a * 4 + 0] * dpv[0] + V[a * 4 + 1] * dpv[1] + V[a * 4 + 2] * dpv[2] + V[a * 4 + 3] * dpv[3];

    // Sigma = L L^T, L = R diag(s)
    float L[9], dL[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) L[i * 3 + j] = rc.r[i * 3 + j] * s[j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 3; k++) acc += (dS[i * 3 + k] + dS[k * 3 + i]) * L[k * 3 + j];
            dL[i * 3 + j] = acc;
        }
    float dr[9];
    g.ds[0] = g.ds[1] = g.ds[2] = 0.f;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            g.ds[j] += dL[i * 3 + j] * rc.r[i * 3 + j];
            dr[i * 3 + j] = dL[i * 3 + j] * s[j];
        }
    const float qw = rc.qw, qx = rc.qx, qy = rc.qy, qz = rc.qz;
    const float dqw = 2.0f * (-qz * dr[1] + qy * dr[2] + qz * dr[3] - qx * dr[5] - qy * dr[6] + qx * dr[7]);
    const float dqx = 2.0f * (qy * dr[1] + qz * dr[2] + qy * dr[3] - qw * dr[5] + qz * dr[6] + qw * dr[7]) -
                      4.0f * qx * (dr[4] + dr[8]);
    const float dqy = 2.0f * (qx * dr[1] + qw * dr[2] + qx * dr[3] + qz * dr[5] - qw * dr[6] + qz * dr[7]) -
                      4.0f * qy * (dr[0] + dr[8]);
    const float dqz = 2.0f * (-qw * dr[1] + qx * dr[2] + qw * dr[3] + qy * dr[5] + qx * dr[6] + qy * dr[7]) -
                      4.0f * qz * (dr[0] + dr[4]);
    const float sn = rc.safeNorm;
    const float dsafe = -(dqw * rq[0] + dqx * rq[1] + dqy * rq[2] + dqz * rq[3]) / (sn * sn);
    const float dnorm = d_max_left(rc.norm, 1e-8f, dsafe);
    const float mx = rc.n2 > 1e-7f ? rc.n2 : 1e-7f;
    const float dn2 = 0.5f / sqrtf(mx) * dnorm;
    g.dq[0] = dqw / sn + 2.0f * rq[0] * dn2;
    g.dq[1] = dqx / sn + 2.0f * rq[1] * dn2;
    g.dq[2] = dqy / sn + 2.0f * rq[2] * dn2;
    g.dq[3] = dqz / sn + 2.0f * rq[3] * dn2;
}

}  // namespace gs
