/*
 * gs_oracle.c -- CPU restatement of the reference's render/backward hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (gaussiansplattingmlx_amd/,
 * include/) may import, link or call this file.  Only tests/, the smoke check
 * in __graft_entry__.py and the `cpu_baseline` leg of bench.py use it, and
 * there only as the checker / reported host baseline.
 *
 * Parity status: the reference (Swift + MLX + Slang->Metal) cannot be built
 * or run in this image (no swift, Metal, slangc or MLX; compiling its Metal
 * source would need hand-written stand-ins for <metal_stdlib>).  This file
 * restates the algorithm from the reference's sources, function by function,
 * citing file:line under /root/reference.  It is pinned by
 *   - the reference tests' fixtures that touch this path (SH polynomials,
 *     quaternion->rotation convention),
 *   - the known-answer values recorded from the reference's kernels in
 *     SURVEY.md Appendix C (3-6 significant digits),
 *   - float64 finite differences of its own forward (the reference backward is
 *     autodiff of the forward, so a backward that matches finite differences
 *     plus the tie conventions below is the reference backward).
 * Beyond those pins parity is UNPINNED.
 *
 * Build:  gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC   (float build)
 *         add -DGSO_DOUBLE for the float64 build used by gradient checks.
 * All arithmetic is done in `real` in the reference's expression order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef GSO_DOUBLE
typedef double real;
#define R(x) x
#define r_exp exp
#define r_sqrt sqrt
#define r_tan tan
#define r_floor floor
#define r_ceil ceil
#define r_fabs fabs
#define GSO_NAME(n) gso64_##n
#else
typedef float real;
#define R(x) x##f
#define r_exp expf
#define r_sqrt sqrtf
#define r_tan tanf
#define r_floor floorf
#define r_ceil ceilf
#define r_fabs fabsf
#define GSO_NAME(n) gso_##n
#endif

#define GSO_API __attribute__((visibility("default")))

/* ---- autodiff tie conventions (expanded Metal in
 * GaussianSplattingMlx/Slang/gaussian_projection_screen_fused_backward_mlx.json
 * `header`: _d_max_0, _d_sqrt_0, _d_clamp_0) -------------------------------- */
static inline real d_max_left(real a, real b, real g)
{
    return a > b ? g : (a < b ? R(0.0) : R(0.5) * g);
}
static inline real d_sqrt(real x, real g)
{
    real m = x > R(1e-7) ? x : R(1e-7);
    return R(0.5) / r_sqrt(m) * g;
}

/* ======================================================================== *
 * a1  Camera  (Trainer/CameraUtil.swift:5-102, Trainer/simd+ext.swift:45-66)
 * ======================================================================== */
static int invert4(const double m[16], double inv[16])
{
    double a[4][8];
    for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 4; j++) { a[i][j] = m[i * 4 + j]; a[i][4 + j] = (i == j); }
    }
    for (int c = 0; c < 4; c++) {
        int p = c;
        for (int r = c + 1; r < 4; r++) if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
        if (a[p][c] == 0.0) return -1;
        if (p != c) for (int j = 0; j < 8; j++) { double t = a[c][j]; a[c][j] = a[p][j]; a[p][j] = t; }
        double d = a[c][c];
        for (int j = 0; j < 8; j++) a[c][j] /= d;
        for (int r = 0; r < 4; r++) if (r != c) {
            double f = a[r][c];
            for (int j = 0; j < 8; j++) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) inv[i * 4 + j] = a[i][4 + j];
    return 0;
}

/* c2w row-major 4x4 (as the reference's MLXArray), focal in pixels.
 * view = (c2w^-1)^T stored row-major so that p_view = [p,1] . view
 * (CameraUtil.swift:30,34); proj rows as CameraUtil.swift:82-101 transposed
 * (:35); FoV = 2 atan(pixels / (2 focal)) evaluated in f32 (:77-79 on
 * MLXArray); matrices built in f64 then cast (simd+ext.swift:45-55). */
GSO_API int gso_camera_build(const double c2w[16], float focalX, float focalY, int W, int H,
                             double znear, double zfar, float view[16], float proj[16],
                             float* fovX, float* fovY, float camCenter[3])
{
    double inv[16];
    if (invert4(c2w, inv) != 0) return -1;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) view[r * 4 + c] = (float)inv[c * 4 + r];
    float fx = 2.0f * atanf((float)W / (2.0f * focalX));
    float fy = 2.0f * atanf((float)H / (2.0f * focalY));
    *fovX = fx; *fovY = fy;
    double tanHalfY = tan((double)fy / 2.0), tanHalfX = tan((double)fx / 2.0);
    double top = tanHalfY * znear, bottom = -top, right = tanHalfX * znear, left = -right;
    double P[16] = {0};
    P[0] = 2 * znear / (right - left);
    P[5] = 2 * znear / (top - bottom);
    P[8] = (right + left) / (right - left);
    P[9] = (top + bottom) / (top - bottom);
    P[10] = zfar / (zfar - znear);
    P[11] = 1.0;
    P[14] = -znear * zfar / (zfar - znear);
    for (int i = 0; i < 16; i++) proj[i] = (float)P[i];
    camCenter[0] = (float)c2w[3]; camCenter[1] = (float)c2w[7]; camCenter[2] = (float)c2w[11];
    return 0;
}

/* ======================================================================== *
 * a2  Activations and their VJPs  (Trainer/GaussianRenderer.swift:936-963)
 * ======================================================================== */
GSO_API void GSO_NAME(activations_forward)(int N, const real* opacity_raw, const real* scales_raw,
                                           const real* rot_raw, real* opacity, real* scales, real* rot)
{
    for (int i = 0; i < N; i++) {
        opacity[i] = R(1.0) / (R(1.0) + r_exp(-opacity_raw[i]));            /* :961-963 */
        for (int k = 0; k < 3; k++) scales[i * 3 + k] = r_exp(scales_raw[i * 3 + k]); /* :936-938 */
        const real* q = rot_raw + i * 4;
        real n = r_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int k = 0; k < 4; k++) rot[i * 4 + k] = q[k] / (n + R(1e-8)); /* :939-947, eps outside sqrt */
    }
}

GSO_API void GSO_NAME(activations_backward)(int N, const real* opacity_raw, const real* scales_raw,
                                            const real* rot_raw, const real* gOpacity, const real* gScales,
                                            const real* gRot, real* dOpacityRaw, real* dScalesRaw, real* dRotRaw)
{
    for (int i = 0; i < N; i++) {
        real s = R(1.0) / (R(1.0) + r_exp(-opacity_raw[i]));
        dOpacityRaw[i] = gOpacity[i] * s * (R(1.0) - s);
        for (int k = 0; k < 3; k++) dScalesRaw[i * 3 + k] = gScales[i * 3 + k] * r_exp(scales_raw[i * 3 + k]);
        const real* q = rot_raw + i * 4;
        const real* g = gRot + i * 4;
        real n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
        real n = r_sqrt(n2);
        real den = n + R(1e-8);
        real dot = g[0] * q[0] + g[1] * q[1] + g[2] * q[2] + g[3] * q[3];
        real dn = -dot / (den * den);           /* d/d(den) of q/den, summed */
        real dn2 = dn * R(0.5) / n;             /* sqrt VJP as MLX: g / (2 sqrt(x)) */
        for (int k = 0; k < 4; k++) dRotRaw[i * 4 + k] = g[k] / den + R(2.0) * q[k] * dn2;
    }
}

/* ======================================================================== *
 * a3/a4  Projection  (slang/gaussian_projection_screen_shared.slang,
 *                     slang/gaussian_projection_kernels.slang)
 * ======================================================================== */
#define SH_C0 R(0.28209479177387814)
#define SH_C1 R(0.4886025119029199)
#define SH_C2A R(1.0925484305920792)
#define SH_C2C R(0.31539156525252005)
#define SH_C2E R(0.5462742152960396)
#define SH_C3A R(0.5900435899266435)
#define SH_C3B R(2.890611442640554)
#define SH_C3C R(0.4570457994644658)
#define SH_C3D R(0.3731763325901154)
#define SH_C3E R(1.445305721320277)
#define SH_C4A R(2.5033429417967046)
#define SH_C4B R(1.7701307697799304)
#define SH_C4C R(0.9461746957575601)
#define SH_C4D R(0.6690465435572892)
#define SH_C4E R(0.10578554691520431)
#define SH_C4F R(0.47308734787878004)
#define SH_C4G R(0.6258357354491761)

/* Basis values b[k] in the reference's expression order
 * (gaussian_projection_screen_shared.slang:269-311). */
static void sh_basis(int degree, real x, real y, real z, real b[25])
{
    for (int k = 0; k < 25; k++) b[k] = R(0.0);
    b[0] = SH_C0;
    if (degree > 0) {
        b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
        if (degree > 1) {
            real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2A * xy;
            b[5] = -SH_C2A * yz;
            b[6] = SH_C2C * (R(2.0) * zz - xx - yy);
            b[7] = -SH_C2A * xz;
            b[8] = SH_C2E * (xx - yy);
            if (degree > 2) {
                b[9] = -SH_C3A * y * (R(3.0) * xx - yy);
                b[10] = SH_C3B * xy * z;
                b[11] = -SH_C3C * y * (R(4.0) * zz - xx - yy);
                b[12] = SH_C3D * z * (R(2.0) * zz - R(3.0) * xx - R(3.0) * yy);
                b[13] = -SH_C3C * x * (R(4.0) * zz - xx - yy);
                b[14] = SH_C3E * z * (xx - yy);
                b[15] = -SH_C3A * x * (xx - R(3.0) * yy);
                if (degree > 3) {
                    b[16] = SH_C4A * xy * (xx - yy);
                    b[17] = -SH_C4B * yz * (R(3.0) * xx - yy);
                    b[18] = SH_C4C * xy * (R(7.0) * zz - R(1.0));
                    b[19] = -SH_C4D * yz * (R(7.0) * zz - R(3.0));
                    b[20] = SH_C4E * (zz * (R(35.0) * zz - R(30.0)) + R(3.0));
                    b[21] = -SH_C4D * xz * (R(7.0) * zz - R(3.0));
                    b[22] = SH_C4F * (xx - yy) * (R(7.0) * zz - R(1.0));
                    b[23] = -SH_C4B * xz * (xx - R(3.0) * yy);
                    b[24] = SH_C4G * (xx * (xx - R(3.0) * yy) - yy * (R(3.0) * xx - yy));
                }
            }
        }
    }
}

/* Gradient of each basis function w.r.t. (x,y,z). */
static void sh_basis_grad(int degree, real x, real y, real z, real gx[25], real gy[25], real gz[25])
{
    for (int k = 0; k < 25; k++) gx[k] = gy[k] = gz[k] = R(0.0);
    if (degree > 0) {
        gy[1] = -SH_C1; gz[2] = SH_C1; gx[3] = -SH_C1;
        if (degree > 1) {
            real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            gx[4] = SH_C2A * y; gy[4] = SH_C2A * x;
            gy[5] = -SH_C2A * z; gz[5] = -SH_C2A * y;
            gx[6] = SH_C2C * (-R(2.0) * x); gy[6] = SH_C2C * (-R(2.0) * y); gz[6] = SH_C2C * (R(4.0) * z);
            gx[7] = -SH_C2A * z; gz[7] = -SH_C2A * x;
            gx[8] = SH_C2E * R(2.0) * x; gy[8] = -SH_C2E * R(2.0) * y;
            if (degree > 2) {
                gx[9] = -SH_C3A * R(6.0) * xy; gy[9] = -SH_C3A * (R(3.0) * xx - R(3.0) * yy);
                gx[10] = SH_C3B * yz; gy[10] = SH_C3B * xz; gz[10] = SH_C3B * xy;
                gx[11] = -SH_C3C * (-R(2.0) * xy); gy[11] = -SH_C3C * (R(4.0) * zz - xx - R(3.0) * yy);
                gz[11] = -SH_C3C * (R(8.0) * yz);
                gx[12] = SH_C3D * (-R(6.0) * xz); gy[12] = SH_C3D * (-R(6.0) * yz);
                gz[12] = SH_C3D * (R(6.0) * zz - R(3.0) * xx - R(3.0) * yy);
                gx[13] = -SH_C3C * (R(4.0) * zz - R(3.0) * xx - yy); gy[13] = -SH_C3C * (-R(2.0) * xy);
                gz[13] = -SH_C3C * (R(8.0) * xz);
                gx[14] = SH_C3E * R(2.0) * xz; gy[14] = -SH_C3E * R(2.0) * yz; gz[14] = SH_C3E * (xx - yy);
                gx[15] = -SH_C3A * (R(3.0) * xx - R(3.0) * yy); gy[15] = -SH_C3A * (-R(6.0) * xy);
                if (degree > 3) {
                    real s7 = R(7.0) * zz;
                    gx[16] = SH_C4A * (R(3.0) * xx * y - yy * y); gy[16] = SH_C4A * (xx * x - R(3.0) * x * yy);
                    gx[17] = -SH_C4B * (R(6.0) * xy * z); gy[17] = -SH_C4B * (R(3.0) * xx * z - R(3.0) * yy * z);
                    gz[17] = -SH_C4B * (R(3.0) * xx * y - yy * y);
                    gx[18] = SH_C4C * y * (s7 - R(1.0)); gy[18] = SH_C4C * x * (s7 - R(1.0));
                    gz[18] = SH_C4C * R(14.0) * xy * z;
                    gy[19] = -SH_C4D * z * (s7 - R(3.0)); gz[19] = -SH_C4D * y * (R(21.0) * zz - R(3.0));
                    gz[20] = SH_C4E * (R(140.0) * zz * z - R(60.0) * z);
                    gx[21] = -SH_C4D * z * (s7 - R(3.0)); gz[21] = -SH_C4D * x * (R(21.0) * zz - R(3.0));
                    gx[22] = SH_C4F * R(2.0) * x * (s7 - R(1.0)); gy[22] = -SH_C4F * R(2.0) * y * (s7 - R(1.0));
                    gz[22] = SH_C4F * R(14.0) * z * (xx - yy);
                    gx[23] = -SH_C4B * (R(3.0) * xx * z - R(3.0) * yy * z); gy[23] = -SH_C4B * (-R(6.0) * xy * z);
                    gz[23] = -SH_C4B * (xx * x - R(3.0) * x * yy);
                    gx[24] = SH_C4G * (R(4.0) * xx * x - R(12.0) * x * yy);
                    gy[24] = SH_C4G * (-R(12.0) * xx * y + R(4.0) * yy * y);
                }
            }
        }
    }
}

typedef struct {
    real c[9];
} cov3_t;

typedef struct {
    real qw, qx, qy, qz, safeNorm, norm, n2;
    real r[9];
} rot_ctx_t;

/* buildCov3dFromScaleRotation (shared.slang:117-168) */
static void build_cov3d(const real s[3], const real rq[4], cov3_t* out, rot_ctx_t* ctx)
{
    real n2 = rq[0] * rq[0] + rq[1] * rq[1] + rq[2] * rq[2] + rq[3] * rq[3];
    real norm = r_sqrt(n2);
    real safeNorm = norm > R(1e-8) ? norm : R(1e-8);
    real qw = rq[0] / safeNorm, qx = rq[1] / safeNorm, qy = rq[2] / safeNorm, qz = rq[3] / safeNorm;
    real r00 = R(1.0) - R(2.0) * (qy * qy + qz * qz);
    real r01 = R(2.0) * (qx * qy - qw * qz);
    real r02 = R(2.0) * (qx * qz + qw * qy);
    real r10 = R(2.0) * (qx * qy + qw * qz);
    real r11 = R(1.0) - R(2.0) * (qx * qx + qz * qz);
    real r12 = R(2.0) * (qy * qz - qw * qx);
    real r20 = R(2.0) * (qx * qz - qw * qy);
    real r21 = R(2.0) * (qy * qz + qw * qx);
    real r22 = R(1.0) - R(2.0) * (qx * qx + qy * qy);
    real l00 = r00 * s[0], l01 = r01 * s[1], l02 = r02 * s[2];
    real l10 = r10 * s[0], l11 = r11 * s[1], l12 = r12 * s[2];
    real l20 = r20 * s[0], l21 = r21 * s[1], l22 = r22 * s[2];
    out->c[0] = l00 * l00 + l01 * l01 + l02 * l02;
    out->c[1] = l00 * l10 + l01 * l11 + l02 * l12;
    out->c[2] = l00 * l20 + l01 * l21 + l02 * l22;
    out->c[3] = l10 * l00 + l11 * l01 + l12 * l02;
    out->c[4] = l10 * l10 + l11 * l11 + l12 * l12;
    out->c[5] = l10 * l20 + l11 * l21 + l12 * l22;
    out->c[6] = l20 * l00 + l21 * l01 + l22 * l02;
    out->c[7] = l20 * l10 + l21 * l11 + l22 * l12;
    out->c[8] = l20 * l20 + l21 * l21 + l22 * l22;
    if (ctx) {
        ctx->qw = qw; ctx->qx = qx; ctx->qy = qy; ctx->qz = qz;
        ctx->safeNorm = safeNorm; ctx->norm = norm; ctx->n2 = n2;
        ctx->r[0] = r00; ctx->r[1] = r01; ctx->r[2] = r02;
        ctx->r[3] = r10; ctx->r[4] = r11; ctx->r[5] = r12;
        ctx->r[6] = r20; ctx->r[7] = r21; ctx->r[8] = r22;
    }
}

typedef struct {
    real t0, t1, t2, limX, limY, clipX, clipY, tx, ty;
    real j00, j02, j11, j12;
    real b[6];  /* b00 b01 b02 b10 b11 b12 */
    real t[6];  /* t00 t01 t02 t10 t11 t12 */
} cov2_ctx_t;

/* buildCov2dFromCov3d (shared.slang:170-243) */
static void build_cov2d(const real m[3], const cov3_t* S, const real* V, real fovX, real fovY,
                        real focalX, real focalY, real out[4], cov2_ctx_t* ctx)
{
    real a00 = V[0], a01 = V[1], a02 = V[2], a10 = V[4], a11 = V[5], a12 = V[6];
    real a20 = V[8], a21 = V[9], a22 = V[10], t30 = V[12], t31 = V[13], t32 = V[14];
    real t0 = m[0] * a00 + m[1] * a10 + m[2] * a20 + t30;
    real t1 = m[0] * a01 + m[1] * a11 + m[2] * a21 + t31;
    real t2 = m[0] * a02 + m[1] * a12 + m[2] * a22 + t32;
    real tanFovX = r_tan(fovX * R(0.5)), tanFovY = r_tan(fovY * R(0.5));
    real limX = tanFovX * R(1.3), limY = tanFovY * R(1.3);
    real clipX = t2 < -limX ? -limX : (t2 > limX ? limX : t2);   /* clamps z, not x/z (:202) */
    real clipY = t2 < -limY ? -limY : (t2 > limY ? limY : t2);
    real tx = t0 / clipX * t2;
    real ty = t1 / clipY * t2;
    real tz = t2;
    real j00 = focalX / tz;
    real j02 = -tx * focalX / (tz * tz);
    real j11 = focalY / tz;
    real j12 = -ty * focalY / (tz * tz);
    real w00 = a00, w01 = a10, w02 = a20, w10 = a01, w11 = a11, w12 = a21, w20 = a02, w21 = a12, w22 = a22;
    real b00 = j00 * w00 + j02 * w20, b01 = j00 * w01 + j02 * w21, b02 = j00 * w02 + j02 * w22;
    real b10 = j11 * w10 + j12 * w20, b11 = j11 * w11 + j12 * w21, b12 = j11 * w12 + j12 * w22;
    const real* c = S->c;
    real t00 = b00 * c[0] + b01 * c[3] + b02 * c[6];
    real t01 = b00 * c[1] + b01 * c[4] + b02 * c[7];
    real t02 = b00 * c[2] + b01 * c[5] + b02 * c[8];
    real t10 = b10 * c[0] + b11 * c[3] + b12 * c[6];
    real t11 = b10 * c[1] + b11 * c[4] + b12 * c[7];
    real t12 = b10 * c[2] + b11 * c[5] + b12 * c[8];
    out[0] = t00 * b00 + t01 * b01 + t02 * b02 + R(0.3);
    out[1] = t00 * b10 + t01 * b11 + t02 * b12;
    out[2] = t10 * b00 + t11 * b01 + t12 * b02;
    out[3] = t10 * b10 + t11 * b11 + t12 * b12 + R(0.3);
    if (ctx) {
        ctx->t0 = t0; ctx->t1 = t1; ctx->t2 = t2; ctx->limX = limX; ctx->limY = limY;
        ctx->clipX = clipX; ctx->clipY = clipY; ctx->tx = tx; ctx->ty = ty;
        ctx->j00 = j00; ctx->j02 = j02; ctx->j11 = j11; ctx->j12 = j12;
        ctx->b[0] = b00; ctx->b[1] = b01; ctx->b[2] = b02; ctx->b[3] = b10; ctx->b[4] = b11; ctx->b[5] = b12;
        ctx->t[0] = t00; ctx->t[1] = t01; ctx->t[2] = t02; ctx->t[3] = t10; ctx->t[4] = t11; ctx->t[5] = t12;
    }
}

/* gaussian_projection_screen_fused_forward (kernels.slang:36-173) */
GSO_API void GSO_NAME(projection_forward)(int N, int K, int degree, const real* scales, const real* rot,
                                          const real* means3d, const real* shs, const real* camCenter,
                                          const real* V, const real* P, real fovX, real fovY, real focalX,
                                          real focalY, real imageW, real imageH, real* means2d, real* depths,
                                          real* color, real* cov2d, real* conic, real* radii, real* rectMin,
                                          real* rectMax)
{
    int coeffCount = (degree + 1) * (degree + 1);
    if (coeffCount > 25) coeffCount = 25;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < N; p++) {
        const real* m = means3d + 3 * p;
        /* evaluateProjectionNdcOutputs (shared.slang:53-107) */
        real pv0 = m[0] * V[0] + m[1] * V[4] + m[2] * V[8] + V[12];
        real pv1 = m[0] * V[1] + m[1] * V[5] + m[2] * V[9] + V[13];
        real pv2 = m[0] * V[2] + m[1] * V[6] + m[2] * V[10] + V[14];
        real pv3 = m[0] * V[3] + m[1] * V[7] + m[2] * V[11] + V[15];
        real pc0 = pv0 * P[0] + pv1 * P[4] + pv2 * P[8] + pv3 * P[12];
        real pc1 = pv0 * P[1] + pv1 * P[5] + pv2 * P[9] + pv3 * P[13];
        real pc3 = pv0 * P[3] + pv1 * P[7] + pv2 * P[11] + pv3 * P[15];
        real wInv = R(1.0) / (pc3 + R(0.000001));
        real ndcX = pc0 * wInv, ndcY = pc1 * wInv;
        real visibleMask = (pv2 >= R(0.2)) ? R(1.0) : R(0.0);      /* kernels.slang:63 */

        /* ndcToScreen (shared.slang:109-115) */
        real sx = ((ndcX + R(1.0)) * imageW - R(1.0)) * R(0.5);
        real sy = ((ndcY + R(1.0)) * imageH - R(1.0)) * R(0.5);

        /* evaluateShColorFromPoint (shared.slang:257-319): direction NOT normalised */
        real x = m[0] - camCenter[0], y = m[1] - camCenter[1], z = m[2] - camCenter[2];
        real b[25];
        sh_basis(degree, x, y, z, b);
        const real* sh = shs + (size_t)p * K * 3;
        real col[3];
        for (int ch = 0; ch < 3; ch++) {
            real acc = b[0] * sh[ch];
            for (int k = 1; k < coeffCount; k++) acc += b[k] * sh[k * 3 + ch];
            acc += R(0.5);
            col[ch] = acc > R(0.0) ? acc : R(0.0);
        }

        cov3_t S;
        build_cov3d(scales + 3 * p, rot + 4 * p, &S, NULL);
        real c2[4];
        build_cov2d(m, &S, V, fovX, fovY, focalX, focalY, c2, NULL);
        /* inverseCov2d (shared.slang:245-255) */
        real det = c2[0] * c2[3] - c2[1] * c2[2];
        real con[4] = {c2[3] / det, -c2[1] / det, -c2[2] / det, c2[0] / det};
        /* computeRadiusFromCov2d (shared.slang:375-382) */
        real mid = R(0.5) * (c2[0] + c2[3]);
        real delta = mid * mid - det;
        if (!(delta > R(1e-5))) delta = R(1e-5);
        real lambdaMax = mid + r_sqrt(delta);
        real radius = R(3.0) * r_ceil(r_sqrt(lambdaMax));
        real visibleRadius = radius * visibleMask;

        means2d[2 * p] = sx; means2d[2 * p + 1] = sy;
        depths[p] = pv2;
        color[3 * p] = col[0]; color[3 * p + 1] = col[1]; color[3 * p + 2] = col[2];
        for (int k = 0; k < 4; k++) { cov2d[4 * p + k] = c2[k]; conic[4 * p + k] = con[k]; }
        radii[p] = visibleRadius;
        /* rect (kernels.slang:158-172): one-sided clamps */
        real maxX = imageW - R(1.0), maxY = imageH - R(1.0);
        real minX = sx - visibleRadius, minY = sy - visibleRadius;
        real maxRX = sx + visibleRadius, maxRY = sy + visibleRadius;
        if (minX < R(0.0)) minX = R(0.0);
        if (minY < R(0.0)) minY = R(0.0);
        if (maxRX > maxX) maxRX = maxX;
        if (maxRY > maxY) maxRY = maxY;
        rectMin[2 * p] = minX; rectMin[2 * p + 1] = minY;
        rectMax[2 * p] = maxRX; rectMax[2 * p + 1] = maxRY;
    }
}

/* gaussian_projection_screen_fused_backward (kernels.slang:205-398): reverse
 * mode of the differentiable functions above, hand-derived; tie conventions
 * from the expanded Metal (see top of file). gradShs must be zero-initialised
 * by the caller beyond coeffCount (the reference passes initValue 0,
 * GaussianRenderer.swift:674). */
GSO_API void GSO_NAME(projection_backward)(int N, int K, int degree, const real* scales, const real* rot,
                                           const real* means3d, const real* shs, const real* camCenter,
                                           const real* V, const real* P, real fovX, real fovY, real focalX,
                                           real focalY, real imageW, real imageH, const real* cotDepths,
                                           const real* cotMeans2d, const real* cotCov2d, const real* cotColor,
                                           const real* cotConic, real* gradScales, real* gradRot,
                                           real* gradMeans3d, real* gradShs, real* gradCamCenterPoint)
{
    int coeffCount = (degree + 1) * (degree + 1);
    if (coeffCount > 25) coeffCount = 25;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < N; p++) {
        const real* m = means3d + 3 * p;
        const real* s = scales + 3 * p;
        cov3_t S;
        rot_ctx_t rc;
        build_cov3d(s, rot + 4 * p, &S, &rc);
        real c2[4];
        cov2_ctx_t cc;
        build_cov2d(m, &S, V, fovX, fovY, focalX, focalY, c2, &cc);

        /* ---- inverseCov2d backward (s_bwd_prop_inverseCov2d_0) ---- */
        const real* gq = cotConic + 4 * p;
        real det = c2[0] * c2[3] - c2[1] * c2[2];
        real det2 = det * det;
        real S29 = gq[3] / det2, S30 = gq[2] / det2, S31 = gq[1] / det2, S32 = gq[0] / det2;
        real S33 = c2[0] * -S29 + -c2[2] * -S30 + -c2[1] * -S31 + c2[3] * -S32;
        real S34 = -S33;
        real dC[4];
        dC[2] = -(det * S30) + c2[1] * S34;   /* c10 */
        dC[1] = -(det * S31) + c2[2] * S34;   /* c01 */
        dC[3] = det * S32 + c2[0] * S33;      /* c11 */
        dC[0] = det * S29 + c2[3] * S33;      /* c00 */
        for (int k = 0; k < 4; k++) dC[k] += cotCov2d[4 * p + k];

        /* ---- buildCov2dFromCov3d backward ---- */
        const real* b = cc.b;   /* b0k = b[k], b1k = b[3+k] */
        const real* t = cc.t;
        real dt[6], db[6];
        for (int k = 0; k < 3; k++) {
            dt[k] = dC[0] * b[k] + dC[1] * b[3 + k];
            dt[3 + k] = dC[2] * b[k] + dC[3] * b[3 + k];
            db[k] = dC[0] * t[k] + dC[2] * t[3 + k];
            db[3 + k] = dC[1] * t[k] + dC[3] * t[3 + k];
        }
        real dS[9];
        for (int l = 0; l < 3; l++) {
            for (int k = 0; k < 3; k++) dS[l * 3 + k] = b[l] * dt[k] + b[3 + l] * dt[3 + k];
            db[l] += dt[0] * S.c[l * 3 + 0] + dt[1] * S.c[l * 3 + 1] + dt[2] * S.c[l * 3 + 2];
            db[3 + l] += dt[3] * S.c[l * 3 + 0] + dt[4] * S.c[l * 3 + 1] + dt[5] * S.c[l * 3 + 2];
        }
        /* W[r][c] = V[c][r] (3x3 part): w0k = V[k*4+0], w1k = V[k*4+1], w2k = V[k*4+2] */
        real dj00 = 0, dj02 = 0, dj11 = 0, dj12 = 0;
        for (int k = 0; k < 3; k++) {
            dj00 += db[k] * V[k * 4 + 0];
            dj02 += db[k] * V[k * 4 + 2];
            dj11 += db[3 + k] * V[k * 4 + 1];
            dj12 += db[3 + k] * V[k * 4 + 2];
        }
        real tz = cc.t2, tz2 = tz * tz;
        real dtz = -focalX / tz2 * dj00 - focalY / tz2 * dj11;
        real dtx = -focalX / tz2 * dj02;
        real dty = -focalY / tz2 * dj12;
        real dtz2 = cc.tx * focalX / (tz2 * tz2) * dj02 + cc.ty * focalY / (tz2 * tz2) * dj12;
        dtz += R(2.0) * tz * dtz2;
        /* tx = (t0/clipX)*t2 */
        real ux = cc.t0 / cc.clipX, uy = cc.t1 / cc.clipY;
        real dt2 = dtz + ux * dtx + uy * dty;
        real dux = cc.t2 * dtx, duy = cc.t2 * dty;
        real dt0 = dux / cc.clipX, dt1 = duy / cc.clipY;
        real dclipX = -cc.t0 / (cc.clipX * cc.clipX) * dux;
        real dclipY = -cc.t1 / (cc.clipY * cc.clipY) * duy;
        if (cc.t2 >= -cc.limX && cc.t2 <= cc.limX) dt2 += dclipX;   /* _d_clamp_0: inclusive */
        if (cc.t2 >= -cc.limY && cc.t2 <= cc.limY) dt2 += dclipY;
        real dm[3];
        for (int a = 0; a < 3; a++) dm[a] = V[a * 4 + 0] * dt0 + V[a * 4 + 1] * dt1 + V[a * 4 + 2] * dt2;

        /* ---- colour backward (evaluateShColorFromPoint) ---- */
        real x = m[0] - camCenter[0], y = m[1] - camCenter[1], z = m[2] - camCenter[2];
        real bas[25], gx[25], gy[25], gz[25];
        sh_basis(degree, x, y, z, bas);
        sh_basis_grad(degree, x, y, z, gx, gy, gz);
        const real* sh = shs + (size_t)p * K * 3;
        real* gsh = gradShs + (size_t)p * K * 3;
        real dx = 0, dy = 0, dz = 0;
        for (int ch = 0; ch < 3; ch++) {
            real acc = bas[0] * sh[ch];
            for (int k = 1; k < coeffCount; k++) acc += bas[k] * sh[k * 3 + ch];
            acc += R(0.5);
            real mg = d_max_left(acc, R(0.0), cotColor[3 * p + ch]);
            for (int k = 0; k < coeffCount; k++) {
                gsh[k * 3 + ch] = bas[k] * mg;
                real w = sh[k * 3 + ch] * mg;
                dx += gx[k] * w; dy += gy[k] * w; dz += gz[k] * w;
            }
        }
        dm[0] += dx; dm[1] += dy; dm[2] += dz;
        gradCamCenterPoint[3 * p] = -dx; gradCamCenterPoint[3 * p + 1] = -dy; gradCamCenterPoint[3 * p + 2] = -dz;

        /* ---- means2d -> ndc -> clip -> view -> point (evaluateProjectionNdcOutputs bwd) ---- */
        real pv0 = m[0] * V[0] + m[1] * V[4] + m[2] * V[8] + V[12];
        real pv1 = m[0] * V[1] + m[1] * V[5] + m[2] * V[9] + V[13];
        real pv2 = m[0] * V[2] + m[1] * V[6] + m[2] * V[10] + V[14];
        real pv3 = m[0] * V[3] + m[1] * V[7] + m[2] * V[11] + V[15];
        real pc0 = pv0 * P[0] + pv1 * P[4] + pv2 * P[8] + pv3 * P[12];
        real pc1 = pv0 * P[1] + pv1 * P[5] + pv2 * P[9] + pv3 * P[13];
        real pc3 = pv0 * P[3] + pv1 * P[7] + pv2 * P[11] + pv3 * P[15];
        real wInv = R(1.0) / (pc3 + R(0.000001));
        real dndcX = cotMeans2d[2 * p] * R(0.5) * imageW;
        real dndcY = cotMeans2d[2 * p + 1] * R(0.5) * imageH;
        real dpc0 = dndcX * wInv, dpc1 = dndcY * wInv;
        real dwInv = pc0 * dndcX + pc1 * dndcY;
        real dpc3 = -dwInv * wInv * wInv;
        real dpv[4];
        for (int i = 0; i < 4; i++) dpv[i] = P[i * 4 + 0] * dpc0 + P[i * 4 + 1] * dpc1 + P[i * 4 + 3] * dpc3;
        dpv[2] += cotDepths[p];
        for (int a = 0; a < 3; a++)
            dm[a] += V[a * 4 + 0] * dpv[0] + V[a * 4 + 1] * dpv[1] + V[a * 4 + 2] * dpv[2] + V[a * 4 + 3] * dpv[3];
        gradMeans3d[3 * p] = dm[0]; gradMeans3d[3 * p + 1] = dm[1]; gradMeans3d[3 * p + 2] = dm[2];

        /* ---- buildCov3dFromScaleRotation backward ---- */
        real L[9], dL[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) L[i * 3 + j] = rc.r[i * 3 + j] * s[j];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            real acc = 0;
            for (int k = 0; k < 3; k++) acc += (dS[i * 3 + k] + dS[k * 3 + i]) * L[k * 3 + j];
            dL[i * 3 + j] = acc;
        }
        real ds[3] = {0, 0, 0}, dr[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            ds[j] += dL[i * 3 + j] * rc.r[i * 3 + j];
            dr[i * 3 + j] = dL[i * 3 + j] * s[j];
        }
        gradScales[3 * p] = ds[0]; gradScales[3 * p + 1] = ds[1]; gradScales[3 * p + 2] = ds[2];
        real qw = rc.qw, qx = rc.qx, qy = rc.qy, qz = rc.qz;
        real dqw = R(2.0) * (-qz * dr[1] + qy * dr[2] + qz * dr[3] - qx * dr[5] - qy * dr[6] + qx * dr[7]);
        real dqx = R(2.0) * (qy * dr[1] + qz * dr[2] + qy * dr[3] - qw * dr[5] + qz * dr[6] + qw * dr[7])
                   - R(4.0) * qx * (dr[4] + dr[8]);
        real dqy = R(2.0) * (qx * dr[1] + qw * dr[2] + qx * dr[3] + qz * dr[5] - qw * dr[6] + qz * dr[7])
                   - R(4.0) * qy * (dr[0] + dr[8]);
        real dqz = R(2.0) * (-qw * dr[1] + qx * dr[2] + qw * dr[3] + qy * dr[5] + qx * dr[6] + qy * dr[7])
                   - R(4.0) * qz * (dr[0] + dr[4]);
        const real* rq = rot + 4 * p;
        real sn = rc.safeNorm;
        real dsafe = -(dqw * rq[0] + dqx * rq[1] + dqy * rq[2] + dqz * rq[3]) / (sn * sn);
        real dnorm = d_max_left(rc.norm, R(1e-8), dsafe);
        real dn2 = d_sqrt(rc.n2, dnorm);
        gradRot[4 * p + 0] = dqw / sn + R(2.0) * rq[0] * dn2;
        gradRot[4 * p + 1] = dqx / sn + R(2.0) * rq[1] * dn2;
        gradRot[4 * p + 2] = dqy / sn + R(2.0) * rq[2] * dn2;
        gradRot[4 * p + 3] = dqz / sn + R(2.0) * rq[3] * dn2;
    }
}

/* Test hooks: expose the covariance and SH-basis building blocks so that the
 * reference tests' fixtures (GaussianSplattingMlxTests.swift:73-108 rotation
 * convention, ShUtilsTests.swift:15-151 SH polynomials) can pin them directly. */
GSO_API void GSO_NAME(cov3d)(const real* s, const real* q, real* out9, real* rot9)
{
    cov3_t S; rot_ctx_t rc;
    build_cov3d(s, q, &S, &rc);
    for (int i = 0; i < 9; i++) { out9[i] = S.c[i]; rot9[i] = rc.r[i]; }
}
GSO_API void GSO_NAME(sh_basis)(int degree, real x, real y, real z, real* b25)
{
    sh_basis(degree, x, y, z, b25);
}

/* a5 buildPackedGaussians (GaussianRenderer.swift:85-99, index map :45-51) */
GSO_API void GSO_NAME(pack_gaussians)(int N, const real* means2d, const real* conic, const real* color,
                                      const real* opacity, const real* depths, real* packed)
{
    for (int i = 0; i < N; i++) {
        real* o = packed + (size_t)i * 11;
        o[0] = means2d[2 * i]; o[1] = means2d[2 * i + 1];
        for (int k = 0; k < 4; k++) o[2 + k] = conic[4 * i + k];
        for (int k = 0; k < 3; k++) o[6 + k] = color[3 * i + k];
        o[9] = opacity[i];
        o[10] = depths[i];
    }
}

/* ======================================================================== *
 * a6  Tile binning  (slang/gaussian_tile_global_kernels.slang:17-404,
 *                    GaussianRenderer.swift:333-490)
 * ======================================================================== */
static inline void tile_rect(const real* rectMin, const real* rectMax, int idx, int tileW, int tileH, int imageW,
                             int imageH, int* tMinX, int* tMinY, int* tMaxX, int* tMaxY, int* gridWOut)
{
    real rMinX = rectMin[idx * 2], rMinY = rectMin[idx * 2 + 1];
    real rMaxX = rectMax[idx * 2], rMaxY = rectMax[idx * 2 + 1];
    int a = (int)r_floor(rMinX / (real)tileW), b = (int)r_floor(rMinY / (real)tileH);
    int c = (int)r_floor(rMaxX / (real)tileW) + 1, d = (int)r_floor(rMaxY / (real)tileH) + 1;
    int gridW = (imageW + tileW - 1) / tileW, gridH = (imageH + tileH - 1) / tileH;
#define CL(v, hi) ((v) < 0 ? 0 : ((v) > (hi) ? (hi) : (v)))
    *tMinX = CL(a, gridW); *tMinY = CL(b, gridH); *tMaxX = CL(c, gridW); *tMaxY = CL(d, gridH);
#undef CL
    *gridWOut = gridW;
}

/* count_tiles_per_gaussian (:17-58) */
GSO_API void GSO_NAME(count_tiles)(int N, int tileW, int tileH, int imageW, int imageH, const real* rectMin,
                                   const real* rectMax, const real* radii, uint32_t* tilesTouched)
{
    for (int i = 0; i < N; i++) {
        if (radii[i] <= R(0.0)) { tilesTouched[i] = 0; continue; }
        int x0, y0, x1, y1, gw;
        tile_rect(rectMin, rectMax, i, tileW, tileH, imageW, imageH, &x0, &y0, &x1, &y1, &gw);
        tilesTouched[i] = (uint32_t)((x1 - x0) * (y1 - y0));
    }
}

/* cumsum - tilesTouched (GaussianRenderer.swift:398-409); returns M */
GSO_API uint32_t gso_exclusive_scan(int N, const uint32_t* tilesTouched, uint32_t* offsets)
{
    uint32_t run = 0;
    for (int i = 0; i < N; i++) { offsets[i] = run; run += tilesTouched[i]; }
    return run;
}

/* generate_keys (:73-126). Depth bits are those of an f32 depth. */
GSO_API void GSO_NAME(generate_keys)(int N, int tileW, int tileH, int imageW, int imageH, const real* depths,
                                     const real* rectMin, const real* rectMax, const real* radii,
                                     const uint32_t* offsets, uint32_t* keysHigh, uint32_t* keysLow,
                                     uint32_t* gaussIdx)
{
    for (int i = 0; i < N; i++) {
        if (radii[i] <= R(0.0)) continue;
        float df = (float)depths[i];
        uint32_t bits;
        memcpy(&bits, &df, 4);
        int x0, y0, x1, y1, gw;
        tile_rect(rectMin, rectMax, i, tileW, tileH, imageW, imageH, &x0, &y0, &x1, &y1, &gw);
        uint32_t off = offsets[i];
        for (int ty = y0; ty < y1; ty++)
            for (int tx = x0; tx < x1; tx++) {
                keysHigh[off] = (uint32_t)(ty * gw + tx);
                keysLow[off] = bits;
                gaussIdx[off] = (uint32_t)i;
                off++;
            }
    }
}

/* radix_sort_tile_keys_fused_forward (:151-305): a stable LSD sort over the
 * low key then the high key == stable sort by (high, low). Restated as a
 * stable 8-bit LSD radix sort over all 64 key bits. */
GSO_API void gso_sort_pairs(uint32_t M, const uint32_t* keysHigh, const uint32_t* keysLow, const uint32_t* values,
                            uint32_t* sortedHigh, uint32_t* sortedLow, uint32_t* sortedValues)
{
    if (M == 0) return;
    uint64_t* k0 = (uint64_t*)malloc(sizeof(uint64_t) * M);
    uint64_t* k1 = (uint64_t*)malloc(sizeof(uint64_t) * M);
    uint32_t* v0 = (uint32_t*)malloc(sizeof(uint32_t) * M);
    uint32_t* v1 = (uint32_t*)malloc(sizeof(uint32_t) * M);
    for (uint32_t i = 0; i < M; i++) { k0[i] = ((uint64_t)keysHigh[i] << 32) | keysLow[i]; v0[i] = values[i]; }
    for (int pass = 0; pass < 8; pass++) {
        size_t hist[257] = {0};
        int sh = pass * 8;
        for (uint32_t i = 0; i < M; i++) hist[((k0[i] >> sh) & 255) + 1]++;
        for (int d = 0; d < 256; d++) hist[d + 1] += hist[d];
        for (uint32_t i = 0; i < M; i++) {
            size_t dst = hist[(k0[i] >> sh) & 255]++;
            k1[dst] = k0[i]; v1[dst] = v0[i];
        }
        uint64_t* tk = k0; k0 = k1; k1 = tk;
        uint32_t* tv = v0; v0 = v1; v1 = tv;
    }
    for (uint32_t i = 0; i < M; i++) {
        sortedHigh[i] = (uint32_t)(k0[i] >> 32); sortedLow[i] = (uint32_t)k0[i]; sortedValues[i] = v0[i];
    }
    free(k0); free(k1); free(v0); free(v1);
}

/* compute_tile_ranges (:314-344); tileRanges zero-initialised by caller
 * semantics (GaussianRenderer.swift:448) -- done here. */
GSO_API void gso_tile_ranges(uint32_t M, uint32_t numTiles, const uint32_t* sortedHigh, uint32_t* tileRanges)
{
    memset(tileRanges, 0, sizeof(uint32_t) * 2 * numTiles);
    for (uint32_t i = 0; i < M; i++) {
        uint32_t cur = sortedHigh[i];
        if (i == 0) tileRanges[cur * 2] = 0;
        else {
            uint32_t prev = sortedHigh[i - 1];
            if (cur != prev) { tileRanges[prev * 2 + 1] = i; tileRanges[cur * 2] = i; }
        }
        if (i == M - 1) tileRanges[cur * 2 + 1] = M;
    }
}

/* compute_tile_counts_from_ranges (:353-367); returns max count (B) */
GSO_API uint32_t gso_tile_counts(uint32_t numTiles, const uint32_t* tileRanges, uint32_t* tileCounts)
{
    uint32_t mx = 0;
    for (uint32_t t = 0; t < numTiles; t++) {
        uint32_t s = tileRanges[t * 2], e = tileRanges[t * 2 + 1];
        tileCounts[t] = e > s ? e - s : 0;
        if (tileCounts[t] > mx) mx = tileCounts[t];
    }
    return mx;
}

/* build_packed_tile_indices (:377-404) */
GSO_API void gso_build_packed_tile_indices(uint32_t numTiles, uint32_t maxTilePairs, const uint32_t* sortedGaussIdx,
                                           const uint32_t* tileRanges, int32_t* packedTileIndices)
{
    for (uint32_t t = 0; t < numTiles; t++) {
        uint32_t s = tileRanges[t * 2], e = tileRanges[t * 2 + 1];
        uint32_t count = e > s ? e - s : 0;
        for (uint32_t slot = 0; slot < maxTilePairs; slot++)
            packedTileIndices[(size_t)t * maxTilePairs + slot] = slot < count ? (int32_t)sortedGaussIdx[s + slot] : 0;
    }
}

/* ======================================================================== *
 * a7  Blend forward  (gaussian_tile_global_kernels.slang:437-614)
 * The reference reads the tile's list from the dense table
 * packedTileIndices[tile][0..count); here the same list is addressed as
 * sortedGaussIdx[tileRanges[tile][0] ...] (identical contents by :377-404).
 * ======================================================================== */
static inline real alpha_from_gaussian(const real* g, real px, real py, real* expOut)
{
    real dx = px - g[0], dy = py - g[1];
    real dxdy = dx * dy;
    real exponent = R(-0.5) * (dx * dx * g[2] + dy * dy * g[5] + dxdy * g[3] + dxdy * g[4]);
    real e = r_exp(exponent);
    if (expOut) *expOut = e;
    real raw = e * g[9];
    return raw > R(0.99) ? R(0.99) : raw;
}

GSO_API void GSO_NAME(blend_forward)(int imageW, int imageH, int tileW, int tileH, int whiteBg,
                                     const real* packed, const uint32_t* sortedGaussIdx,
                                     const uint32_t* tileRanges, real* outColor, real* outDepth, real* outAlpha,
                                     uint32_t* lastContrib)
{
    int gridW = (imageW + tileW - 1) / tileW;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < imageH; y++) {
        for (int x = 0; x < imageW; x++) {
            int p = y * imageW + x;
            int tile = (y / tileH) * gridW + (x / tileW);
            uint32_t s = tileRanges[tile * 2], e = tileRanges[tile * 2 + 1];
            uint32_t count = e > s ? e - s : 0;
            real px = (real)x, py = (real)y;          /* integer pixel coordinates (:555-556) */
            real cx = 0, cy = 0, cz = 0, dd = 0, T = R(1.0);
            uint32_t nContrib = count;
            for (uint32_t i = 0; i < count; i++) {
                const real* g = packed + (size_t)sortedGaussIdx[s + i] * 11;
                real a = alpha_from_gaussian(g, px, py, NULL);
                real contrib = T * a;
                cx = cx + contrib * g[6]; cy = cy + contrib * g[7]; cz = cz + contrib * g[8];
                dd = dd + contrib * g[10];
                T = T * (R(1.0) - a);
                if (T < R(1e-4)) { nContrib = i + 1; break; }   /* after the update (:598-603) */
            }
            real bg = whiteBg ? T : R(0.0);
            outColor[3 * p] = cx + bg; outColor[3 * p + 1] = cy + bg; outColor[3 * p + 2] = cz + bg;
            outDepth[p] = dd;
            outAlpha[p] = R(1.0) - T;
            lastContrib[p] = nContrib;
        }
    }
}

/* ======================================================================== *
 * a8  Blend backward  (gaussian_tile_global_kernels.slang:501-521, 648-881;
 * derivative bodies from gaussian_tile_global_backward_mlx.json `header`:
 * s_bwd_prop_updateTileGlobalPixelState_0, s_bwd_prop_tileGlobalAlphaFromGaussian_0)
 * Per-pixel arithmetic in `real`; the cross-pixel sum (simd_sum + float
 * atomics in the reference, order-nondeterministic) is accumulated in double
 * in a fixed order.
 * ======================================================================== */
GSO_API void GSO_NAME(blend_backward)(int N, int imageW, int imageH, int tileW, int tileH, int whiteBg,
                                      const real* packed, const uint32_t* sortedGaussIdx,
                                      const uint32_t* tileRanges, const real* cotColor, const real* cotDepth,
                                      const real* cotAlpha, const real* outColor, const real* outDepth,
                                      const real* outAlpha, const uint32_t* lastContrib, real* gradPacked)
{
    int gridW = (imageW + tileW - 1) / tileW, gridH = (imageH + tileH - 1) / tileH;
    int numTiles = gridW * gridH;
    (void)outColor; (void)outDepth;   /* colour/depth state is undone in the reference but never read back */
    uint32_t M = 0;
    for (int t = 0; t < numTiles; t++) if (tileRanges[2 * t + 1] > M) M = tileRanges[2 * t + 1];
    double* pair = (double*)calloc((size_t)M * 11 + 1, sizeof(double));
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < numTiles; tile++) {
        uint32_t s = tileRanges[tile * 2], e = tileRanges[tile * 2 + 1];
        uint32_t count = e > s ? e - s : 0;
        if (!count) continue;
        int tx0 = (tile % gridW) * tileW, ty0 = (tile / gridW) * tileH;
        for (int ly = 0; ly < tileH; ly++) {
            int y = ty0 + ly;
            if (y >= imageH) break;
            for (int lx = 0; lx < tileW; lx++) {
                int x = tx0 + lx;
                if (x >= imageW) break;
                int p = y * imageW + x;
                real cCx = cotColor[3 * p], cCy = cotColor[3 * p + 1], cCz = cotColor[3 * p + 2];
                real cD = cotDepth[p];
                real T = R(1.0) - outAlpha[p];
                real cT = -cotAlpha[p] + (whiteBg ? (cCx + cCy + cCz) : R(0.0));
                real px = (real)x, py = (real)y;
                uint32_t n = lastContrib[p];
                if (n > count) n = count;
                for (int ii = (int)n - 1; ii >= 0; ii--) {
                    uint32_t gi = sortedGaussIdx[s + ii];
                    const real* g = packed + (size_t)gi * 11;
                    real ex;
                    real a = alpha_from_gaussian(g, px, py, &ex);
                    /* undoTileGlobalPixelState (:501-521) */
                    real denom = R(1.0) - a;
                    if (denom < R(1e-6)) denom = R(1e-6);
                    real Tprev = T / denom;
                    real contrib = Tprev * a;
                    /* s_bwd_prop_updateTileGlobalPixelState_0 */
                    real S13 = g[10] * cD + g[8] * cCz + g[7] * cCy + g[6] * cCx;
                    real dAlpha = -(Tprev * cT) + Tprev * S13;
                    real cTnew = (R(1.0) - a) * cT + a * S13;
                    /* s_bwd_prop_tileGlobalAlphaFromGaussian_0 */
                    real S32 = (ex * g[9]) > R(0.99) ? R(0.0) : dAlpha;
                    real dOp = ex * S32;
                    real dE = g[9] * S32 * ex;
                    real S36 = R(-0.5) * dE;
                    real dx = px - g[0], dy = py - g[1], dxdy = dx * dy;
                    real S37 = dxdy * S36;
                    real S38 = dy * dy * S36;
                    real S39 = dy * (g[5] * S36);
                    real S40 = dx * dx * S36;
                    real S41 = dx * (g[2] * S36);
                    real S42 = g[4] * S36 + g[3] * S36;
                    real dMy = -(S39 + S39 + dx * S42);
                    real dMx = -(S41 + S41 + dy * S42);
                    double* o = pair + (size_t)(s + ii) * 11;
                    o[0] += dMx; o[1] += dMy; o[2] += S40; o[3] += S37; o[4] += S37; o[5] += S38;
                    o[6] += contrib * cCx; o[7] += contrib * cCy; o[8] += contrib * cCz;
                    o[9] += dOp; o[10] += contrib * cD;
                    T = Tprev;
                    cT = cTnew;
                }
            }
        }
    }
    double* acc = (double*)calloc((size_t)N * 11 + 1, sizeof(double));
    for (int tile = 0; tile < numTiles; tile++) {
        uint32_t s = tileRanges[tile * 2], e = tileRanges[tile * 2 + 1];
        for (uint32_t i = s; i < e; i++) {
            double* a = acc + (size_t)sortedGaussIdx[i] * 11;
            const double* o = pair + (size_t)i * 11;
            for (int k = 0; k < 11; k++) a[k] += o[k];
        }
    }
    for (size_t i = 0; i < (size_t)N * 11; i++) gradPacked[i] = (real)acc[i];
    free(acc); free(pair);
}

/* ======================================================================== *
 * a10  SSIM  (slang/ssim_kernels.slang; window LossUtil.swift:47-54,
 *             GaussianTrainer.swift:308-314)
 * ======================================================================== */
/* gaussian(windowSize, sigma): centre = windowSize / 2.0 (5.5 for 11: off-centre),
 * f32 arithmetic; 2-D window = outer product (f32 matmul). */
GSO_API void gso_ssim_window(int K, float sigma, float* window /*[K*K]*/)
{
    float g[64];
    float center = (float)K / 2.0f, sum = 0.0f;
    for (int x = 0; x < K; x++) {
        float d = (float)x - center;
        g[x] = expf(-(d * d) / (2.0f * (sigma * sigma)));
        sum += g[x];
    }
    for (int x = 0; x < K; x++) g[x] = g[x] / sum;
    for (int i = 0; i < K; i++) for (int j = 0; j < K; j++) window[i * K + j] = g[i] * g[j];
}

/* ssim_forward (:94-155) */
GSO_API void GSO_NAME(ssim_forward)(int H, int W, int C, int K, const real* img1, const real* img2,
                                    const real* window, real* outSsim, real* outMu1, real* outMu2,
                                    real* outSigma1, real* outSigma2, real* outSigma12)
{
    int pad = K / 2;
    const real C1 = R(0.0001), C2 = R(0.0009);
#pragma omp parallel for schedule(static)
    for (int h = 0; h < H; h++)
        for (int w = 0; w < W; w++)
            for (int c = 0; c < C; c++) {
                real mu1 = 0, mu2 = 0, s11 = 0, s22 = 0, s12 = 0;
                for (int ki = 0; ki < K; ki++) {
                    int sh = h + ki - pad;
                    if (sh < 0 || sh >= H) continue;            /* zero contribution outside (:123) */
                    for (int kj = 0; kj < K; kj++) {
                        int sw = w + kj - pad;
                        if (sw < 0 || sw >= W) continue;
                        real wt = window[ki * K + kj];
                        size_t si = ((size_t)sh * W + sw) * C + c;
                        real v1 = img1[si], v2 = img2[si];
                        mu1 = mu1 + wt * v1; mu2 = mu2 + wt * v2;
                        s11 = s11 + wt * v1 * v1; s22 = s22 + wt * v2 * v2; s12 = s12 + wt * v1 * v2;
                    }
                }
                real sig1 = s11 - mu1 * mu1, sig2 = s22 - mu2 * mu2, sig12 = s12 - mu1 * mu2;
                real a = R(2.0) * mu1 * mu2 + C1, b = R(2.0) * sig12 + C2;
                real c_ = mu1 * mu1 + mu2 * mu2 + C1, d = sig1 + sig2 + C2;
                size_t idx = ((size_t)h * W + w) * C + c;
                outSsim[idx] = (a * b) / (c_ * d);
                outMu1[idx] = mu1; outMu2[idx] = mu2;
                outSigma1[idx] = sig1; outSigma2[idx] = sig2; outSigma12[idx] = sig12;
            }
}

/* ssim_backward (:181-266): gather over the K*K window centres containing
 * the pixel, un-flipped weight index, saved mu/sigma maps. */
GSO_API void GSO_NAME(ssim_backward)(int H, int W, int C, int K, const real* gradOut, const real* img1,
                                     const real* img2, const real* window, const real* mu1m, const real* mu2m,
                                     const real* sig1m, const real* sig2m, const real* sig12m, real* gradImg1,
                                     real* gradImg2)
{
    int pad = K / 2;
    const real C1 = R(0.0001), C2 = R(0.0009);
#pragma omp parallel for schedule(static)
    for (int h = 0; h < H; h++)
        for (int w = 0; w < W; w++)
            for (int c = 0; c < C; c++) {
                size_t idx = ((size_t)h * W + w) * C + c;
                real v1 = img1[idx], v2 = img2[idx];
                real g1 = 0, g2 = 0;
                for (int ki = 0; ki < K; ki++) {
                    int cx = h - ki + pad;
                    if (cx < 0 || cx >= H) continue;
                    for (int kj = 0; kj < K; kj++) {
                        int cy = w - kj + pad;
                        if (cy < 0 || cy >= W) continue;
                        real wt = window[ki * K + kj];
                        size_t ci = ((size_t)cx * W + cy) * C + c;
                        real up = gradOut[ci];
                        real m1 = mu1m[ci], m2 = mu2m[ci];
                        /* ssimFromAccumState on (mu1, mu2, E11, E22, E12) and its reverse mode */
                        real E11 = sig1m[ci] + m1 * m1, E22 = sig2m[ci] + m2 * m2, E12 = sig12m[ci] + m1 * m2;
                        real s1 = E11 - m1 * m1, s2 = E22 - m2 * m2, s12 = E12 - m1 * m2;
                        real a = R(2.0) * m1 * m2 + C1, b = R(2.0) * s12 + C2;
                        real c_ = m1 * m1 + m2 * m2 + C1, d = s1 + s2 + C2;
                        real num = a * b, den = c_ * d;
                        real dnum = up / den, dden = -up * num / (den * den);
                        real da = dnum * b, db = dnum * a, dc = dden * d, dd = dden * c_;
                        /* d wrt sigma terms */
                        real ds1 = dd, ds2 = dd, ds12 = R(2.0) * db;
                        real dE11 = ds1, dE22 = ds2, dE12 = ds12;
                        real dm1 = da * R(2.0) * m2 + dc * R(2.0) * m1 - ds1 * R(2.0) * m1 - ds12 * m2;
                        real dm2 = da * R(2.0) * m1 + dc * R(2.0) * m2 - ds2 * R(2.0) * m2 - ds12 * m1;
                        /* updateSsimAccumState reverse: d v1, d v2 */
                        g1 += wt * dm1 + wt * v1 * dE11 + wt * v1 * dE11 + wt * v2 * dE12;
                        g2 += wt * dm2 + wt * v2 * dE22 + wt * v2 * dE22 + wt * v1 * dE12;
                    }
                }
                gradImg1[idx] = g1; gradImg2[idx] = g2;
            }
}

/* ======================================================================== *
 * a11  Loss assembly (GaussianTrainer.swift:689-714, LossUtil.swift:39-41)
 * L = (1-l)*mean|R-G| + l*(1 - mean ssim) + ld * sum(|D-Dgt|*mask)/max(sum mask,1e-6)
 * Returns loss; writes cotangents of render colour [H,W,3] and depth [H,W].
 * depthMask/targetDepth may be NULL when lambdaDepth == 0.
 * ======================================================================== */
GSO_API double GSO_NAME(loss_forward_backward)(int H, int W, const real* render, const real* target,
                                               const real* renderDepth, const real* targetDepth,
                                               const unsigned char* depthMask, real lambdaDssim,
                                               real lambdaDepth, real* cotColor, real* cotDepth,
                                               double* l1Out, double* ssimOut)
{
    size_t n = (size_t)H * W * 3;
    float win[121];
    gso_ssim_window(11, 1.5f, win);
    real winr[121];
    for (int i = 0; i < 121; i++) winr[i] = (real)win[i];
    real* maps = (real*)malloc(sizeof(real) * n * 8);
    real *ss = maps, *m1 = maps + n, *m2 = maps + 2 * n, *s1 = maps + 3 * n, *s2 = maps + 4 * n,
         *s12 = maps + 5 * n, *up = maps + 6 * n, *g2 = maps + 7 * n;
    GSO_NAME(ssim_forward)(H, W, 3, 11, render, target, winr, ss, m1, m2, s1, s2, s12);
    double l1 = 0, sm = 0;
    for (size_t i = 0; i < n; i++) { l1 += r_fabs(render[i] - target[i]); sm += ss[i]; }
    l1 /= (double)n; sm /= (double)n;
    real upv = -lambdaDssim / (real)n;
    for (size_t i = 0; i < n; i++) up[i] = upv;
    GSO_NAME(ssim_backward)(H, W, 3, 11, up, render, target, winr, m1, m2, s1, s2, s12, cotColor, g2);
    real l1w = (R(1.0) - lambdaDssim) / (real)n;
    for (size_t i = 0; i < n; i++) {
        real d = render[i] - target[i];
        real sg = d > 0 ? R(1.0) : (d < 0 ? R(-1.0) : R(0.0));
        cotColor[i] += l1w * sg;
    }
    double depthLoss = 0;
    size_t np = (size_t)H * W;
    if (cotDepth) for (size_t i = 0; i < np; i++) cotDepth[i] = 0;
    if (lambdaDepth != 0 && depthMask && targetDepth && renderDepth) {
        double wsum = 0, acc = 0;
        for (size_t i = 0; i < np; i++) if (depthMask[i]) { wsum += 1; acc += r_fabs(renderDepth[i] - targetDepth[i]); }
        double safe = wsum > 1e-6 ? wsum : 1e-6;
        depthLoss = acc / safe;
        if (cotDepth) for (size_t i = 0; i < np; i++) if (depthMask[i]) {
            real d = renderDepth[i] - targetDepth[i];
            real sg = d > 0 ? R(1.0) : (d < 0 ? R(-1.0) : R(0.0));
            cotDepth[i] = lambdaDepth * sg / (real)safe;
        }
    }
    free(maps);
    if (l1Out) *l1Out = l1;
    if (ssimOut) *ssimOut = sm;
    return (1.0 - (double)lambdaDssim) * l1 + (double)lambdaDssim * (1.0 - sm) + (double)lambdaDepth * depthLoss;
}

/* ======================================================================================================
 * Next row f2: densify / prune (Trainer/GaussianTrainer.swift:317-427 kernels, :724-908 host sequence)
 * No reference test pins this row and the noise comes from MLXRandom: PARITY UNPINNED beyond the kernel text.
 * ==================================================================================================== */

/* accum_grad_norm, GaussianTrainer.swift:320-338.  accumIn may be NULL (zeros). */
GSO_API void GSO_NAME(accum_grad_norm)(int N, const real* xyzGrad, const real* accumIn, real* accumOut)
{
    for (int i = 0; i < N; i++) {
        real gx = xyzGrad[i * 3 + 0], gy = xyzGrad[i * 3 + 1], gz = xyzGrad[i * 3 + 2];
        real norm = r_sqrt(gx * gx + gy * gy + gz * gz);
        accumOut[i] = (accumIn ? accumIn[i] : R(0.0)) + norm;
    }
}

/* classify_gaussians, GaussianTrainer.swift:343-393.  denom is the scalar the host broadcasts to [N] (:796).
 * actions: 0 keep, 1 split, 2 clone, 3 prune; counts 1, 2, 2, 0. */
GSO_API void GSO_NAME(classify_gaussians)(int N, const real* gradAccum, real denom, const real* scales,
                                          int scaleStride, const real* opacity, real gradThreshold,
                                          real maxScaleThresh, real minOpacityThresh, int allowDensify,
                                          int32_t* actions, int32_t* outputCounts)
{
    for (int i = 0; i < N; i++) {
        real g = gradAccum[i];
        real avg = denom > R(0.0) ? g / denom : R(0.0);
        real s0 = r_exp(scales[(size_t)i * scaleStride + 0]);
        real s1 = r_exp(scales[(size_t)i * scaleStride + 1]);
        real s2 = r_exp(scales[(size_t)i * scaleStride + 2]);
        real m01 = s0 > s1 ? s0 : s1;
        real maxScale = m01 > s2 ? m01 : s2;
        real op = R(1.0) / (R(1.0) + r_exp(-opacity[i]));
        int action, cnt;
        if (op < minOpacityThresh) { action = 3; cnt = 0; }
        else if (allowDensify && avg > gradThreshold) {
            if (maxScale > maxScaleThresh) { action = 1; cnt = 2; }
            else { action = 2; cnt = 2; }
        } else { action = 0; cnt = 1; }
        actions[i] = action;
        outputCounts[i] = cnt;
    }
}

/* offsets = cumsum(counts) - counts (GaussianTrainer.swift:813-816); stats = total, keep, split, clone, prune */
GSO_API void gso_densify_offsets(int N, const int32_t* actions, const int32_t* outputCounts, int32_t* offsets,
                                 int64_t stats[5])
{
    int64_t run = 0, h[4] = {0, 0, 0, 0};
    for (int i = 0; i < N; i++) {
        offsets[i] = (int32_t)run;
        run += outputCounts[i];
        h[actions[i] & 3]++;
    }
    stats[0] = run; stats[1] = h[0]; stats[2] = h[1]; stats[3] = h[2]; stats[4] = h[3];
}

/* build_densify_output_map, GaussianTrainer.swift:398-427 */
GSO_API void gso_build_densify_output_map(int N, const int32_t* actions, const int32_t* offsets,
                                          int32_t* gatherIndices, int32_t* noiseMode)
{
    for (int i = 0; i < N; i++) {
        int a = actions[i], o = offsets[i];
        if (a == 0) { gatherIndices[o] = i; noiseMode[o] = 0; }
        else if (a == 1) { gatherIndices[o] = i; noiseMode[o] = 1; gatherIndices[o + 1] = i; noiseMode[o + 1] = 2; }
        else if (a == 2) { gatherIndices[o] = i; noiseMode[o] = 0; gatherIndices[o + 1] = i; noiseMode[o + 1] = 3; }
    }
}

/* Phases 4-5, GaussianTrainer.swift:858-893: gather all six tensors; split children get scale - log(1.6) and
 * xyz +/- mean(exp(source scale)) * 0.1 * noise; clone copies get xyz + 0.01 * noise.  baseNoise [total,3] is an
 * input (MLXRandom.normal in the reference); NULL = the numSplit == numClone == 0 branch (no modification). */
GSO_API void GSO_NAME(densify_gather)(int total, int K, const real* xyz, const real* fdc, const real* frest,
                                      const real* scales, const real* rot, const real* opacity,
                                      const int32_t* gatherIndices, const int32_t* noiseMode, const real* baseNoise,
                                      real* oXyz, real* oFdc, real* oFrest, real* oScales, real* oRot, real* oOpacity)
{
    const int L = (K - 1) * 3;
    const real scaleRed = (real)(-log(1.6));
    for (int j = 0; j < total; j++) {
        const size_t s = (size_t)gatherIndices[j];
        const int mode = noiseMode[j];
        memcpy(oFdc + (size_t)j * 3, fdc + s * 3, sizeof(real) * 3);
        if (L > 0) memcpy(oFrest + (size_t)j * L, frest + s * L, sizeof(real) * L);
        memcpy(oRot + (size_t)j * 4, rot + s * 4, sizeof(real) * 4);
        oOpacity[j] = opacity[s];
        const real isSplit = (mode == 1 || mode == 2) ? R(1.0) : R(0.0);
        for (int a = 0; a < 3; a++) {
            oScales[(size_t)j * 3 + a] = baseNoise ? scales[s * 3 + a] + isSplit * scaleRed : scales[s * 3 + a];
            oXyz[(size_t)j * 3 + a] = xyz[s * 3 + a];
        }
        if (!baseNoise) continue;
        const real e0 = r_exp(scales[s * 3 + 0]), e1 = r_exp(scales[s * 3 + 1]), e2 = r_exp(scales[s * 3 + 2]);
        const real meanScale = ((e0 + e1) + e2) * (R(1.0) / R(3.0));      /* MLX mean = sum * (1/n) */
        const real sign = (mode == 1 ? R(1.0) : R(0.0)) - (mode == 2 ? R(1.0) : R(0.0));
        const real isClone = mode == 3 ? R(1.0) : R(0.0);
        for (int a = 0; a < 3; a++) {
            const real nz = baseNoise[(size_t)j * 3 + a];
            const real splitNoise = sign * meanScale * R(0.1) * nz;
            const real cloneNoise = isClone * R(0.01) * nz;
            oXyz[(size_t)j * 3 + a] = xyz[s * 3 + a] + splitNoise + cloneNoise;
        }
    }
}
