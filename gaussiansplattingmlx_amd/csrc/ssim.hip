// ssim.hip -- fused SSIM map forward/backward and the training-loss assembly.
//
// Replaces ssim_forward / ssim_backward (slang/ssim_kernels.slang:94-155, 181-266) and the loss
// arithmetic of GaussianTrainer.swift:689-714.  The reference evaluates K*K taps x 5 accumulators per
// element straight from device memory; here a workgroup stages its 16x16 tile plus the (K-1) halo of
// both images (forward) or of the per-centre derivative terms (backward) in LDS once, so each input
// element crosses HBM once per tile instead of K*K times.  Out-of-image taps contribute zero exactly
// as the reference's `continue` does.  HBM-bound: 3P*32 B forward, 3P*40 B backward.
#include "gs_ctx.h"
#include "gs_bwd_prep.h"

namespace gs {

constexpr int ST = 16;          // tile edge
constexpr float SSIM_C1 = 0.0001f, SSIM_C2 = 0.0009f;

// dynamic LDS layout forward: window[K*K] | img1 tile[(ST+K-1)^2] | img2 tile
__global__ __launch_bounds__(ST * ST) void ssim_fwd_kernel(int H, int W, int C, int K, const float* __restrict__ img1,
                                                           const float* __restrict__ img2,
                                                           const float* __restrict__ window, float* __restrict__ oSsim,
                                                           float* __restrict__ oMu1, float* __restrict__ oMu2,
                                                           float* __restrict__ oS1, float* __restrict__ oS2,
                                                           float* __restrict__ oS12)
{
    extern __shared__ float smem[];
    const int pad = K / 2, TW = ST + K - 1;
    float* sw = smem;
    float* t1 = sw + K * K;
    float* t2 = t1 + TW * TW;
    const int tid = threadIdx.x, c = blockIdx.z;
    const int h0 = blockIdx.y * ST, w0 = blockIdx.x * ST;
    for (int i = tid; i < K * K; i += ST * ST) sw[i] = window[i];
    for (int i = tid; i < TW * TW; i += ST * ST) {
        const int r = i / TW, q = i - r * TW;
        const int sh = h0 + r - pad, sc = w0 + q - pad;
        float a = 0.f, b = 0.f;
        if (sh >= 0 && sh < H && sc >= 0 && sc < W) {
            const size_t si = ((size_t)sh * W + sc) * C + c;
            a = img1[si]; b = img2[si];
        }
        t1[i] = a; t2[i] = b;
    }
    __syncthreads();
    const int ly = tid / ST, lx = tid - ly * ST;
    const int h = h0 + ly, w = w0 + lx;
    if (h >= H || w >= W) return;
    float mu1 = 0.f, mu2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
    for (int ki = 0; ki < K; ki++) {
        const float* r1 = t1 + (ly + ki) * TW + lx;
        const float* r2 = t2 + (ly + ki) * TW + lx;
        const float* wr = sw + ki * K;
        for (int kj = 0; kj < K; kj++) {
            const float wt = wr[kj], v1 = r1[kj], v2 = r2[kj];
            mu1 = mu1 + wt * v1; mu2 = mu2 + wt * v2;
            s11 = s11 + wt * v1 * v1; s22 = s22 + wt * v2 * v2; s12 = s12 + wt * v1 * v2;
        }
    }
    const float sig1 = s11 - mu1 * mu1, sig2 = s22 - mu2 * mu2, sig12 = s12 - mu1 * mu2;
    const float a = 2.0f * mu1 * mu2 + SSIM_C1, b = 2.0f * sig12 + SSIM_C2;
    const float c_ = mu1 * mu1 + mu2 * mu2 + SSIM_C1, d = sig1 + sig2 + SSIM_C2;
    const size_t idx = ((size_t)h * W + w) * C + c;
    oSsim[idx] = (a * b) / (c_ * d);
    oMu1[idx] = mu1; oMu2[idx] = mu2; oS1[idx] = sig1; oS2[idx] = sig2; oS12[idx] = sig12;
}

// dynamic LDS layout backward: window[K*K] | 5 derivative planes[(ST+K-1)^2]
// gradOut may be null: then every element's upstream is gradOutConst (fused loss: -lambda/(3P)).
// l1Weight != 0 adds l1Weight * sign(img1 - img2) to grad_img1 (the L1 term of the loss); g2 may be null.
__global__ __launch_bounds__(ST * ST) void ssim_bwd_kernel(int H, int W, int C, int K, const float* __restrict__ gradOut,
                                                           float gradOutConst, const float* __restrict__ img1,
                                                           const float* __restrict__ img2,
                                                           const float* __restrict__ window,
                                                           const float* __restrict__ mu1m, const float* __restrict__ mu2m,
                                                           const float* __restrict__ s1m, const float* __restrict__ s2m,
                                                           const float* __restrict__ s12m, float* __restrict__ g1,
                                                           float* __restrict__ g2, float l1Weight)
{
    extern __shared__ float smem[];
    const int pad = K / 2, TW = ST + K - 1, lo = K - 1 - pad;
    float* sw = smem;
    float* pM1 = sw + K * K;
    float* pE11 = pM1 + TW * TW;
    float* pE12 = pE11 + TW * TW;
    float* pM2 = pE12 + TW * TW;
    float* pE22 = pM2 + TW * TW;
    const int tid = threadIdx.x, c = blockIdx.z;
    const int h0 = blockIdx.y * ST, w0 = blockIdx.x * ST;
    for (int i = tid; i < K * K; i += ST * ST) sw[i] = window[i];
    for (int i = tid; i < TW * TW; i += ST * ST) {
        const int r = i / TW, q = i - r * TW;
        const int ch = h0 + r - lo, cw = w0 + q - lo;     // window centre
        float dm1 = 0.f, dE11 = 0.f, dE12 = 0.f, dm2 = 0.f, dE22 = 0.f;
        if (ch >= 0 && ch < H && cw >= 0 && cw < W) {
            const size_t ci = ((size_t)ch * W + cw) * C + c;
            const float up = gradOut ? gradOut[ci] : gradOutConst;
            const float m1 = mu1m[ci], m2 = mu2m[ci];
            const float E11 = s1m[ci] + m1 * m1, E22 = s2m[ci] + m2 * m2, E12 = s12m[ci] + m1 * m2;
            const float s1 = E11 - m1 * m1, s2 = E22 - m2 * m2, s12 = E12 - m1 * m2;
            const float a = 2.0f * m1 * m2 + SSIM_C1, b = 2.0f * s12 + SSIM_C2;
            const float c_ = m1 * m1 + m2 * m2 + SSIM_C1, d = s1 + s2 + SSIM_C2;
            const float num = a * b, den = c_ * d;
            const float dnum = up / den, dden = -up * num / (den * den);
            const float da = dnum * b, db = dnum * a, dc = dden * d, dd = dden * c_;
            dE11 = dd; dE22 = dd; dE12 = 2.0f * db;
            dm1 = da * 2.0f * m2 + dc * 2.0f * m1 - dd * 2.0f * m1 - dE12 * m2;
            dm2 = da * 2.0f * m1 + dc * 2.0f * m2 - dd * 2.0f * m2 - dE12 * m1;
        }
        pM1[i] = dm1; pE11[i] = dE11; pE12[i] = dE12; pM2[i] = dm2; pE22[i] = dE22;
    }
    __syncthreads();
    const int ly = tid / ST, lx = tid - ly * ST;
    const int h = h0 + ly, w = w0 + lx;
    if (h >= H || w >= W) return;
    float A = 0.f, B = 0.f, Cc = 0.f, A2 = 0.f, B2 = 0.f;
    // centre (h - ki + pad, w - kj + pad)  ->  plane coords (ly + K-1 - ki, lx + K-1 - kj); un-flipped weight
    for (int ki = 0; ki < K; ki++) {
        const int row = (ly + K - 1 - ki) * TW + lx + K - 1;
        const float* wr = sw + ki * K;
        for (int kj = 0; kj < K; kj++) {
            const float wt = wr[kj];
            const int o = row - kj;
            A += wt * pM1[o]; B += wt * pE11[o]; Cc += wt * pE12[o]; A2 += wt * pM2[o]; B2 += wt * pE22[o];
        }
    }
    const size_t idx = ((size_t)h * W + w) * C + c;
    const float v1 = img1[idx], v2 = img2[idx];
    float r1 = A + 2.0f * v1 * B + v2 * Cc;
    if (l1Weight != 0.0f) {
        const float d = v1 - v2;
        r1 += l1Weight * (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f));
    }
    g1[idx] = r1;
    if (g2) g2[idx] = A2 + 2.0f * v2 * B2 + v1 * Cc;
}

// ---- fused loss kernel (K = 11) ---------------------------------------------------------------------------
// One launch computes, for a TX x TY tile of one channel: the SSIM statistics of the (TX+10) x (TY+10) window centres
// whose windows reach the tile (from a (TX+20) x (TY+20) input patch, separably: the window is an outer product
// g (x) g), the SSIM value of the tile's own pixels (partial sum of the loss), the three derivative planes of the
// centres, and their separable correlation back onto the tile's pixels, plus the L1 term:
//     cot(render) = l1w * sign(R - G) + [A + 2 R B + G C],   A,B,C = corr(g (x) g, upstream * d ssim / d(mu1, E11, E12))
// Nothing but the two images is read from HBM and nothing but the cotangent and 2 partial sums per block is
// written: the five statistic maps of the reference's two-kernel structure (ssim_kernels.slang:94-266) never
// leave LDS.  Separable sums differ from the reference's 121-tap order by rounding only.
//
// Tile size.  The kernel is bound by vector-instruction issue (round 2's SQ counters: 34.3 M wave-instructions per
// launch = 0.90 of the issue slots of its 67 us), and what it issues is mostly the halo: per output pixel the four
// passes cost ((TY+20)(TX+10) + (TY+10)(TX+10)) 55 + ((TY+10) TX + TY TX) 33 multiply-adds / (TX TY) -- 433 for the
// 16 x 16 tiles of rounds 1-2, 288 for 32 x 32, 264 for 64 x 32, 176 without any halo.  The tile is a template
// parameter; every output still sums its taps in the same order (k ascending), so the values do not depend on it.
constexpr int LK = 11, LPAD = 5;

// Block -> (tile, channel): workgroups go round-robin over the 8 XCDs, each with its own L2, and a block reads a
// patch of interleaved RGB for one channel.  With the plain (x, y, channel) grid every XCD ended up fetching the
// whole of both images (114 MB of HBM-side reads for 15 MB of input); here XCD x owns a contiguous band of tiles and
// its consecutive blocks are the three channels of one tile, so the second and third find the lines in that L2.
struct SsimTaps {
    float g[LK];
};

// outputs per work item of a separable pass: the shortest strip (from `lo` up) with which the pass's items fit the block's
// threads in one round -- a second round that only a few threads take part in costs as much as a full one (32 x 32 tile,
// 512 threads: strips of 4 columns make 572 items, strips of 5 make 468)
constexpr int loss_strip(int lines, int outputs, int threads, int lo, int hi)
{
    for (int w = lo; w <= hi; w++)
        if (lines * ((outputs + w - 1) / w) <= threads) return w;
    return lo;
}

// TCACHE: the TARGET's windowed statistics (mean and mean of squares under the window) are the same at every visit of a
// training view, and computing them is two of the five statistic planes of both forward passes.  0: no cache; 1: compute
// them as always and keep them (tcache: [2][3][H][W], caller-owned, one per view: gs_set_loss_target_cache); 2: read them
// -- three planes instead of five in both passes and in LDS (48 KB instead of 65: three blocks per CU).  The values are
// the very floats mode 1 computed, so the loss and its cotangent are bit-identical with and without the cache.
template <int TX, int TY, int NT, int TCACHE>
__global__ __launch_bounds__(NT) void loss_fused_kernel(int H, int W, int ntx, int nty,
                                                        const float* __restrict__ img1,
                                                        const float* __restrict__ img2, float upstream,
                                                        float l1Weight, float* __restrict__ cot,
                                                        float* __restrict__ partials, BwdPrepArgs prep, int prepBlocks, int cutBlocks, SsimTaps taps,
                                                        float* __restrict__ tcache)
{
    constexpr int LCX = TX + LK - 1, LCY = TY + LK - 1;        // window centres that reach the tile
    constexpr int LIX = LCX + LK - 1, LIY = LCY + LK - 1;      // input patch
    constexpr int NPL = TCACHE == 2 ? 3 : 5;                   // statistic planes computed here: (mu1, E11, E12) or (mu1, mu2, E11, E22, E12)
    // The first prepBlocks workgroups (a multiple of 8, so the tiles keep their XCDs) are not loss work at all: they
    // prepare the fused blend BACKWARD of the forward whose render this loss is taken of -- the first GS_ITEM_PARTS build its
    // work-item list (a scan of the per-block sweep lengths), the others clear its accumulator (gs_bwd_prep.h).  Both depend
    // on the forward only, so they ride along here instead of standing between this kernel and the backward (as do the
    // cutBlocks blocks behind the item blocks, which renew the view's depth cuts).
    if ((int)blockIdx.x < prepBlocks) {
        __shared__ uint32_t prepSm[17];
        constexpr int IP = GS_ITEM_PARTS;
        if ((int)blockIdx.x < IP) bwd_items_scan<GS_SEG_LEN>(prep, prepSm, (int)blockIdx.x);
        else if ((int)blockIdx.x < IP + cutBlocks) bwd_cut_renew(prep, (int)(blockIdx.x - IP) * NT + (int)threadIdx.x);
        else bwd_clear_part(prep, blockIdx.x - IP - cutBlocks, (size_t)prepBlocks - IP - cutBlocks);
        return;
    }
    const unsigned lossBlock = blockIdx.x - (unsigned)prepBlocks;
    // the eleven taps come in as a kernel argument (scalar registers; computed on the host exactly as gs_ssim_window
    // does): every block used to spend its first microsecond on eleven serial expf calls in eleven lanes
    const float* const g = taps.g;
    // the derivative planes reuse the input patches, the backward's row sums reuse the forward's
    // (dynamic LDS: a 32 x 32 tile needs 65 KB, more than a static allocation may hold; the CU has 160 KB)
    extern __shared__ float lossLds[];
    float* const smIn = lossLds;                       // in1 | in2 [2 LIY LIX], later D[3][LCY*LCX]
    float* const smH = lossLds + 2 * LIY * LIX;        // Hs[NPL][LIY*LCX], later HB[3][LCY*TX]
    __shared__ float red[NT / 64][2];
    float* const in1 = smIn;
    float* const in2 = smIn + LIY * LIX;
    float (*const Hs)[LIY * LCX] = reinterpret_cast<float (*)[LIY * LCX]>(smH);   // horizontal sums: rows = input rows, cols = centre cols
    float (*const D)[LCY * LCX] = reinterpret_cast<float (*)[LCY * LCX]>(smIn);   // derivative planes at centres (times upstream); inputs are dead by then
    float (*const HB)[LCY * TX] = reinterpret_cast<float (*)[LCY * TX]>(smH);     // horizontal pass of the backward correlation; Hs is dead by then
    static_assert(3 * LCY * LCX <= 2 * LIY * LIX && 3 * LCY * TX <= NPL * LIY * LCX, "aliased planes must fit");
    static_assert(NT % 64 == 0 && (TX * TY) % NT == 0 && TX % 4 == 0, "tile / thread shape");
    const int tid = threadIdx.x;
    const int nTiles = ntx * nty, perXcd = (nTiles + 7) >> 3;
    const int seq = (int)(lossBlock >> 3), tileId = (int)(lossBlock & 7u) * perXcd + seq / 3, c = seq % 3;
    if (tileId >= nTiles) return;          // whole block: the grid is 8 * 3 * perXcd
    const int ty = tileId / ntx, tx = tileId - ty * ntx;
    const int h0 = ty * TY, w0 = tx * TX;
    {   // the two input patches.  All of a thread's loads are issued before the first is used (unconditional, from
        // clamped addresses): as a loop with the loads under `if (inside the image)` every iteration waited for its
        // own pair -- six memory latencies in a row at the head of every block of a latency-bound kernel
        constexpr int NLD = (LIY * LIX + NT - 1) / NT;
        float va[NLD], vb[NLD];
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int i = tid + k * NT;
            const int r = i / LIX, q = i - r * LIX;
            const int sh = h0 - 2 * LPAD + r, sw = w0 - 2 * LPAD + q;
            const bool ok = i < LIY * LIX && sh >= 0 && sh < H && sw >= 0 && sw < W;
            const int shc = min(max(sh, 0), H - 1), swc = min(max(sw, 0), W - 1);
            const size_t si = ((size_t)shc * W + swc) * 3 + c;
            const float a = img1[si], b = img2[si];
            va[k] = ok ? a : 0.0f;
            vb[k] = ok ? b : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int i = tid + k * NT;
            if (i < LIY * LIX) { in1[i] = va[k]; in2[i] = vb[k]; }
        }
    }
    __syncthreads();
    // this thread's own pixels (pixel p = tid + k NT of the tile, row-major), before the patches are reused
    constexpr int PPT = TX * TY / NT;
    float own1[PPT], own2[PPT];
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        const int p = tid + k * NT, ly = p / TX, lx = p - ly * TX;
        own1[k] = in1[(ly + 2 * LPAD) * LIX + lx + 2 * LPAD];
        own2[k] = in2[(ly + 2 * LPAD) * LIX + lx + 2 * LPAD];
    }
    // The four separable passes are LDS-read bound if every tap is fetched per output, so each work item produces a
    // short strip of outputs along the filter axis from a register window.  Every output still sums its taps in the
    // same order, k ascending.
    //
    // forward horizontal: centre column q uses input columns q .. q+10.  Item = (input row r, strip of HW centre columns).
    {
        constexpr int HW = loss_strip(LIY, LCX, NT, 4, 8), SH = (LCX + HW - 1) / HW;
        for (int it = tid; it < LIY * SH; it += NT) {
            const int r = it / SH, sidx = it - r * SH;
            const int q0 = HW * sidx, nout = min(HW, LCX - q0);
            float a[HW + 10], b[HW + 10];
#pragma unroll
            for (int i = 0; i < HW + 10; i++) {
                const bool ok = q0 + i < LIX;
                a[i] = ok ? in1[r * LIX + q0 + i] : 0.0f;
                b[i] = ok ? in2[r * LIX + q0 + i] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < HW; j++) {
                if (j < nout) {
                    float s1 = 0.f, s2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
                    for (int k = 0; k < LK; k++) {
                        const float w = g[k], v1 = a[j + k], v2 = b[j + k];
                        s1 = fmaf(w, v1, s1);
                        s11 = fmaf(w * v1, v1, s11); s12 = fmaf(w * v1, v2, s12);
                        if constexpr (TCACHE != 2) { s2 = fmaf(w, v2, s2); s22 = fmaf(w * v2, v2, s22); }
                    }
                    const int o = r * LCX + q0 + j;
                    if constexpr (TCACHE != 2) { Hs[0][o] = s1; Hs[1][o] = s2; Hs[2][o] = s11; Hs[3][o] = s22; Hs[4][o] = s12; }
                    else { Hs[0][o] = s1; Hs[1][o] = s11; Hs[2][o] = s12; }
                }
            }
        }
    }
    __syncthreads();
    // forward vertical at the centres, SSIM value and its derivatives.  Item = (centre column q, strip of VW centre rows).
    float ssimSum = 0.0f;
    {
        constexpr int VW = loss_strip(LCX, LCY, NT, 3, 6), SV = (LCY + VW - 1) / VW;
        for (int it = tid; it < LCX * SV; it += NT) {
            const int sp = it / LCX, q = it - sp * LCX;
            const int p0 = VW * sp, nout = min(VW, LCY - p0);
            float st[NPL][VW];
#pragma unroll
            for (int pl = 0; pl < NPL; pl++) {
                float col[VW + 10];
#pragma unroll
                for (int i = 0; i < VW + 10; i++) col[i] = (p0 + i < LIY) ? Hs[pl][(p0 + i) * LCX + q] : 0.0f;
#pragma unroll
                for (int j = 0; j < VW; j++) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < LK; k++) acc = fmaf(g[k], col[j + k], acc);
                    st[pl][j] = acc;
                }
            }
#pragma unroll
            for (int j = 0; j < VW; j++) {
                if (j < nout) {
                    const int pp = p0 + j;
                    const int ch = h0 - LPAD + pp, cw = w0 - LPAD + q;
                    float dm1 = 0.f, dE11 = 0.f, dE12 = 0.f;
                    if (ch >= 0 && ch < H && cw >= 0 && cw < W) {
                        const size_t ci = ((size_t)c * H + ch) * W + cw, cplane = (size_t)3 * H * W;
                        float m1, m2, E11, E22, E12;
                        if constexpr (TCACHE == 2) {
                            m1 = st[0][j]; E11 = st[1][j]; E12 = st[2][j];
                            m2 = tcache[ci]; E22 = tcache[cplane + ci];
                        } else {
                            m1 = st[0][j]; m2 = st[1][j]; E11 = st[2][j]; E22 = st[3][j]; E12 = st[4][j];
                            // every centre is some tile's own pixel exactly once: that tile keeps the target's statistics
                            if (TCACHE == 1 && pp >= LPAD && pp < LPAD + TY && q >= LPAD && q < LPAD + TX) {
                                tcache[ci] = m2; tcache[cplane + ci] = E22;
                            }
                        }
                        const float s1 = E11 - m1 * m1, s2 = E22 - m2 * m2, s12 = E12 - m1 * m2;
                        const float a = 2.0f * m1 * m2 + SSIM_C1, b = 2.0f * s12 + SSIM_C2;
                        const float c_ = m1 * m1 + m2 * m2 + SSIM_C1, d = s1 + s2 + SSIM_C2;
                        const float num = a * b, den = c_ * d;
                        if (pp >= LPAD && pp < LPAD + TY && q >= LPAD && q < LPAD + TX) ssimSum += num / den;   // the tile's own pixels
                        const float dnum = upstream / den, dden = -upstream * num / (den * den);
                        const float da = dnum * b, db = dnum * a, dc = dden * d, ddd = dden * c_;
                        dE11 = ddd; dE12 = 2.0f * db;
                        dm1 = da * 2.0f * m2 + dc * 2.0f * m1 - ddd * 2.0f * m1 - dE12 * m2;
                    }
                    const int o = pp * LCX + q;
                    D[0][o] = dm1; D[1][o] = dE11; D[2][o] = dE12;
                }
            }
        }
    }
    __syncthreads();
    // backward horizontal: pixel column x gathers centre columns x+10-k with the un-flipped weight g[k].
    // Item = (centre row p, strip of 4 pixel columns).
    {
        constexpr int SB = TX / 4;
        for (int it = tid; it < LCY * SB; it += NT) {
            const int p = it / SB, x0 = 4 * (it - p * SB);
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                float dv[14];
#pragma unroll
                for (int i = 0; i < 14; i++) dv[i] = D[pl][p * LCX + x0 + i];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < LK; k++) acc = fmaf(g[k], dv[j + 2 * LPAD - k], acc);
                    HB[pl][p * TX + x0 + j] = acc;
                }
            }
        }
    }
    __syncthreads();
    float l1 = 0.0f;
#pragma unroll
    for (int kk = 0; kk < PPT; kk++) {
        const int p = tid + kk * NT, ly = p / TX, lx = p - ly * TX;
        const int h = h0 + ly, w = w0 + lx;
        if (h < H && w < W) {
            float A = 0.f, B = 0.f, Cc = 0.f;
#pragma unroll
            for (int k = 0; k < LK; k++) {
                const float wt = g[k];
                const int o = (ly + 2 * LPAD - k) * TX + lx;
                A = fmaf(wt, HB[0][o], A); B = fmaf(wt, HB[1][o], B); Cc = fmaf(wt, HB[2][o], Cc);
            }
            const float v1 = own1[kk], v2 = own2[kk];
            const float d = v1 - v2;
            l1 += fabsf(d);
            cot[((size_t)h * W + w) * 3 + c] = A + 2.0f * v1 * B + v2 * Cc + l1Weight * (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f));
        }
    }
    // block partial sums: |R-G| and ssim
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { l1 += __shfl_xor(l1, s, 64); ssimSum += __shfl_xor(ssimSum, s, 64); }
    if ((tid & 63) == 0) { red[tid >> 6][0] = l1; red[tid >> 6][1] = ssimSum; }
    __syncthreads();
    if (tid == 0) {
        const int b = (c * nty + ty) * ntx + tx;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int k = 0; k < NT / 64; k++) { s0 += red[k][0]; s1 += red[k][1]; }
        partials[b * 4 + 0] = s0;
        partials[b * 4 + 1] = s1;
        partials[b * 4 + 2] = 0.0f; partials[b * 4 + 3] = 0.0f;
    }
}

// depth-loss partial sums only (used beside loss_fused_kernel when lambda_depth != 0)
__global__ __launch_bounds__(256) void depth_reduce_kernel(size_t np, const float* __restrict__ renderDepth,
                                                           const float* __restrict__ targetDepth,
                                                           const unsigned char* __restrict__ mask,
                                                           float* __restrict__ partials)
{
    __shared__ float sm[4][2];
    float c = 0.f, d = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < np; i += stride)
        if (mask[i]) { c += fabsf(renderDepth[i] - targetDepth[i]); d += 1.0f; }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { c += __shfl_xor(c, s, 64); d += __shfl_xor(d, s, 64); }
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = c; sm[threadIdx.x >> 6][1] = d; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 4 + 2] += sm[0][0] + sm[1][0] + sm[2][0] + sm[3][0];
        partials[blockIdx.x * 4 + 3] += sm[0][1] + sm[1][1] + sm[2][1] + sm[3][1];
    }
}

// ---- loss reductions ------------------------------------------------------------------------
// one workgroup: loss_out = {total, l1, mean ssim, depth loss}; aux[0] = max(sum mask, 1e-6)
__global__ __launch_bounds__(1024) void loss_final_kernel(int nb, const float* __restrict__ partials, double n3,
                                                         float lambdaDssim, float lambdaDepth, float* __restrict__ lossOut,
                                                         float* __restrict__ aux)
{
    __shared__ double sm[16][4];
    double a = 0, b = 0, c = 0, d = 0;
    const float4* p4 = reinterpret_cast<const float4*>(partials);
    for (int i = threadIdx.x; i < nb; i += 1024) {
        const float4 v = p4[i];
        a += v.x; b += v.y; c += v.z; d += v.w;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        a += __shfl_xor(a, s, 64); b += __shfl_xor(b, s, 64); c += __shfl_xor(c, s, 64); d += __shfl_xor(d, s, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        sm[w][0] = a; sm[w][1] = b; sm[w][2] = c; sm[w][3] = d;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = b = c = d = 0;
        for (int w = 0; w < 16; w++) { a += sm[w][0]; b += sm[w][1]; c += sm[w][2]; d += sm[w][3]; }
        const double l1 = a / n3, ss = b / n3;
        const double safe = d > 1e-6 ? d : 1e-6;
        const double dl = (lambdaDepth != 0.0f) ? c / safe : 0.0;
        lossOut[0] = (float)((1.0 - (double)lambdaDssim) * l1 + (double)lambdaDssim * (1.0 - ss) +
                             (double)lambdaDepth * dl);
        lossOut[1] = (float)l1; lossOut[2] = (float)ss; lossOut[3] = (float)dl;
        aux[0] = (float)safe;
    }
}

__global__ void depth_cot_kernel(size_t np, const float* __restrict__ renderDepth, const float* __restrict__ targetDepth,
                                 const unsigned char* __restrict__ mask, float lambdaDepth, const float* __restrict__ aux,
                                 float* __restrict__ cotDepth)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    float v = 0.0f;
    if (mask && mask[i]) {
        const float d = renderDepth[i] - targetDepth[i];
        v = lambdaDepth * (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f)) / aux[0];
    }
    cotDepth[i] = v;
}

// ---- launchers --------------------------------------------------------------------------------
int launch_ssim_forward(gs_ctx* c, int H, int W, int C, int K, const float* img1, const float* img2,
                        const float* window, float* ssim, float* mu1, float* mu2, float* s1, float* s2, float* s12)
{
    if (H == 0 || W == 0 || C == 0) return GS_OK;
    const int TW = ST + K - 1;
    const size_t lds = sizeof(float) * ((size_t)K * K + 2 * (size_t)TW * TW);
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3(gs_div_up(W, ST), gs_div_up(H, ST), C), dim3(ST * ST), lds, c->stream, H,
                       W, C, K, img1, img2, window, ssim, mu1, mu2, s1, s2, s12);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_ssim_backward(gs_ctx* c, int H, int W, int C, int K, const float* gradOut, float gradOutConst,
                         const float* img1, const float* img2, const float* window, const float* mu1,
                         const float* mu2, const float* s1, const float* s2, const float* s12, float* g1, float* g2,
                         float l1Weight)
{
    if (H == 0 || W == 0 || C == 0) return GS_OK;
    const int TW = ST + K - 1;
    const size_t lds = sizeof(float) * ((size_t)K * K + 5 * (size_t)TW * TW);
    hipLaunchKernelGGL(ssim_bwd_kernel, dim3(gs_div_up(W, ST), gs_div_up(H, ST), C), dim3(ST * ST), lds, c->stream, H,
                       W, C, K, gradOut, gradOutConst, img1, img2, window, mu1, mu2, s1, s2, s12, g1, g2, l1Weight);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_loss(gs_ctx* c, const float* render, const float* target, const float* renderDepth,
                const float* targetDepth, const unsigned char* depthMask, float lambdaDssim, float lambdaDepth,
                float* lossOut, float* cotColor, float* cotDepth)
{
    const int H = c->H, W = c->W;
    const size_t np = (size_t)H * W, n3 = np * 3;
    const bool depthOn = lambdaDepth != 0.0f && depthMask && targetDepth && renderDepth;
#ifndef GS_LOSS_TX
#define GS_LOSS_TX 32
#define GS_LOSS_TY 32
#define GS_LOSS_NT 512
#endif
    constexpr int LTX = GS_LOSS_TX, LTY = GS_LOSS_TY, LNT = GS_LOSS_NT;
    const dim3 grid(gs_div_up(W, LTX), gs_div_up(H, LTY), 3);
    const int nb = (int)(grid.x * grid.y * grid.z);
    if (nb > c->lossPartialBlocks) return GS_ERR_SIZE_MISMATCH;
    const int perXcd = (int)(grid.x * grid.y + 7) / 8;
    // the loss of a fused forward, taken through the library: the backward's preparation rides along (see the kernel)
    SsimTaps taps;
    {   // gaussian(windowSize: 11, sigma: 1.5) with the reference's off-centre 5.5 (LossUtil.swift:48-53), as gs_ssim_window
        float sum = 0.0f;
        for (int x = 0; x < LK; x++) {
            const float d = (float)x - 5.5f;
            taps.g[x] = expf(-(d * d) / (2.0f * (1.5f * 1.5f)));
            sum += taps.g[x];
        }
        for (int x = 0; x < LK; x++) taps.g[x] = taps.g[x] / sum;
    }
    BwdPrepArgs prep = {};
    int prepBlocks = 0, cutBlocks = 0;
    if (c->fast16 && c->fwd.valid && !c->fwd.consumed && !c->fwd.blendBackwardDone && c->fwd.N > 0 && c->itemBlock &&
        c->fwd.statePlanes != 0) {         // (a render-only forward has no backward to prepare)
        const uint32_t qs = (uint32_t)blend_backward_v2_grid(c);
        fill_bwd_prep(c, c->fwd.N, qs, prep);
        const size_t parts = (prep.clearCount + 4095) / 4096;
        cutBlocks = prep.cutStore ? gs_div_up(prep.nBlocks, LNT) : 0;
        prepBlocks = (int)(((GS_ITEM_PARTS + cutBlocks + (parts < 496 ? parts : 496)) + 7) / 8 * 8);
        c->fwd.bwdPrepared = true;
        c->fwd.preparedQueueStart = qs;
        c->fwd.preparedN = c->fwd.N;
    }
    constexpr size_t lds5 = sizeof(float) * ((size_t)2 * (LTY + 20) * (LTX + 20) + (size_t)5 * (LTY + 20) * (LTX + 10));
    constexpr size_t lds3 = sizeof(float) * ((size_t)2 * (LTY + 20) * (LTX + 20) + (size_t)3 * (LTY + 20) * (LTX + 10));
    static bool ldsAllowed = false;
    if (!ldsAllowed) {
        GS_HIP_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&loss_fused_kernel<LTX, LTY, LNT, 0>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5));
        GS_HIP_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&loss_fused_kernel<LTX, LTY, LNT, 1>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5));
        GS_HIP_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&loss_fused_kernel<LTX, LTY, LNT, 2>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        ldsAllowed = true;
    }
    // the target's windowed statistics: kept / reused when the caller handed a per-view cache (gs_set_loss_target_cache)
    const int tmode = c->lossTargetCache ? (c->lossTargetCacheFilled ? 2 : 1) : 0;
    const dim3 lgrid(prepBlocks + 8 * 3 * perXcd), lblock(LNT);
    const float up = -lambdaDssim / (float)n3, l1w = (1.0f - lambdaDssim) / (float)n3;
    if (tmode == 2)
        hipLaunchKernelGGL((loss_fused_kernel<LTX, LTY, LNT, 2>), lgrid, lblock, lds3, c->stream, H, W, (int)grid.x, (int)grid.y, render, target,
                           up, l1w, cotColor, c->lossPartials, prep, prepBlocks, cutBlocks, taps, c->lossTargetCache);
    else if (tmode == 1)
        hipLaunchKernelGGL((loss_fused_kernel<LTX, LTY, LNT, 1>), lgrid, lblock, lds5, c->stream, H, W, (int)grid.x, (int)grid.y, render, target,
                           up, l1w, cotColor, c->lossPartials, prep, prepBlocks, cutBlocks, taps, c->lossTargetCache);
    else
        hipLaunchKernelGGL((loss_fused_kernel<LTX, LTY, LNT, 0>), lgrid, lblock, lds5, c->stream, H, W, (int)grid.x, (int)grid.y, render, target,
                           up, l1w, cotColor, c->lossPartials, prep, prepBlocks, cutBlocks, taps, nullptr);
    if (tmode == 1) c->lossTargetCacheFilled = true;
    if (depthOn)
        hipLaunchKernelGGL(depth_reduce_kernel, dim3(nb < 512 ? nb : 512), dim3(256), 0, c->stream, np, renderDepth,
                           targetDepth, depthMask, c->lossPartials);
    float* aux = c->lossPartials + (size_t)c->lossPartialBlocks * 4;
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(1024), 0, c->stream, nb, c->lossPartials, (double)n3,
                       lambdaDssim, depthOn ? lambdaDepth : 0.0f, lossOut, aux);
    GS_HIP_CHECK(c, hipGetLastError());
    if (cotDepth) {
        hipLaunchKernelGGL(depth_cot_kernel, dim3(gs_div_up(np, 256)), dim3(256), 0, c->stream, np, renderDepth,
                           targetDepth, depthOn ? depthMask : nullptr, lambdaDepth, aux, cotDepth);
        GS_HIP_CHECK(c, hipGetLastError());
    }
    return GS_OK;
}

}  // namespace gs
