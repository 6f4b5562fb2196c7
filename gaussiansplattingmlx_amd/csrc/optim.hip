// optim.hip -- fused Adam over the flat parameter arena.
// One HBM pass: reads p, g, m, v (16 B/element), writes p, m, v (12 B/element) = 28 B/element, float4-vectorised.
#include "gs_ctx.h"

namespace gs {

struct AdamSegs {
    long long end[8];
    float lr[8];
    int n;
};

__device__ __forceinline__ float seg_lr(const AdamSegs& s, long long i)
{
    float lr = s.lr[0];
#pragma unroll
    for (int k = 1; k < 8; k++)
        if (k < s.n && i >= s.end[k - 1]) lr = s.lr[k];
    return lr;
}

__global__ __launch_bounds__(256) void adam_kernel(long long n, float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, AdamSegs segs,
                                                   float b1, float b2, float eps, float gscale,
                                                   const uint32_t* __restrict__ gate, uint32_t* __restrict__ seen,
                                                   const float* __restrict__ add, long long addN)
{
    // add (round 6, the data-parallel step): a second gradient term for the LEADING addN elements, g[i] + add[i] -- the
    // view-direction part of the xyz gradient, rebuilt locally while the rest was being all-reduced (projection.hip,
    // sh_views_dir_adam_kernel); 16-byte aligned, readable up to addN rounded up to four (the pad zero)
    if (*gate) {            // the step's forward overflowed its reserved pair capacity: no update from a blank render
        if (seen && blockIdx.x == 0 && threadIdx.x == 0) *seen = 1u;      // (gs_set_gate_seen: "some step was gated")
        return;
    }
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        if (add && i * 4 < addN) {
            const float4 aa = reinterpret_cast<const float4*>(add)[i];
            gg.x += aa.x; gg.y += aa.y; gg.z += aa.z; gg.w += aa.w;
        }
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* pa = &pp.x; const float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float lr = seg_lr(segs, i * 4 + k);
            const float gr = ga[k] * gscale;
            ma[k] = b1 * ma[k] + (1.0f - b1) * gr;
            va[k] = b2 * va[k] + (1.0f - b2) * gr * gr;
            pa[k] = pa[k] - gs_adam_delta(lr, ma[k], va[k], eps);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail
    for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float lr = seg_lr(segs, i);
        const float gr = (g[i] + ((add && i < addN) ? add[i] : 0.0f)) * gscale;
        const float mn = b1 * m[i] + (1.0f - b1) * gr;
        const float vn = b2 * v[i] + (1.0f - b2) * gr * gr;
        m[i] = mn; v[i] = vn;
        p[i] = p[i] - gs_adam_delta(lr, mn, vn, eps);
    }
}

int launch_adam(gs_ctx* c, long long n, float* params, const float* grads, float* m, float* v, int nseg,
                const long long* segEnd, const float* segLr, float b1, float b2, float eps, float gradScale,
                const float* add, long long addN)
{
    if (n == 0) return GS_OK;
    AdamSegs s;
    s.n = nseg;
    for (int i = 0; i < 8; i++) { s.end[i] = i < nseg ? segEnd[i] : n; s.lr[i] = i < nseg ? segLr[i] : 0.0f; }
    GsStageTimer t(c, GS_STAGE_ADAM);
    long long nb = (n / 4 + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, c->stream, n, params, grads, m, v, s, b1, b2, eps,
                       gradScale, c->adamGate, c->gateSeen, add, add ? addN : 0);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

}  // namespace gs
