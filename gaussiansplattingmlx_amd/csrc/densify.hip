// densify.hip -- densify / prune (Trainer/GaussianTrainer.swift:317-427 kernels, :766-907 sequence)
//
// All of it is HBM-bound streaming over N (classify, scan, map) or over the output rows (gather): one coalesced pass
// each.  It runs once per 100 iterations, so the design goal is one host synchronisation (the output count the
// caller needs to allocate the new tensors -- the reference's `.item()`, :816) and no per-tensor temporaries.
#include <math.h>

#include "gs_ctx.h"

namespace gs {

constexpr int DN_THREADS = 256;
constexpr int DN_SCAN_ITEMS = 4;                       // ints per thread in the scan kernels
constexpr int DN_SCAN_TILE = DN_THREADS * DN_SCAN_ITEMS;

// accum_grad_norm, GaussianTrainer.swift:320-338
__global__ __launch_bounds__(DN_THREADS) void accum_grad_norm_kernel(int N, const float* __restrict__ xyzGrad,
                                                                     const float* accumIn, float* accumOut)
{
    const int i = blockIdx.x * DN_THREADS + threadIdx.x;
    if (i >= N) return;
    const float gx = xyzGrad[3 * i], gy = xyzGrad[3 * i + 1], gz = xyzGrad[3 * i + 2];
    const float norm = sqrtf(gx * gx + gy * gy + gz * gz);   // compiled -ffp-contract=off; sqrtf is correctly rounded
    accumOut[i] = (accumIn ? accumIn[i] : 0.0f) + norm;
}

// classify_gaussians, GaussianTrainer.swift:343-393
__global__ __launch_bounds__(DN_THREADS) void classify_kernel(int N, const float* __restrict__ gradAccum, float denom,
                                                              const float* __restrict__ scales, int scaleStride,
                                                              const float* __restrict__ opacity, float gradThreshold,
                                                              float maxScaleThresh, float minOpacityThresh,
                                                              int allowDensify, int* __restrict__ actions,
                                                              int* __restrict__ outputCounts)
{
    const int i = blockIdx.x * DN_THREADS + threadIdx.x;
    if (i >= N) return;
    const float g = gradAccum[i];
    const float avg = denom > 0.0f ? g / denom : 0.0f;
    const float* s = scales + (size_t)i * scaleStride;
    const float maxScale = fmaxf(fmaxf(expf(s[0]), expf(s[1])), expf(s[2]));
    const float op = 1.0f / (1.0f + expf(-opacity[i]));
    int action, cnt;
    if (op < minOpacityThresh) { action = 3; cnt = 0; }
    else if (allowDensify && avg > gradThreshold) { action = maxScale > maxScaleThresh ? 1 : 2; cnt = 2; }
    else { action = 0; cnt = 1; }
    actions[i] = action;
    outputCounts[i] = cnt;
}

// ---- exclusive scan of the output counts + action histogram (MLX cumsum - counts, :813-816) -------------------
__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// block-wide exclusive scan of one int per thread; returns the exclusive prefix, *total = block sum
__device__ __forceinline__ int block_excl_scan(int v, int* total)
{
    __shared__ int waveSum[DN_THREADS / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int incl = wave_incl_scan(v, lane);
    if (lane == 63) waveSum[wv] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < DN_THREADS / 64; w++) {
        if (w < wv) base += waveSum[w];
        tot += waveSum[w];
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(DN_THREADS) void dn_tile_sums_kernel(int N, const int* __restrict__ counts,
                                                                  const int* __restrict__ actions,
                                                                  int* __restrict__ tileSums, uint32_t* hist4)
{
    const int base = blockIdx.x * DN_SCAN_TILE + threadIdx.x * DN_SCAN_ITEMS;
    int s = 0;
    uint32_t h = 0;                                   // four 8-bit counters (<= DN_SCAN_ITEMS each)
#pragma unroll
    for (int k = 0; k < DN_SCAN_ITEMS; k++)
        if (base + k < N) { s += counts[base + k]; h += 1u << (8 * (actions[base + k] & 3)); }
    int tot;
    block_excl_scan(s, &tot);
    if (threadIdx.x == 0) tileSums[blockIdx.x] = tot;
    // histogram: wave reduce the packed byte counters one class at a time
#pragma unroll
    for (int a = 0; a < 4; a++) {
        int c = (h >> (8 * a)) & 255;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(&hist4[a], (uint32_t)c);
    }
}

// one block: exclusive scan of the tile sums in place, total -> *totalOut
__global__ __launch_bounds__(DN_THREADS) void dn_tile_offsets_kernel(int nTiles, int* tileSums, uint32_t* totalOut)
{
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int t0 = 0; t0 < nTiles; t0 += DN_THREADS) {
        const int t = t0 + threadIdx.x;
        const int v = t < nTiles ? tileSums[t] : 0;
        int tot;
        const int ex = block_excl_scan(v, &tot);
        const int c = carry;
        if (t < nTiles) tileSums[t] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *totalOut = (uint32_t)carry;
}

__global__ __launch_bounds__(DN_THREADS) void dn_offsets_kernel(int N, const int* __restrict__ counts,
                                                                const int* __restrict__ tileOffsets,
                                                                int* __restrict__ offsets)
{
    const int base = blockIdx.x * DN_SCAN_TILE + threadIdx.x * DN_SCAN_ITEMS;
    int v[DN_SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < DN_SCAN_ITEMS; k++) { v[k] = base + k < N ? counts[base + k] : 0; s += v[k]; }
    int tot;
    int run = tileOffsets[blockIdx.x] + block_excl_scan(s, &tot);
#pragma unroll
    for (int k = 0; k < DN_SCAN_ITEMS; k++) {
        if (base + k < N) offsets[base + k] = run;
        run += v[k];
    }
}

// build_densify_output_map, GaussianTrainer.swift:398-427
__global__ __launch_bounds__(DN_THREADS) void build_map_kernel(int N, int total, const int* __restrict__ actions,
                                                               const int* __restrict__ offsets,
                                                               int* __restrict__ gather, int* __restrict__ noiseMode)
{
    const int i = blockIdx.x * DN_THREADS + threadIdx.x;
    if (i >= N) return;
    const int a = actions[i], o = offsets[i];
    if (a < 0 || a > 2 || o < 0 || o + (a == 0 ? 1 : 2) > total) return;   // prune, or a map that does not fit `total`
    gather[o] = i;
    noiseMode[o] = a == 1 ? 1 : 0;
    if (a != 0) { gather[o + 1] = i; noiseMode[o + 1] = a == 1 ? 2 : 3; }
}

// ---- phases 4-5 (:858-893) ---------------------------------------------------------------------------------
// generic row gather: out[j, :] = in[gather[j], :]; one thread per float, consecutive threads walk a row
__global__ __launch_bounds__(DN_THREADS) void gather_rows_kernel(long long totalElems, int rowLen,
                                                                 const float* __restrict__ in,
                                                                 const int* __restrict__ gather,
                                                                 float* __restrict__ out)
{
    const long long e = (long long)blockIdx.x * DN_THREADS + threadIdx.x;
    if (e >= totalElems) return;
    const long long j = e / rowLen;
    const int k = (int)(e - j * rowLen);
    out[e] = in[(size_t)gather[j] * rowLen + k];
}

// the small tensors in one pass: xyz (+ noise), scales (+ split reduction), rotation, opacity, features_dc
__global__ __launch_bounds__(DN_THREADS) void gather_small_kernel(
    int total, const float* __restrict__ xyz, const float* __restrict__ fdc, const float* __restrict__ scales,
    const float* __restrict__ rot, const float* __restrict__ opacity, const int* __restrict__ gather,
    const int* __restrict__ noiseMode, const float* __restrict__ baseNoise, float scaleReduction,
    float* __restrict__ oXyz, float* __restrict__ oFdc, float* __restrict__ oScales, float* __restrict__ oRot,
    float* __restrict__ oOpacity)
{
    const int j = blockIdx.x * DN_THREADS + threadIdx.x;
    if (j >= total) return;
    const size_t s = (size_t)gather[j];
    const int mode = noiseMode[j];
    const float sc[3] = {scales[s * 3], scales[s * 3 + 1], scales[s * 3 + 2]};
    const float isSplit = (mode == 1 || mode == 2) ? 1.0f : 0.0f;
    if (baseNoise) {
        const float mean = __fmul_rn(__fadd_rn(__fadd_rn(expf(sc[0]), expf(sc[1])), expf(sc[2])), 1.0f / 3.0f);
        const float sign = (mode == 1 ? 1.0f : 0.0f) - (mode == 2 ? 1.0f : 0.0f);
        const float isClone = mode == 3 ? 1.0f : 0.0f;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float nz = baseNoise[(size_t)j * 3 + a];
            const float splitNoise = __fmul_rn(__fmul_rn(__fmul_rn(sign, mean), 0.1f), nz);
            const float cloneNoise = __fmul_rn(__fmul_rn(isClone, 0.01f), nz);
            oXyz[(size_t)j * 3 + a] = __fadd_rn(__fadd_rn(xyz[s * 3 + a], splitNoise), cloneNoise);
            oScales[(size_t)j * 3 + a] = __fadd_rn(sc[a], __fmul_rn(isSplit, scaleReduction));
        }
    } else {
#pragma unroll
        for (int a = 0; a < 3; a++) { oXyz[(size_t)j * 3 + a] = xyz[s * 3 + a]; oScales[(size_t)j * 3 + a] = sc[a]; }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) oFdc[(size_t)j * 3 + a] = fdc[s * 3 + a];
#pragma unroll
    for (int a = 0; a < 4; a++) oRot[(size_t)j * 4 + a] = rot[s * 4 + a];
    oOpacity[j] = opacity[s];
}

// ---- launchers ---------------------------------------------------------------------------------------------
int launch_accum_grad_norm(gs_ctx* c, int N, const float* xyzGrad, const float* accumIn, float* accumOut)
{
    if (N == 0) return GS_OK;
    hipLaunchKernelGGL(accum_grad_norm_kernel, dim3(gs_div_up(N, DN_THREADS)), dim3(DN_THREADS), 0, c->stream, N,
                       xyzGrad, accumIn, accumOut);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_classify(gs_ctx* c, int N, const float* gradAccum, float denom, const float* scales, int scaleStride,
                    const float* opacity, float gradThreshold, float maxScale, float minOpacity, int allowDensify,
                    int* actions, int* outputCounts)
{
    if (N == 0) return GS_OK;
    hipLaunchKernelGGL(classify_kernel, dim3(gs_div_up(N, DN_THREADS)), dim3(DN_THREADS), 0, c->stream, N, gradAccum,
                       denom, scales, scaleStride, opacity, gradThreshold, maxScale, minOpacity, allowDensify, actions,
                       outputCounts);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_densify_offsets(gs_ctx* c, int N, const int* actions, const int* outputCounts, int* offsets,
                           long long stats[5])
{
    for (int i = 0; i < 5; i++) stats[i] = 0;
    if (N == 0) return GS_OK;
    const int nTiles = gs_div_up(N, DN_SCAN_TILE);
    if (nTiles > c->densifyTileCap) {
        if (c->densifyTiles) GS_HIP_CHECK(c, hipFree(c->densifyTiles));
        c->densifyTiles = nullptr;
        c->densifyTileCap = 0;
        GS_HIP_CHECK(c, hipMalloc(&c->densifyTiles, sizeof(int) * (size_t)(nTiles + 8)));
        c->densifyTileCap = nTiles;
    }
    uint32_t* cnt = reinterpret_cast<uint32_t*>(c->densifyTiles + nTiles);   // [0..3] histogram, [4] total
    GS_HIP_CHECK(c, hipMemsetAsync(cnt, 0, sizeof(uint32_t) * 8, c->stream));
    hipLaunchKernelGGL(dn_tile_sums_kernel, dim3(nTiles), dim3(DN_THREADS), 0, c->stream, N, outputCounts, actions,
                       c->densifyTiles, cnt);
    hipLaunchKernelGGL(dn_tile_offsets_kernel, dim3(1), dim3(DN_THREADS), 0, c->stream, nTiles, c->densifyTiles,
                       cnt + 4);
    hipLaunchKernelGGL(dn_offsets_kernel, dim3(nTiles), dim3(DN_THREADS), 0, c->stream, N, outputCounts,
                       c->densifyTiles, offsets);
    GS_HIP_CHECK(c, hipGetLastError());
    uint32_t h[8];
    GS_HIP_CHECK(c, hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    stats[0] = h[4]; stats[1] = h[0]; stats[2] = h[1]; stats[3] = h[2]; stats[4] = h[3];
    return GS_OK;
}

int launch_build_densify_map(gs_ctx* c, int N, const int* actions, const int* offsets, int total, int* gather,
                             int* noiseMode)
{
    if (total > 0) {
        GS_HIP_CHECK(c, hipMemsetAsync(gather, 0, sizeof(int) * (size_t)total, c->stream));     // initValue: 0 (:852)
        GS_HIP_CHECK(c, hipMemsetAsync(noiseMode, 0, sizeof(int) * (size_t)total, c->stream));
    }
    if (N == 0) return GS_OK;
    hipLaunchKernelGGL(build_map_kernel, dim3(gs_div_up(N, DN_THREADS)), dim3(DN_THREADS), 0, c->stream, N, total,
                       actions, offsets, gather, noiseMode);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_densify_gather(gs_ctx* c, int total, int K, const float* xyz, const float* fdc, const float* frest,
                          const float* scales, const float* rot, const float* opacity, const int* gather,
                          const int* noiseMode, const float* baseNoise, float* oXyz, float* oFdc, float* oFrest,
                          float* oScales, float* oRot, float* oOpacity)
{
    if (total == 0) return GS_OK;
    const float scaleReduction = (float)(-log(1.6));                    // Float(-log(1.6)), :866
    hipLaunchKernelGGL(gather_small_kernel, dim3(gs_div_up(total, DN_THREADS)), dim3(DN_THREADS), 0, c->stream, total,
                       xyz, fdc, scales, rot, opacity, gather, noiseMode, baseNoise, scaleReduction, oXyz, oFdc,
                       oScales, oRot, oOpacity);
    const int L = (K - 1) * 3;
    if (L > 0) {
        const long long elems = (long long)total * L;
        hipLaunchKernelGGL(gather_rows_kernel, dim3(gs_div_up(elems, DN_THREADS)), dim3(DN_THREADS), 0, c->stream,
                           elems, L, frest, gather, oFrest);
    }
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

}  // namespace gs
