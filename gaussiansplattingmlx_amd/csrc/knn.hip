// knn.hip -- point-cloud initialisation helper: distTopK of Trainer/GaussianModel.swift:11-31.
// For each query point: mean of its k smallest squared distances to ALL N points (the point itself included, as in
// the reference: dist2 = |x_i - x_j|^2 over the full set, top-k of the negated row).  Brute force, the N points
// streamed through LDS in tiles; one lane per query keeps its k (<= 8) smallest in registers.
#include "gs_ctx.h"

namespace gs {

constexpr int KNN_THREADS = 256, KNN_TILE = 1024, KNN_MAXK = 8;

__global__ __launch_bounds__(KNN_THREADS) void dist_topk_kernel(int N, int k, int qBegin, int qCount,
                                                                const float* __restrict__ xyz, float* __restrict__ out)
{
    __shared__ float tx[KNN_TILE], ty[KNN_TILE], tz[KNN_TILE];
    const int qi = blockIdx.x * KNN_THREADS + threadIdx.x;
    const bool active = qi < qCount;
    const int q = qBegin + (active ? qi : 0);
    const float x = xyz[3 * q], y = xyz[3 * q + 1], z = xyz[3 * q + 2];
    float best[KNN_MAXK];                 // ascending
#pragma unroll
    for (int i = 0; i < KNN_MAXK; i++) best[i] = __builtin_inff();
    for (int base = 0; base < N; base += KNN_TILE) {
        const int cnt = min(KNN_TILE, N - base);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += KNN_THREADS) {
            tx[i] = xyz[3 * (base + i)]; ty[i] = xyz[3 * (base + i) + 1]; tz[i] = xyz[3 * (base + i) + 2];
        }
        __syncthreads();
        for (int i = 0; i < cnt; i++) {
            // sum(square(diff)) over the last axis, in x, y, z order (MLX.sum over 3 elements)
            const float dx = x - tx[i], dy = y - ty[i], dz = z - tz[i];
            float d = dx * dx + dy * dy + dz * dz;
            if (d < best[KNN_MAXK - 1]) {
#pragma unroll
                for (int s = 0; s < KNN_MAXK; s++) {          // insertion into the sorted register list
                    const float lo = fminf(best[s], d);
                    d = fmaxf(best[s], d);
                    best[s] = lo;
                }
            }
        }
    }
    if (active) {
        float sum = 0.0f;
#pragma unroll
        for (int s = 0; s < KNN_MAXK; s++)
            if (s < k) sum += best[s];
        out[q] = sum * (1.0f / (float)k);                      // MLX mean = sum * (1/n)
    }
}

int launch_dist_topk(gs_ctx* c, int N, int k, int qBegin, int qCount, const float* xyz, float* out)
{
    if (qCount <= 0 || N <= 0) return GS_OK;
    hipLaunchKernelGGL(dist_topk_kernel, dim3(gs_div_up(qCount, KNN_THREADS)), dim3(KNN_THREADS), 0, c->stream, N, k,
                       qBegin, qCount, xyz, out);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

}  // namespace gs
