// Does s_setprio change how a SIMD's issue slots are shared on gfx950?  One workgroup of 64 threads per wave slot; waves on a
// SIMD run the same dependent-chain loop (4 independent FMA chains per lane, like a blend sweep); wave "0 of every 4" raises its
// priority.  Prints the cycles the favoured waves and the others take, with the priority on and off.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_prio tools/microbench_prio.hip && /tmp/mb_prio
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, float a, int iters, int usePrio)
{
    const bool fav = (blockIdx.x & 3) == 0;
    if (usePrio && fav) __builtin_amdgcn_s_setprio(3);
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x0 = fmaf(x0, a, 1.0f); x1 = fmaf(x1, a, 1.0f); x2 = fmaf(x2, a, 1.0f); x3 = fmaf(x3, a, 1.0f);
        }
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    const int waves = 256 * 4 * 4;      // four per SIMD
    float* out; unsigned long long* cyc;
    hipMalloc(&out, waves * 64 * 4); hipMalloc(&cyc, waves * 8);
    std::vector<unsigned long long> h(waves);
    for (int prio = 0; prio < 2; prio++) {
        for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, out, cyc, 0.999f, 2000, prio);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
        double f = 0, o = 0; int nf = 0, no = 0;
        for (int i = 0; i < waves; i++) { if ((i & 3) == 0) { f += h[i]; nf++; } else { o += h[i]; no++; } }
        printf("s_setprio %s: favoured waves %.0f cycles, the others %.0f (2000 x 32 dependent-chain FMAs per wave, four waves per SIMD)\n",
               prio ? "3 " : "off", f / nf, o / no);
    }
    return 0;
}
