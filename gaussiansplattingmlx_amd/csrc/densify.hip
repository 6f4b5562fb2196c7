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

// (the action histogram: per scan tile, summed by dn_tile_offsets_kernel's one workgroup.  Rounds 1-4 added every wave's
// four counts to four global words -- 4.7 k same-address atomics at 300 k Gaussians, ~6 ns each in series: 28 of the
// kernel's 32 us)
__global__ __launch_bounds__(DN_THREADS) void dn_tile_sums_kernel(int N, const int* __restrict__ counts,
                                                                  const int* __restrict__ actions,
                                                                  int* __restrict__ tileSums, uint32_t* __restrict__ tileHist)
{
    __shared__ int wh[DN_THREADS / 64][4];
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
        if ((threadIdx.x & 63) == 0) wh[threadIdx.x >> 6][a] = c;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        int c = 0;
#pragma unroll
        for (int w = 0; w < DN_THREADS / 64; w++) c += wh[w][threadIdx.x];
        tileHist[(size_t)blockIdx.x * 4 + threadIdx.x] = (uint32_t)c;
    }
}

// one block: exclusive scan of the tile sums in place, total -> cntOut[4]; the tiles' action counts summed -> cntOut[0..3]
__global__ __launch_bounds__(DN_THREADS) void dn_tile_offsets_kernel(int nTiles, int* tileSums, const uint32_t* __restrict__ tileHist,
                                                                     uint32_t* cntOut)
{
    __shared__ int carry;
    __shared__ uint32_t hs[4];
    if (threadIdx.x == 0) carry = 0;
    if (threadIdx.x < 4) hs[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t h[4] = {0u, 0u, 0u, 0u};
    for (int t = threadIdx.x; t < nTiles; t += DN_THREADS) {
#pragma unroll
        for (int a = 0; a < 4; a++) h[a] += tileHist[(size_t)t * 4 + a];
    }
#pragma unroll
    for (int a = 0; a < 4; a++) {
        uint32_t c = h[a];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d, 64);
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(&hs[a], c);      // (LDS, four waves)
    }
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
    if (threadIdx.x == 0) cntOut[4] = (uint32_t)carry;
    if (threadIdx.x < 4) cntOut[threadIdx.x] = hs[threadIdx.x];
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

// ---- the event without a drain of the queue (round 5) ---------------------------------------------------------------
// The reference reads the output count on the host (`.item()`, :813-817) and builds everything behind it from that number;
// rounds 1-4 mirrored the read: the queue drained, the host then sized and queued the map, the noise, the gather and the
// commit on an idle device (~0.3 ms of idle per event; 3 % of a 20-step bench).  Now the count stays on the device for the
// kernels that need it: dn_plan_kernel leaves the event's PLAN -- new count, whether anything changes, the action counts --
// in device words (and in pinned host memory, behind an event of its own), the map and the gather run over capacity-sized
// grids and read the plan, the noise is a counter-based generator keyed by (seed, output row) instead of a tensor of
// `total` rows, and the host waits for the PLAN ALONE -- with the gather and the resets already queued behind it, the device
// has work while the host lays out the new model and queues the next step.
// plan words: 0 new count (= total when the event applies, else N)  1 applies  2 total  3 keep  4 split  5 clone  6 prune  7 N
__global__ void dn_plan_kernel(int N, const uint32_t* __restrict__ cnt, uint32_t* __restrict__ plan, uint32_t* __restrict__ planHost)
{
    const uint32_t total = cnt[4], keep = cnt[0], split = cnt[1], clone = cnt[2], prune = cnt[3];
    // the reference's early-outs as a predicate: all pruned (:828-832) or nothing to do (:819-826, :843-847) leave the model as it is
    const uint32_t applies = (total > 0u && (split | clone | prune) != 0u) ? 1u : 0u;
    const uint32_t w[8] = {applies ? total : (uint32_t)N, applies, total, keep, split, clone, prune, (uint32_t)N};
    for (int i = 0; i < 8; i++) { plan[i] = w[i]; planHost[i] = w[i]; }
}

// build_densify_output_map under a plan: the same map when the event applies, the identity when it does not
__global__ __launch_bounds__(DN_THREADS) void build_map_planned_kernel(int N, int cap, const uint32_t* __restrict__ plan,
                                                                       const int* __restrict__ actions,
                                                                       const int* __restrict__ offsets,
                                                                       int* __restrict__ gather, int* __restrict__ noiseMode)
{
    const int i = blockIdx.x * DN_THREADS + threadIdx.x;
    if (i >= N) return;
    if (!plan[1]) { if (i < cap) { gather[i] = i; noiseMode[i] = 0; } return; }
    const int total = (int)min(plan[2], (uint32_t)cap);
    const int a = actions[i], o = offsets[i];
    if (a < 0 || a > 2 || o < 0 || o + (a == 0 ? 1 : 2) > total) return;
    gather[o] = i;
    noiseMode[o] = a == 1 ? 1 : 0;
    if (a != 0) { gather[o + 1] = i; noiseMode[o + 1] = a == 1 ? 2 : 3; }
}

// The planned gather into a PACKED arena (round 6, ABI 6): a data-parallel step all-reduces the leading geometry slice of the
// gradient arena, so its model is laid out with every tensor's segment sized by the new count, padded to four floats -- a
// layout whose tensor starts depend on the count the host does not know yet.  They are computed here, behind the plan:
// table[t] = base + the floats of the segments in front of tensor t in `order` (tensor ids in arena order: 0 xyz, 1 f_dc,
// 2 f_rest, 3 scales, 4 rotation, 5 opacity), each of per[t] * n floats rounded up to a multiple of four.
__global__ void dn_packed_table_kernel(const uint32_t* __restrict__ plan, float* base, int K, int o0, int o1, int o2, int o3, int o4,
                                       int o5, float** __restrict__ table, int cap)
{
    // A plan that does not fit the staging buffer (N_new > cap: the host learns it from the plan words, regrows and gathers
    // again) must still leave every write of THIS gather inside the buffer: the starts are laid out for the `cap` rows the gather
    // can write at most.  (Round 6, found by tools/soak_dp.py: laid out for N_new they pushed the last tensors' rows past the
    // buffer's end -- silent corruption at the run's first regrow, a write fault at its second.)
#ifdef GS_PACKED_TABLE_UNCLAMPED      // (experiment build: the behaviour before the fix, to see that the test catches it)
    const long long n = (long long)plan[0];
#else
    const long long n = min((long long)plan[0], (long long)cap);
#endif
    const int order[6] = {o0, o1, o2, o3, o4, o5};
    const int per[6] = {3, 3, 3 * (K - 1), 3, 4, 1};
    long long off = 0;
    for (int i = 0; i < 6; i++) {
        table[order[i]] = base + off;
        off += (n * per[order[i]] + 3) & ~3LL;
    }
}

// Standard normal noise of output row j, three components, from a counter-based generator (Philox4x32-10, counter = the
// row, key = the seed; Box-Muller on its four words): the same numbers whatever the number of rows is and whoever asks --
// the planned gather below, or gs_densify_noise filling a [total, 3] tensor for the gather that takes one.  (The reference
// draws MLXRandom.normal([total, 3]), :881: its stream is an input here, parity unpinned.)
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}
__device__ __forceinline__ void densify_noise3(unsigned long long seed, uint32_t row, float (&z)[3])
{
    uint32_t c[4] = {row, 0u, 0x64656e73u, 0x69667921u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    const float u0 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = ((float)(c[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
    float sa, ca;
    sincosf(6.283185307179586f * u1, &sa, &ca);
    z[0] = ra * ca; z[1] = ra * sa; z[2] = rb * cosf(6.283185307179586f * u3);
}
__global__ __launch_bounds__(DN_THREADS) void densify_noise_kernel(unsigned long long seed, int rows, float* __restrict__ out)
{
    const int j = blockIdx.x * DN_THREADS + threadIdx.x;
    if (j >= rows) return;
    float z[3];
    densify_noise3(seed, (uint32_t)j, z);
    out[(size_t)j * 3] = z[0]; out[(size_t)j * 3 + 1] = z[1]; out[(size_t)j * 3 + 2] = z[2];
}

// ---- phases 4-5 (:858-893) ---------------------------------------------------------------------------------
// generic row gather: out[j, :] = in[gather[j], :]; one thread per float, consecutive threads walk a row
// plan != nullptr: the rows are the plan's new count (totalElems then bounds the grid: capacity x rowLen)
__global__ __launch_bounds__(DN_THREADS) void gather_rows_kernel(long long totalElems, int rowLen,
                                                                 const float* __restrict__ in,
                                                                 const int* __restrict__ gather,
                                                                 float* __restrict__ out, const uint32_t* __restrict__ plan,
                                                                 float* const* __restrict__ table)
{
    const long long e = (long long)blockIdx.x * DN_THREADS + threadIdx.x;
    if (plan && e >= (long long)plan[0] * rowLen) return;
    if (e >= totalElems) return;
    if (table) out = table[2];          // (packed: the f_rest segment's start, known on the device only)
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
    float* __restrict__ oOpacity, const uint32_t* __restrict__ plan, unsigned long long noiseSeed, float* const* __restrict__ table)
{
    const int j = blockIdx.x * DN_THREADS + threadIdx.x;
    if (j >= total) return;
    if (plan && (uint32_t)j >= plan[0]) return;          // (planned: `total` is the capacity the grid covers)
    if (table) { oXyz = table[0]; oFdc = table[1]; oScales = table[3]; oRot = table[4]; oOpacity = table[5]; }      // (packed)
    const size_t s = (size_t)gather[j];
    const int mode = noiseMode[j];
    const float sc[3] = {scales[s * 3], scales[s * 3 + 1], scales[s * 3 + 2]};
    const float isSplit = (mode == 1 || mode == 2) ? 1.0f : 0.0f;
    // planned: the row's own noise from the generator (nothing to add where the event changes nothing: modes are all 0)
    const bool ownNoise = plan != nullptr && plan[1] != 0u && (plan[4] | plan[5]) != 0u;
    if (baseNoise || ownNoise) {
        float gen[3] = {0.f, 0.f, 0.f};
        if (!baseNoise) densify_noise3(noiseSeed, (uint32_t)j, gen);
        const float mean = __fmul_rn(__fadd_rn(__fadd_rn(expf(sc[0]), expf(sc[1])), expf(sc[2])), 1.0f / 3.0f);
        const float sign = (mode == 1 ? 1.0f : 0.0f) - (mode == 2 ? 1.0f : 0.0f);
        const float isClone = mode == 3 ? 1.0f : 0.0f;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float nz = baseNoise ? baseNoise[(size_t)j * 3 + a] : gen[a];
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

// the same, four floats per thread, for rows of a multiple of four floats between 16-byte aligned tensors (K = 25: 72 floats;
// the one-float kernel above moved its 220 MB at 2.7 TB/s, an integer division per element)
__global__ __launch_bounds__(DN_THREADS) void gather_rows4_kernel(long long totalQuads, int rowQuads,
                                                                  const float4* __restrict__ in,
                                                                  const int* __restrict__ gather,
                                                                  float4* __restrict__ out, const uint32_t* __restrict__ plan,
                                                                  float* const* __restrict__ table)
{
    const long long e = (long long)blockIdx.x * DN_THREADS + threadIdx.x;
    if (plan && e >= (long long)plan[0] * rowQuads) return;
    if (e >= totalQuads) return;
    if (table) out = reinterpret_cast<float4*>(table[2]);      // (segments are multiples of four floats: 16-B aligned with the base)
    const long long j = e / rowQuads;
    const int k = (int)(e - j * rowQuads);
    out[e] = in[(size_t)gather[j] * rowQuads + k];
}

// table != nullptr: `out` is the BASE of the packed arena (its alignment is the segments' alignment)
static void launch_gather_rows(gs_ctx* c, long long rows, int L, const float* in, const int* gather, float* out, const uint32_t* plan,
                               float* const* table = nullptr)
{
    if ((L & 3) == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
        const long long quads = rows * (L / 4);
        hipLaunchKernelGGL(gather_rows4_kernel, dim3(gs_div_up(quads, DN_THREADS)), dim3(DN_THREADS), 0, c->stream, quads, L / 4,
                           reinterpret_cast<const float4*>(in), gather, reinterpret_cast<float4*>(out), plan, table);
    } else {
        const long long elems = rows * L;
        hipLaunchKernelGGL(gather_rows_kernel, dim3(gs_div_up(elems, DN_THREADS)), dim3(DN_THREADS), 0, c->stream, elems, L, in,
                           gather, out, plan, table);
    }
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

static int densify_scan(gs_ctx* c, int N, const int* actions, const int* outputCounts, int* offsets, uint32_t** cntOut);

int launch_densify_offsets(gs_ctx* c, int N, const int* actions, const int* outputCounts, int* offsets,
                           long long stats[5])
{
    for (int i = 0; i < 5; i++) stats[i] = 0;
    if (N == 0) return GS_OK;
    uint32_t* cnt = nullptr;
    const int rc = densify_scan(c, N, actions, outputCounts, offsets, &cnt);
    if (rc) return rc;
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
                       oScales, oRot, oOpacity, nullptr, 0ull, nullptr);
    const int L = (K - 1) * 3;
    if (L > 0) launch_gather_rows(c, total, L, frest, gather, oFrest, nullptr);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

// ---- the planned event (kernels above) -----------------------------------------------------------------------
static int densify_scan(gs_ctx* c, int N, const int* actions, const int* outputCounts, int* offsets, uint32_t** cntOut)
{
    const int nTiles = gs_div_up(N, DN_SCAN_TILE);
    if (nTiles > c->densifyTileCap) {
        GS_HIP_CHECK(c, hipStreamSynchronize(c->stream));      // (kernels of an earlier event may still read the old scratch)
        if (c->densifyTiles) GS_HIP_CHECK(c, hipFree(c->densifyTiles));
        c->densifyTiles = nullptr;
        c->densifyTileCap = 0;
        GS_HIP_CHECK(c, hipMalloc(&c->densifyTiles, sizeof(int) * (size_t)(5 * nTiles + 8)));
        c->densifyTileCap = nTiles;
    }
    // scratch: [nTiles] tile sums -> offsets | [8] counts: [0..3] action histogram, [4] total | [nTiles][4] histogram per tile
    uint32_t* cnt = reinterpret_cast<uint32_t*>(c->densifyTiles + nTiles);
    uint32_t* tileHist = cnt + 8;
    hipLaunchKernelGGL(dn_tile_sums_kernel, dim3(nTiles), dim3(DN_THREADS), 0, c->stream, N, outputCounts, actions,
                       c->densifyTiles, tileHist);
    hipLaunchKernelGGL(dn_tile_offsets_kernel, dim3(1), dim3(DN_THREADS), 0, c->stream, nTiles, c->densifyTiles, tileHist,
                       cnt);
    hipLaunchKernelGGL(dn_offsets_kernel, dim3(nTiles), dim3(DN_THREADS), 0, c->stream, N, outputCounts,
                       c->densifyTiles, offsets);
    GS_HIP_CHECK(c, hipGetLastError());
    *cntOut = cnt;
    return GS_OK;
}

int launch_densify_plan(gs_ctx* c, int N, const int* actions, const int* outputCounts, int* offsets)
{
    if (!c->densifyPlan) {
        GS_HIP_CHECK(c, hipMalloc((void**)&c->densifyPlan, 8 * sizeof(uint32_t)));
        GS_HIP_CHECK(c, hipHostMalloc((void**)&c->densifyPlanHost, 8 * sizeof(uint32_t), hipHostMallocMapped));
        GS_HIP_CHECK(c, hipHostGetDevicePointer((void**)&c->densifyPlanHostDev, c->densifyPlanHost, 0));
        GS_HIP_CHECK(c, hipEventCreateWithFlags(&c->densifyDone, hipEventDisableTiming));
    }
    c->densifyPlanned = false;
    if (N == 0) {      // nothing to scan: the plan says so (new count 0, nothing applies)
        GS_HIP_CHECK(c, hipMemsetAsync(c->densifyPlan, 0, 8 * sizeof(uint32_t), c->stream));
        for (int i = 0; i < 8; i++) c->densifyPlanHost[i] = 0;
    } else {
        uint32_t* cnt = nullptr;
        const int rc = densify_scan(c, N, actions, outputCounts, offsets, &cnt);
        if (rc) return rc;
        hipLaunchKernelGGL(dn_plan_kernel, dim3(1), dim3(1), 0, c->stream, N, cnt, c->densifyPlan, c->densifyPlanHostDev);
        GS_HIP_CHECK(c, hipGetLastError());
    }
    GS_HIP_CHECK(c, hipEventRecord(c->densifyDone, c->stream));
    c->densifyPlanned = true;
    return GS_OK;
}

int densify_plan_read(gs_ctx* c, int wait, long long stats[8], int* ready)
{
    *ready = 0;
    if (!c->densifyPlanned) { c->err = "gs_densify_plan_read: no gs_densify_plan on this context"; return GS_ERR_NO_FORWARD; }
    // busy-wait: the answer unblocks the launches of the next step, and a blocking wait wakes up too late to keep the
    // queue filled (api.hip, settle_cut_forward)
    for (;;) {
        const hipError_t q = hipEventQuery(c->densifyDone);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) GS_HIP_CHECK(c, q);
        if (!wait) return GS_OK;
    }
    for (int i = 0; i < 8; i++) stats[i] = (long long)c->densifyPlanHost[i];
    *ready = 1;
    return GS_OK;
}

int launch_build_densify_map_planned(gs_ctx* c, int N, const int* actions, const int* offsets, int cap, int* gather, int* noiseMode)
{
    if (cap > 0) {
        GS_HIP_CHECK(c, hipMemsetAsync(gather, 0, sizeof(int) * (size_t)cap, c->stream));     // initValue: 0 (:852)
        GS_HIP_CHECK(c, hipMemsetAsync(noiseMode, 0, sizeof(int) * (size_t)cap, c->stream));
    }
    if (N == 0) return GS_OK;
    hipLaunchKernelGGL(build_map_planned_kernel, dim3(gs_div_up(N, DN_THREADS)), dim3(DN_THREADS), 0, c->stream, N, cap,
                       c->densifyPlan, actions, offsets, gather, noiseMode);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_densify_gather_planned(gs_ctx* c, int cap, int K, const float* xyz, const float* fdc, const float* frest,
                                  const float* scales, const float* rot, const float* opacity, const int* gather,
                                  const int* noiseMode, unsigned long long noiseSeed, float* oXyz, float* oFdc, float* oFrest,
                                  float* oScales, float* oRot, float* oOpacity)
{
    if (cap == 0) return GS_OK;
    const float scaleReduction = (float)(-log(1.6));                    // Float(-log(1.6)), :866
    hipLaunchKernelGGL(gather_small_kernel, dim3(gs_div_up(cap, DN_THREADS)), dim3(DN_THREADS), 0, c->stream, cap,
                       xyz, fdc, scales, rot, opacity, gather, noiseMode, nullptr, scaleReduction, oXyz, oFdc,
                       oScales, oRot, oOpacity, c->densifyPlan, noiseSeed, nullptr);
    const int L = (K - 1) * 3;
    if (L > 0) launch_gather_rows(c, cap, L, frest, gather, oFrest, c->densifyPlan);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_densify_gather_planned_packed(gs_ctx* c, int cap, int K, const float* xyz, const float* fdc, const float* frest,
                                         const float* scales, const float* rot, const float* opacity, const int* gather,
                                         const int* noiseMode, unsigned long long noiseSeed, float* outBase, const int order[6])
{
    if (cap == 0) return GS_OK;
    if (!c->densifyTable) GS_HIP_CHECK(c, hipMalloc((void**)&c->densifyTable, 8 * sizeof(float*)));
    hipLaunchKernelGGL(dn_packed_table_kernel, dim3(1), dim3(1), 0, c->stream, c->densifyPlan, outBase, K, order[0], order[1],
                       order[2], order[3], order[4], order[5], c->densifyTable, cap);
    const float scaleReduction = (float)(-log(1.6));                    // Float(-log(1.6)), :866
    hipLaunchKernelGGL(gather_small_kernel, dim3(gs_div_up(cap, DN_THREADS)), dim3(DN_THREADS), 0, c->stream, cap,
                       xyz, fdc, scales, rot, opacity, gather, noiseMode, nullptr, scaleReduction, nullptr, nullptr,
                       nullptr, nullptr, nullptr, c->densifyPlan, noiseSeed, c->densifyTable);
    const int L = (K - 1) * 3;
    if (L > 0) launch_gather_rows(c, cap, L, frest, gather, outBase, c->densifyPlan, c->densifyTable);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_densify_noise(gs_ctx* c, unsigned long long seed, int rows, float* out)
{
    if (rows == 0) return GS_OK;
    hipLaunchKernelGGL(densify_noise_kernel, dim3(gs_div_up(rows, DN_THREADS)), dim3(DN_THREADS), 0, c->stream, seed, rows, out);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

}  // namespace gs
