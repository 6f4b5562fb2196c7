// binning.hip -- per-tile duplication + radix sort + tile ranges.
//
// Replaces count_tiles_per_gaussian / cumsum / generate_keys /
// radix_sort_tile_keys_fused_forward / compute_tile_ranges /
// compute_tile_counts_from_ranges / build_packed_tile_indices
// (slang/gaussian_tile_global_kernels.slang:17-404, GaussianRenderer.swift:333-490).
//
// Required output order (tile, depth bits, Gaussian index) -- the reference gets
// it from a stable LSD sort of (tile, depth) keys emitted in Gaussian order, run
// in ONE 128-thread threadgroup over all M pairs.  Here the same total order is
// produced with far fewer bytes:
//   1. stable LSD sort of the N (depth bits, index) records      -- 4 x 8-bit passes over N
//   2. exclusive scan of tiles-touched in that order              -- offsets, M (kept on device)
//   3. expansion to (tile id, index) pairs in that order          -- M pairs, 8 B each
//   4. stable LSD sort of the pairs by tile id only               -- ceil(tileBits/8) passes over M
//   5. boundary detection -> tile ranges
// Stability of step 4 keeps each tile's list in (depth bits, index) order.
// Everything is integer work and bit-identical to the reference order.
// No host synchronisation: M lives in ctx->counters[GS_CNT_M]; kernels read it
// there and run over grids sized from the reserved capacity.
#include "gs_ctx.h"
#include "gs_rider.h"
#include "gs_bwd_prep.h"

#ifdef GS_PROBE   // experiment builds only (python -m gaussiansplattingmlx_amd.build --variant probe -DGS_PROBE; tools/probe_read.py):
                  // thread 0 of a workgroup stamps the 100-MHz clock at the phases of a kernel
__device__ unsigned long long g_probe[1 << 16];
#define GS_PROBE_MARK(slot, k) do { if (threadIdx.x == 0 && (size_t)(slot) * 16 + (k) < (1u << 16)) g_probe[(size_t)(slot) * 16 + (k)] = wall_clock64(); } while (0)
extern "C" __attribute__((visibility("default"))) int gs_debug_probe_read(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe), (size_t)n * 8);
}
#define GS_PROBE_VAL(slot, k, v) do { if (threadIdx.x == 0 && (size_t)(slot) * 16 + (k) < (1u << 16)) g_probe[(size_t)(slot) * 16 + (k)] = (v); } while (0)
#else
#define GS_PROBE_MARK(slot, k) do { } while (0)
#define GS_PROBE_VAL(slot, k, v) do { } while (0)
#endif

namespace gs {

// ---------------------------------------------------------------------------------------------
// op-level prep: float rects -> tile rect, tiles touched, depth key  (count_tiles_per_gaussian :17-58)
// ---------------------------------------------------------------------------------------------
__global__ void bin_prep_kernel(int N, int tileW, int tileH, int gridW, int gridH, const float* __restrict__ rectMin,
                                const float* __restrict__ rectMax, const float* __restrict__ radii,
                                const float* __restrict__ depths, ushort4* __restrict__ tileRect,
                                uint32_t* __restrict__ tilesTouched, uint32_t* __restrict__ depthKey,
                                uint32_t* __restrict__ depthVal, uint32_t* __restrict__ visPerBlock,
                                uint32_t* __restrict__ counters, int noKeyForUntouched)
{
    if (blockIdx.x == 0 && threadIdx.x < GS_CNT_COUNT) counters[threadIdx.x] = 0;   // first kernel of the binning
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool visible = false;
    if (i < N) {
        uint32_t touched = 0;
        ushort4 tr = make_ushort4(0, 0, 0, 0);
        if (radii[i] > 0.0f) {
            int x0, y0, x1, y1;
            tile_rect(rectMin[2 * i], rectMin[2 * i + 1], rectMax[2 * i], rectMax[2 * i + 1], tileW, tileH, gridW,
                      gridH, x0, y0, x1, y1);
            touched = (uint32_t)((x1 - x0) * (y1 - y0));
            tr = make_ushort4((unsigned short)x0, (unsigned short)y0, (unsigned short)x1, (unsigned short)y1);
            visible = true;
        }
        tileRect[i] = tr;
        tilesTouched[i] = touched;
        // a Gaussian that touches no tile emits no pair: where the sort puts it does not matter, and keeping its key out
        // of the way lets the small depth sort skip the bytes the real keys share (GS_SORT_NO_KEY,
        // radix_hist_small_kernel).  Only there: with the big sort nothing is skipped, and on the 2 M-Gaussian garden
        // scene packing the Gaussians that do have pairs together doubled the load of the heaviest expansion waves
        // (binning 0.48 -> 0.71 ms)
        depthKey[i] = (touched || !noKeyForUntouched) ? __float_as_uint(depths[i]) : GS_SORT_NO_KEY;
        depthVal[i] = (uint32_t)i;
    }
    const int nvis = __syncthreads_count(visible);       // summed on demand (tile_counts_kernel), no atomics here
    if (threadIdx.x == 0) visPerBlock[blockIdx.x] = (uint32_t)nvis;
}

// ---------------------------------------------------------------------------------------------
// block-wide exclusive scan helper (256 threads)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive prefix of v over the block; *total receives the block sum. sm needs blockDim/64 + 1 words.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* sm, uint32_t* total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t incl = wave_incl_scan(v);
    if (lane == 63) sm[w] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int i = 0; i < nw; i++) {
        const uint32_t s = sm[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// ---------------------------------------------------------------------------------------------
// scan of tiles-touched in depth-sorted order
// ---------------------------------------------------------------------------------------------
// out[i] = sum of in[j * stride] over j < i, for i in [0, n]  (out[n] = the total).  One block.  The expansion and the
// compaction below sum the counts of the blocks before them themselves when there are few (<= GS_FUSED_SCAN_MAX: one
// launch less); that is quadratic in the block count, so large inputs (2 M Gaussians: 7813 blocks) go through this.
__global__ __launch_bounds__(1024) void prefix_u32_kernel(int n, const uint32_t* __restrict__ in, int stride,
                                                          unsigned long long* __restrict__ out)
{
    __shared__ uint32_t sm[20];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0ull;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < n ? in[(size_t)i * stride] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(v, sm, &tot);
        const unsigned long long c = carry;
        if (i < n) out[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry;
}

// the same for many elements (the sliced cut expansion of 2 M Gaussians has 250 k segments), in three launches: sums of
// 1024-element chunks, their prefix (prefix_u64_inplace_kernel), chunk-local scans on top of it
__global__ __launch_bounds__(1024) void chunk_sum_kernel(int n, const uint32_t* __restrict__ in, int stride,
                                                         unsigned long long* __restrict__ sums)
{
    __shared__ uint32_t sm[20];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const uint32_t v = i < n ? in[(size_t)i * stride] : 0u;
    uint32_t tot;
    block_excl_scan(v, sm, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void prefix_u64_inplace_kernel(int n, unsigned long long* __restrict__ a)
{   // exclusive prefix of a[0..n) in place, a[n] = total; one block, serial over 1024-element rounds
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned long long v = i < n ? a[i] : 0ull;
        unsigned long long incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long t = (unsigned long long)__shfl_up((long long)incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        unsigned long long wbase = 0ull, tot = 0ull;
        for (int k = 0; k < 16; k++) { const unsigned long long x = wsum[k]; if (k < w) wbase += x; tot += x; }
        const unsigned long long c = carry;
        if (i < n) a[i] = c + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) a[n] = carry;
}

__global__ __launch_bounds__(1024) void chunk_scan_kernel(int n, const uint32_t* __restrict__ in, int stride,
                                                          const unsigned long long* __restrict__ sums,
                                                          unsigned long long* __restrict__ out)
{
    __shared__ uint32_t sm[20];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const uint32_t v = i < n ? in[(size_t)i * stride] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan(v, sm, &tot);
    const unsigned long long base = sums[blockIdx.x];
    if (i < n) out[i] = base + ex;
    if (i == n - 1) out[n] = base + ex + v;
}

// out[0..n] = exclusive prefix of in[0..n) (stride in words), out[n] = total
static void launch_prefix(gs_ctx* c, int n, const uint32_t* in, int stride, unsigned long long* out)
{
    if (n <= 8192) {
        hipLaunchKernelGGL(prefix_u32_kernel, dim3(1), dim3(1024), 0, c->stream, n, in, stride, out);
        return;
    }
    const int nChunks = gs_div_up(n, 1024);
    hipLaunchKernelGGL(chunk_sum_kernel, dim3(nChunks), dim3(1024), 0, c->stream, n, in, stride, c->scanTmp);
    hipLaunchKernelGGL(prefix_u64_inplace_kernel, dim3(1), dim3(1024), 0, c->stream, nChunks, c->scanTmp);
    hipLaunchKernelGGL(chunk_scan_kernel, dim3(nChunks), dim3(1024), 0, c->stream, n, in, stride, c->scanTmp, out);
}

// Depth cuts (binning under a per-tile depth-key limit kept from the same view's previous forward; blend_v2.hip,
// bwd_items_kernel): a pair (Gaussian, tile) is binned only if the Gaussian's depth key does not exceed the tile's
// cut.  Lists are in key order, so what is binned is a prefix of the full list; a tile whose pixels have not all
// terminated on a cut list raises a flag and the caller repeats the forward without cuts (gs_forward_missed), so
// results never depend on the cuts.
__device__ __forceinline__ uint32_t cut_key(const uint32_t* __restrict__ cutStore, uint32_t tile)
{
    return 0xFFFFFFFFu - cutStore[tile];        // stored inverted: a zeroed buffer means "no cut"
}

// Coarse form of a view's depth cuts for the cut expansion: super-tile (sx, sy) = the tiles [4 sx, 4 sx + 4) x [4 sy, 4 sy + 4)
// holds the SMALLEST stored word of its tiles = the deepest cut among them (words are stored inverted; 0 = a tile without a
// cut, which then stands for the whole super-tile).  A Gaussian whose depth key lies beyond the super-cut of every super-tile
// its rect touches would lose every one of its candidate pairs one by one: expand_kernel<true> drops its rect unseen.
__global__ void cut_super_kernel(int gridW, int gridH, int sW, int sH, const uint32_t* __restrict__ cutStore,
                                 uint32_t* __restrict__ superCut)
{
    const int st = blockIdx.x * blockDim.x + threadIdx.x;
    if (st >= sW * sH) return;
    const int sy = st / sW, sx = st - sy * sW;
    uint32_t m = 0xFFFFFFFFu;
    for (int y = sy * GS_CUT_SUPER; y < min(gridH, (sy + 1) * GS_CUT_SUPER); y++)
        for (int x = sx * GS_CUT_SUPER; x < min(gridW, (sx + 1) * GS_CUT_SUPER); x++) m = min(m, cutStore[y * gridW + x]);
    superCut[st] = m;
}

__global__ __launch_bounds__(GS_SCAN_BLOCK) void scan_blocksum_kernel(int N, const uint32_t* __restrict__ sortedG,
                                                                      const uint32_t* __restrict__ tilesTouched,
                                                                      uint32_t* __restrict__ blockSums)
{
    __shared__ uint32_t sm[8];
    const int i = blockIdx.x * GS_SCAN_BLOCK + threadIdx.x;
    const uint32_t v = i < N ? tilesTouched[sortedG[i]] : 0u;
    uint32_t tot;
    block_excl_scan(v, sm, &tot);
    if (threadIdx.x == 0) blockSums[blockIdx.x] = tot;
}

// expansion (generate_keys :73-126) in depth-sorted order.  Each wave emits the pairs of its 64 Gaussians
// cooperatively: output position q of the wave belongs to the Gaussian whose exclusive offset is the largest one
// <= q (binary search over the wave's 64 offsets in LDS), so consecutive lanes write consecutive words whatever
// the splats' footprints are (a lane-per-Gaussian loop serialised on the largest footprint of the wave and wrote
// 64 unrelated addresses per instruction).
// idxBits > 0: one packed word (tile << idxBits | index) per pair; idxBits == 0: key = tile id, value = index.
// The scan over the block sums is done here by every block for itself (nb words, L2-resident) instead of by a
// single-block launch in between: sum of the blocks before this one = its output offset, sum of all = M.  Block 0
// publishes M and the capacity check; every block reaches the same verdict and leaves on overflow.  The tile ranges
// are cleared here too (empty tiles keep (0, 0); no memset launch).
// PIECES: the Gaussians' rects come as four row groups each (trimmed rects, gs_math.h rect_row_groups4: tilePieces[g] = first
// column | columns << 16 per group; group p = tile rows y0 + (h p >> 2) .. y0 + (h (p + 1) >> 2)); a Gaussian's positions run
// through group 0 row by row, then group 1 ...; tilesTouched is the four groups' tile count.
template <bool CUT, bool PIECES>
__global__ __launch_bounds__(GS_SCAN_BLOCK) void expand_kernel(int N, int gridW, int idxBits,
                                                               const uint32_t* __restrict__ sortedG,
                                                               const uint32_t* __restrict__ tilesTouched,
                                                               const ushort4* __restrict__ tileRect,
                                                               const uint32_t* __restrict__ blockSums,
                                                               uint32_t* __restrict__ counters,
                                                               unsigned long long capM,
                                                               uint32_t* __restrict__ tileRanges, int nRangeWords,
                                                               uint32_t* __restrict__ pairKey,
                                                               uint32_t* __restrict__ pairVal,
                                                               const uint32_t* __restrict__ sortedKey,
                                                               const uint32_t* __restrict__ cutStore,
                                                               uint2* __restrict__ waveSeg,
                                                               const unsigned long long* __restrict__ blockPrefix,
                                                               uint32_t* __restrict__ hostWords, uint32_t sliceMinPairs,
                                                               const uint32_t* __restrict__ superCut, int superW,
                                                               const uint4* __restrict__ tilePieces)
{
    __shared__ uint32_t sm[8];
    // a Gaussian's four groups: {positions in front of the group, first column | first tile row << 16, columns, 1 / columns}
    __shared__ uint4 sPg[PIECES ? GS_SCAN_BLOCK / 64 : 1][PIECES ? 64 : 1][4];
    __shared__ uint4 sPc[PIECES ? GS_SCAN_BLOCK / 64 : 1][PIECES ? 64 : 1];            // positions in front of groups 1, 2, 3
    __shared__ uint32_t sKey[GS_SCAN_BLOCK / 64][64];
    __shared__ unsigned long long sSum[GS_SCAN_BLOCK / 64][2];
    __shared__ uint32_t sOff[GS_SCAN_BLOCK / 64][64];    // exclusive offsets inside the wave
    __shared__ uint32_t sG[GS_SCAN_BLOCK / 64][64];
    __shared__ ushort4 sR[GS_SCAN_BLOCK / 64][64];
    // 1 / rect width per Gaussian: position -> (row, column) of the rect by one float multiply instead of an integer
    // division (~25 instructions of a kernel that is bound by instruction issue).  Exact for every position below 2^21:
    // (local + 0.5) / w is never closer than 0.5 / w to an integer, and the float product is off by less than
    // local 2^-22 / w (grids of 2^21 tiles or more -- images beyond 23 k x 23 k pixels -- take the division)
    __shared__ float sInv[GS_SCAN_BLOCK / 64][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // (PIECES) this lane's Gaussian: its row groups into the wave's LDS rows
    auto park_pieces = [&](uint32_t g, bool has, const ushort4 rr) {
        if (!PIECES) return;
        const uint4 pc = has ? tilePieces[g] : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t pw[4] = {pc.x, pc.y, pc.z, pc.w};
        const uint32_t h = (uint32_t)(rr.w - rr.y);
        uint32_t run = 0, c[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            c[p] = run;
            const uint32_t r0 = (h * (uint32_t)p) >> 2, r1 = (h * (uint32_t)(p + 1)) >> 2, cols = max(pw[p] >> 16, 1u);
            run += (pw[p] >> 16) * (r1 - r0);
            sPg[w][lane][p] = make_uint4(c[p], (pw[p] & 0xFFFFu) | (((uint32_t)rr.y + r0) << 16), cols,
                                         __float_as_uint(1.0f / (float)cols));
        }
        sPc[w][lane] = make_uint4(c[1], c[2], c[3], run);
    };
    // position `local` of the wave's Gaussian `lo` -> tile
    auto tile_of = [&](int lo, uint32_t local, bool fastDiv_) -> uint32_t {
        if (!PIECES) {
            const ushort4 r = sR[w][lo];
            const uint32_t rw = (uint32_t)(r.z - r.x);
            const uint32_t ty = fastDiv_ ? (uint32_t)(((float)local + 0.5f) * sInv[w][lo]) : local / rw;
            const uint32_t tx = local - ty * rw;
            return (r.y + ty) * (uint32_t)gridW + r.x + tx;
        }
        const uint4 pcum = sPc[w][lo];
        const uint32_t p = (local >= pcum.x ? 1u : 0u) + (local >= pcum.y ? 1u : 0u) + (local >= pcum.z ? 1u : 0u);
        const uint4 gr = sPg[w][lo][p];
        const uint32_t l2 = local - gr.x;
        const uint32_t ty = fastDiv_ ? (uint32_t)(((float)l2 + 0.5f) * __uint_as_float(gr.w)) : l2 / gr.z;
        const uint32_t tx = l2 - ty * gr.z;
        return ((gr.y >> 16) + ty) * (uint32_t)gridW + (gr.y & 0xFFFFu) + tx;
    };
    // gridDim.y > 1 (large inputs): slice y of every wave's positions goes to block (x, y) -- a wave whose 64 Gaussians
    // cover the whole screen otherwise walks 260 k positions alone while the rest of the chip has long finished
    // Only blocks with many positions are sliced (the block sums are known): the others would pay the gathers of
    // the prologue once per slice for nothing.
    const uint32_t slice = blockIdx.y;
    GS_PROBE_MARK(blockIdx.y == 0 && blockIdx.x < 3000u ? blockIdx.x : 4093u, 10);
    const bool fastDiv = nRangeWords < (1 << 22);        // 2 T words: fewer than 2^21 tiles
    const uint32_t nSlice = (gridDim.y > 1 && blockSums[blockIdx.x] >= sliceMinPairs) ? gridDim.y : 1u;
    if (slice >= nSlice) {
        if (CUT && threadIdx.x < GS_SCAN_BLOCK / 64)        // empty segments for the slices that do not exist
            waveSeg[(blockIdx.x * (GS_SCAN_BLOCK / 64) + threadIdx.x) * gridDim.y + slice] = make_uint2(0u, 0u);
        return;
    }
    if (slice == 0)
        for (int t = blockIdx.x * GS_SCAN_BLOCK + threadIdx.x; t < nRangeWords; t += gridDim.x * GS_SCAN_BLOCK) tileRanges[t] = 0;
    unsigned long long before = 0ull, total = 0ull;
    if (blockPrefix) { before = blockPrefix[blockIdx.x]; total = blockPrefix[gridDim.x]; }
    else {
        for (int b = threadIdx.x; b < (int)gridDim.x; b += GS_SCAN_BLOCK) {
            const uint32_t x = blockSums[b];
            total += x;
            if (b < (int)blockIdx.x) before += x;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            before += (unsigned long long)__shfl_xor((long long)before, d, 64);
            total += (unsigned long long)__shfl_xor((long long)total, d, 64);
        }
        if (lane == 0) { sSum[w][0] = before; sSum[w][1] = total; }
        __syncthreads();
        before = 0ull; total = 0ull;
#pragma unroll
        for (int k = 0; k < GS_SCAN_BLOCK / 64; k++) { before += sSum[k][0]; total += sSum[k][1]; }
    }
    if (blockIdx.x == 0 && slice == 0 && threadIdx.x == 0) {
        counters[GS_CNT_MREQ] = (uint32_t)(total > 0xFFFFFFFFull ? 0xFFFFFFFFull : total);
        if (total > capM) {
            counters[GS_CNT_OVERFLOW] = 1; counters[GS_CNT_M] = 0;
            // sticky words in host memory: the next API call that looks (no wait) reports the overflow (api.hip)
            hostWords[5] = counters[GS_CNT_MREQ]; hostWords[4] = 1u;
        } else {
            counters[GS_CNT_M] = (uint32_t)total;
            if (!CUT) hostWords[8] = (uint32_t)(total > 0xFFFFFFFFull ? 0xFFFFFFFFull : total);      // (the next forward's kernel choice: blend_v2.hip)
        }
    }
    if (total > capM) return;
    GS_PROBE_MARK(blockIdx.y == 0 && blockIdx.x < 3000u ? blockIdx.x : 4093u, 11);
    const int i = blockIdx.x * GS_SCAN_BLOCK + threadIdx.x;
    uint32_t g = 0, v = 0, area = 0;
    if (i < N) { g = sortedG[i]; area = tilesTouched[g]; v = area; }
    uint32_t tot;
    const uint32_t off = block_excl_scan(v, sm, &tot) + (uint32_t)before;
    const uint32_t waveBase = __shfl(off, 0, 64);
    if (CUT) {
        // Enumerate the wave's CANDIDATE pairs (full rects, Gaussians in rank order, tiles row-major) 64 at a time
        // and write the survivors in that order -- the order the uncut expansion has -- to the front of the region
        // the uncut pairs of the wave would occupy.  How many there are is known only now: compact_pairs_kernel
        // closes the gaps between the waves' segments (waveSeg = start, length).
        // (the region the wave's survivors are written to is sized by the FULL areas above; what is enumerated is the rects of
        // the Gaussians that are not beyond every super-cut of theirs: cut_super_kernel)
        const uint32_t myKey = i < N ? sortedKey[i] : 0u;
        ushort4 rr = area ? tileRect[g] : make_ushort4(0, 0, 1, 1);
        if (area && superCut) {
            bool reach = false;
            const int sx1 = (rr.z - 1) / GS_CUT_SUPER, sy1 = (rr.w - 1) / GS_CUT_SUPER;
            for (int sy = rr.y / GS_CUT_SUPER; sy <= sy1 && !reach; sy++)
                for (int sx = rr.x / GS_CUT_SUPER; sx <= sx1; sx++)
                    if (myKey <= 0xFFFFFFFFu - superCut[sy * superW + sx]) { reach = true; break; }
            if (!reach) { area = 0; rr = make_ushort4(0, 0, 1, 1); }
        }
        const uint32_t aIncl = wave_incl_scan(area);
        const uint32_t candTotal = __shfl(aIncl, 63, 64);
        sOff[w][lane] = aIncl - area;
        sG[w][lane] = g;
        sKey[w][lane] = myKey;
        sR[w][lane] = rr;
        sInv[w][lane] = 1.0f / (float)(rr.z - rr.x);
        park_pieces(g, area != 0, rr);
        const uint32_t per = ((candTotal + nSlice - 1) / nSlice + 63u) & ~63u;      // candidates per slice
        const uint32_t cBeg = min(candTotal, slice * per), cEnd = min(candTotal, (slice + 1) * per);
        uint32_t done = 0;
        const uint32_t myOffC = aIncl - area;
        // four 64-candidate groups per trip: the chains (LDS search, cut load) of the groups overlap -- a wave whose
        // Gaussians cover the whole screen (the nearest ones of a scene the camera stands in) walks thousands of groups
        for (uint32_t q0 = cBeg; q0 < cEnd; q0 += 256) {
            bool keep[4];
            uint32_t word[4], gg[4], tile[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t q = q0 + 64u * u + lane;
                keep[u] = false; word[u] = 0; gg[u] = 0; tile[u] = 0;
                // (owners of 64 consecutive candidates: a run of the wave's Gaussians, as in the uncut loop below)
                const uint32_t Qg = q0 + 64u * u;
                const int gLo = (int)__popcll(__ballot(myOffC <= Qg)) - 1;
                const int gHi = (int)__popcll(__ballot(myOffC < Qg + 64u)) - 1;
                int lo = gLo < 0 ? 0 : gLo;
                if (gHi - gLo <= 8) {
                    for (int k = gLo + 1; k <= gHi; k++) {
                        const uint32_t ok = (uint32_t)__builtin_amdgcn_readlane((int)myOffC, k);
                        lo = q >= ok ? k : lo;
                    }
                } else {
                    lo = 0;
#pragma unroll
                    for (int step = 32; step >= 1; step >>= 1)
                        if (lo + step < 64 && sOff[w][lo + step] <= q) lo += step;
                }
                if (q < cEnd) {
                    tile[u] = tile_of(lo, q - sOff[w][lo], fastDiv);
                    gg[u] = sG[w][lo];
                    keep[u] = sKey[w][lo] <= cut_key(cutStore, tile[u]);
                    word[u] = (tile[u] << idxBits) | gg[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const unsigned long long m = __ballot(keep[u]);
                if (keep[u]) {
                    const uint32_t pos = waveBase + cBeg + done + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (idxBits) pairKey[pos] = word[u];
                    else { pairKey[pos] = tile[u]; pairVal[pos] = gg[u]; }
                }
                done += (uint32_t)__popcll(m);
            }
        }
        // segment order = (wave, slice): a slice's survivors sit at the front of the slice's own stretch of the wave's region
        if (lane == 0) waveSeg[(blockIdx.x * (GS_SCAN_BLOCK / 64) + w) * gridDim.y + slice] = make_uint2(waveBase + cBeg, done);
        return;
    }
    const uint32_t waveTotal = __shfl(off + v, 63, 64) - waveBase;
    sOff[w][lane] = off - waveBase;
    sG[w][lane] = g;
    {
        const ushort4 rr = v ? tileRect[g] : make_ushort4(0, 0, 1, 1);
        sR[w][lane] = rr;
        sInv[w][lane] = 1.0f / (float)(rr.z - rr.x);
        park_pieces(g, v != 0, rr);
    }
    // wave-private LDS, DS operations of one wave complete in order: no barrier
    const uint32_t per = ((waveTotal + nSlice - 1) / nSlice + 63u) & ~63u;       // positions per slice, whole groups of 64
    const uint32_t qEnd = min(waveTotal, (slice + 1) * per);
    // The owner of position q is the largest j with offset_j <= q (zero-footprint entries share offsets: the LAST of equal
    // offsets owns the position).  The 64 positions of a trip are consecutive and the offsets do not decrease, so the owners
    // of a trip are a run [gLo, gHi] of the wave's Gaussians -- two ballots over the lanes' own offsets -- and a lane finds its
    // own among them with one compare per candidate (typically one to three: a Gaussian covers ~20 tiles).  Six dependent
    // LDS reads per position before (the binary search, kept for trips that span more than eight Gaussians).
    const uint32_t myOff = off - waveBase;
    GS_PROBE_MARK(blockIdx.y == 0 && blockIdx.x < 3000u ? blockIdx.x : 4093u, 12);
    for (uint32_t Q = slice * per; Q < qEnd; Q += 64) {
        const uint32_t q = Q + (uint32_t)lane;
        const int gLo = (int)__popcll(__ballot(myOff <= Q)) - 1;
        const int gHi = (int)__popcll(__ballot(myOff < Q + 64u)) - 1;
        int lo = gLo;
        if (gHi - gLo <= 8) {
            for (int k = gLo + 1; k <= gHi; k++) {
                const uint32_t ok = (uint32_t)__builtin_amdgcn_readlane((int)myOff, k);
                lo = q >= ok ? k : lo;
            }
        } else {
            lo = 0;
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1)
                if (lo + step < 64 && sOff[w][lo + step] <= q) lo += step;
        }
        if (q >= qEnd) continue;
        const uint32_t tile = tile_of(lo, q - sOff[w][lo], fastDiv);
        const uint32_t gg = sG[w][lo];
        if (idxBits) pairKey[waveBase + q] = (tile << idxBits) | gg;
        else { pairKey[waveBase + q] = tile; pairVal[waveBase + q] = gg; }
    }
    GS_PROBE_MARK(blockIdx.y == 0 && blockIdx.x < 3000u ? blockIdx.x : 4093u, 13);
}

// closes the gaps between the waves' segments of a cut expansion: segment w moves to the sum of the lengths before
// it.  Every block sums the lengths itself (4 per expansion block, L2-resident); block 0 publishes the pair count.
__global__ __launch_bounds__(GS_SCAN_BLOCK) void compact_pairs_kernel(int nSeg, const uint2* __restrict__ waveSeg,
                                                                      const uint32_t* __restrict__ keyIn,
                                                                      const uint32_t* __restrict__ valIn,
                                                                      uint32_t* __restrict__ keyOut,
                                                                      uint32_t* __restrict__ valOut,
                                                                      uint32_t* __restrict__ counters,
                                                                      uint32_t* __restrict__ hostWords,
                                                                      const unsigned long long* __restrict__ segPrefix,
                                                                      const uint32_t* __restrict__ dropPerBlock, int dropBlocks)
{
    __shared__ unsigned long long sSum[GS_SCAN_BLOCK / 64][2];
    __shared__ unsigned long long sDrop[GS_SCAN_BLOCK / 64];
    if (counters[GS_CNT_OVERFLOW]) return;
    // (block 0: the candidate pairs of the Gaussians the projection dropped whole under the view's super-cuts belong to what a
    // full binning would have made -- the count the caller's cut policy looks at)
    unsigned long long dropped = 0ull;
    if (blockIdx.x == 0 && dropBlocks > 0) {
        for (int k = threadIdx.x; k < dropBlocks; k += GS_SCAN_BLOCK) dropped += dropPerBlock[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) dropped += (unsigned long long)__shfl_xor((long long)dropped, d, 64);
        if ((threadIdx.x & 63) == 0) sDrop[threadIdx.x >> 6] = dropped;
        __syncthreads();
        dropped = 0ull;
#pragma unroll
        for (int k = 0; k < GS_SCAN_BLOCK / 64; k++) dropped += sDrop[k];
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int seg0 = blockIdx.x * (GS_SCAN_BLOCK / 64);
    unsigned long long before = 0ull, total = 0ull;
    if (segPrefix) { before = segPrefix[seg0 < nSeg ? seg0 : nSeg]; total = segPrefix[nSeg]; }
    else {
        for (int k = threadIdx.x; k < nSeg; k += GS_SCAN_BLOCK) {
            const uint32_t n = waveSeg[k].y;
            total += n;
            if (k < seg0) before += n;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            before += (unsigned long long)__shfl_xor((long long)before, d, 64);
            total += (unsigned long long)__shfl_xor((long long)total, d, 64);
        }
        if (lane == 0) { sSum[w][0] = before; sSum[w][1] = total; }
        __syncthreads();
        before = 0ull; total = 0ull;
#pragma unroll
        for (int k = 0; k < GS_SCAN_BLOCK / 64; k++) { before += sSum[k][0]; total += sSum[k][1]; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hostWords[1] = (uint32_t)total;             // pairs kept / pairs a full binning would have made: what the
        {                                           // caller's policy looks at (gs_cut_stats)
            const unsigned long long full = (unsigned long long)counters[GS_CNT_MREQ] + dropped;
            hostWords[2] = (uint32_t)(full > 0xFFFFFFFFull ? 0xFFFFFFFFull : full);
        }
        counters[GS_CNT_M] = (uint32_t)total;
        hostWords[8] = (uint32_t)total;             // (the next forward's kernel choice: blend_v2.hip)
    }
    uint32_t dst = (uint32_t)before;
    for (int k = 0; k < w; k++) if (seg0 + k < nSeg) dst += waveSeg[seg0 + k].y;
    if (seg0 + w >= nSeg) return;
    const uint2 sg = waveSeg[seg0 + w];
    for (uint32_t q0 = lane; q0 < sg.y; q0 += 256) {        // four loads in flight per lane: some segments are long
        uint32_t k[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t q = q0 + 64u * u;
            k[u] = q < sg.y ? keyIn[sg.x + q] : 0u;
            v[u] = (valIn && q < sg.y) ? valIn[sg.x + q] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t q = q0 + 64u * u;
            if (q < sg.y) { keyOut[dst + q] = k[u]; if (valIn) valOut[dst + q] = v[u]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// stable LSD radix sort, 8-bit digits, (u32 key, u32 value); n is read from device memory.
// Block b owns elements [b*TILE, (b+1)*TILE).  hist is digit-major: hist[d*nbCap + b].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GS_SORT_THREADS) void radix_hist_kernel(const uint32_t* __restrict__ keys,
                                                                     const uint32_t* __restrict__ nPtr, uint32_t nMax,
                                                                     int shift, int nbCap, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t h[256];
    uint32_t n = nPtr ? *nPtr : nMax;
    if (n > nMax) n = nMax;
    // the grid is sized from the CAPACITY (the count lives on the device): blocks walk the tiles that exist instead of
    // thousands of blocks starting up only to find their tile beyond n
    for (uint32_t tile = blockIdx.x; (unsigned long long)tile * GS_SORT_TILE < n; tile += gridDim.x) {
        const uint32_t base = tile * GS_SORT_TILE;
        h[threadIdx.x] = 0;
        __syncthreads();
        uint32_t k[GS_SORT_ITEMS];          // every load of the thread in flight before the first LDS atomic
#pragma unroll
        for (int r = 0; r < GS_SORT_ITEMS; r++) k[r] = keys[min(base + r * GS_SORT_THREADS + threadIdx.x, n - 1u)];
#pragma unroll
        for (int r = 0; r < GS_SORT_ITEMS; r++)
            if (base + r * GS_SORT_THREADS + threadIdx.x < n) atomicAdd(&h[(k[r] >> shift) & 255u], 1u);
        __syncthreads();
        hist[threadIdx.x * nbCap + tile] = h[threadIdx.x];
        __syncthreads();
    }
}

// block d scans row d over the active blocks; row total -> rowTotal[d]
__global__ __launch_bounds__(256) void radix_rowscan_kernel(const uint32_t* __restrict__ nPtr, uint32_t nMax, int nbCap,
                                                            uint32_t* __restrict__ hist, uint32_t* __restrict__ rowTotal)
{
    __shared__ uint32_t sm[8];
    uint32_t n = nPtr ? *nPtr : nMax;
    if (n > nMax) n = nMax;
    const int nb = (int)((n + GS_SORT_TILE - 1) / GS_SORT_TILE);
    uint32_t* row = hist + (size_t)blockIdx.x * nbCap;
    uint32_t carry = 0;
    for (int base = 0; base < nb; base += 256) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < nb ? row[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(v, sm, &tot);
        if (i < nb) row[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) rowTotal[blockIdx.x] = carry;
}

// ---- the depth sort of few tiles (N up to GS_SMALL_SORT_BLOCKS * 4096): two launches per pass instead of three --------
// AND / OR over the keys that matter (GS_SORT_NO_KEY marks Gaussians without a pair), per block by the first pass's
// histogram kernel; every later kernel folds the nb pairs itself (one wave-load each).
__device__ __forceinline__ uint2 reduce_key_bits(const uint2* __restrict__ blockBits, int nb)
{
    const int lane = threadIdx.x & 63;
    uint32_t a = 0xFFFFFFFFu, o = 0u;
    for (int i = lane; i < nb; i += 64) { const uint2 v = blockBits[i]; a &= v.x; o |= v.y; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a &= (uint32_t)__shfl_xor((int)a, d, 64); o |= (uint32_t)__shfl_xor((int)o, d, 64); }
    return make_uint2(a, o);
}
__device__ __forceinline__ bool sort_pass_needed(uint2 bits, int shift) { return (((bits.x ^ bits.y) >> shift) & 255u) != 0u; }

template <bool FIRST, int ITEMS>
__global__ __launch_bounds__(GS_SORT_THREADS) void radix_hist_small_kernel(const uint32_t* __restrict__ keys, uint32_t n, int shift,
                                                                           uint32_t* __restrict__ histB,
                                                                           uint2* __restrict__ blockBits)
{
    __shared__ uint32_t h[256];
    __shared__ uint32_t sAnd[GS_SORT_THREADS / 64], sOr[GS_SORT_THREADS / 64];
    if (!FIRST && !sort_pass_needed(reduce_key_bits(blockBits, (int)gridDim.x), shift)) return;   // the scatter only copies
    const uint32_t base = blockIdx.x * (GS_SORT_THREADS * ITEMS);
    h[threadIdx.x] = 0;
    __syncthreads();
    uint32_t a = 0xFFFFFFFFu, o = 0u;
#pragma unroll 4
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t i = base + r * GS_SORT_THREADS + threadIdx.x;
        if (i < n) {
            const uint32_t k = keys[i];
            atomicAdd(&h[(k >> shift) & 255u], 1u);
            if (FIRST && k != GS_SORT_NO_KEY) { a &= k; o |= k; }
        }
    }
    __syncthreads();
    histB[blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
    if (FIRST) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { a &= (uint32_t)__shfl_xor((int)a, d, 64); o |= (uint32_t)__shfl_xor((int)o, d, 64); }
        if ((threadIdx.x & 63) == 0) { sAnd[threadIdx.x >> 6] = a; sOr[threadIdx.x >> 6] = o; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < GS_SORT_THREADS / 64; k++) { a &= sAnd[k]; o |= sOr[k]; }
            blockBits[blockIdx.x] = make_uint2(a, o);
        }
    }
}

// Stable scatter.  The block ranks its 4096 elements into LDS in digit order (stable: element order = wave, round,
// lane, and wave w owns the contiguous elements [1024 w, 1024 w + 1024)), then streams LDS out linearly: elements of
// one digit leave as one contiguous run, so the global writes are 64-B-plus segments instead of 256 scattered
// dwords per round.  Ranking is wave-private -- per round every lane ORs its bit into the wave's 256-entry match
// table and reads the entry back, the group leader advances the wave's running digit count -- so the sixteen
// rounds need no workgroup barrier (the barrier-per-round version spent >50 % of its wave-cycles waiting).
// HAS_VALS = false (the tile passes over packed pair words) drops the value staging buffer: 29 KB of LDS per block
// instead of 45, five resident blocks per CU instead of three.
// SMALL (the depth sort of up to GS_SMALL_SORT_BLOCKS tiles): no row-scan launch -- the histograms are block-major
// (histB[b][digit], written by radix_hist_small_kernel) and thread d sums column d over the blocks itself, coalesced
// 1-KB rows, nb of them; and a pass whose byte is the same in every real key (blockBits) only copies its tile.
// THREADS: 256, or 1024 with a quarter of the elements per thread -- the same tile on sixteen waves ranks in ITEMS / 4 rounds (the
// rounds of a wave are a dependent chain through its LDS tables; see wide_scatter_kernel).  For the passes whose tiles are
// about as many as the CUs: the LSD depth sort above 655 k records.
template <bool HAS_VALS, bool SMALL, int ITEMS, int THREADS = GS_SORT_THREADS>
__global__ __launch_bounds__(THREADS) void radix_scatter_kernel(
    const uint32_t* __restrict__ keysIn, const uint32_t* __restrict__ valsIn, uint32_t* __restrict__ keysOut,
    uint32_t* __restrict__ valsOut, const uint32_t* __restrict__ nPtr, uint32_t nMax, int shift, int nbCap,
    const uint32_t* __restrict__ hist, const uint32_t* __restrict__ rowTotal, const uint2* __restrict__ blockBits,
    int firstPass, uint32_t* __restrict__ oob)
{
    __shared__ uint32_t digitBase[256];            // global destination of this block's first element of digit d
    __shared__ uint32_t blockStart[256];           // LDS position of this block's first element of digit d
    constexpr int NW = THREADS / 64;
    __shared__ uint32_t waveRun[NW][256];          // per wave: running count of digit d, then its offset in the block
    // the match tables (per wave: lanes holding digit d in the current round) are dead once the ranks are known and
    // the reorder buffer is not live before: they share memory (8 KB less LDS per block, more resident blocks)
    constexpr int TILE = THREADS * ITEMS;
    __shared__ __attribute__((aligned(16))) uint32_t keyS[TILE < NW * 512 ? NW * 512 : TILE];     // at least the match tables
    unsigned long long (*match)[256] = reinterpret_cast<unsigned long long (*)[256]>(keyS);
    static_assert(sizeof(unsigned long long) * NW * 256 <= sizeof(keyS), "match tables must fit in keyS");
    __shared__ uint32_t valS[HAS_VALS ? TILE : 1];
    __shared__ uint32_t sm[NW + 1];
    uint32_t n = nPtr ? *nPtr : nMax;
    if (n > nMax) n = nMax;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int PER_WAVE = TILE / NW;
    if (SMALL && !firstPass && !sort_pass_needed(reduce_key_bits(blockBits, (int)gridDim.x), shift)) {
        // every real key has the same digit here: the pass is the identity on their order (keys without a pair may
        // land anywhere).  The buffers still swap, so the tile is copied.
        const uint32_t base = blockIdx.x * TILE;
        for (uint32_t i = base + tid; i < min(base + (uint32_t)TILE, n); i += THREADS) {
            keysOut[i] = keysIn[i];
            if (HAS_VALS) valsOut[i] = valsIn[i];
        }
        return;
    }
    for (uint32_t tile = blockIdx.x; (unsigned long long)tile * TILE < n; tile += gridDim.x) {   // see radix_hist_kernel
    const uint32_t base = tile * TILE;
    const uint32_t cnt = min((uint32_t)TILE, n - base);

#pragma unroll
    for (int k = 0; k < 4; k++) { (&waveRun[0][0])[tid + k * THREADS] = 0u; (&match[0][0])[tid + k * THREADS] = 0ull; }
    __syncthreads();

    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
    const unsigned long long myBit = 1ull << lane;
    // unconditional loads from clamped addresses (see wide_scatter_kernel): elements beyond cnt are never ranked
    const uint32_t lastIdx = base + cnt - 1u;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t i = base + w * PER_WAVE + r * 64 + lane;
        key[r] = keysIn[min(i, lastIdx)];
        val[r] = HAS_VALS ? valsIn[min(i, lastIdx)] : 0u;
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t i = w * PER_WAVE + r * 64 + lane;
        const bool valid = i < cnt;
        const uint32_t d = valid ? (key[r] >> shift) & 255u : 0u;
        if (valid) atomicOr(&match[w][d], myBit);
        const unsigned long long peers = valid ? reinterpret_cast<volatile unsigned long long*>(&match[w][0])[d] : 0ull;
        const uint32_t before = valid ? reinterpret_cast<volatile uint32_t*>(&waveRun[w][0])[d] : 0u;
        const uint32_t inRound = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        rank[r] = before + inRound;                  // rank among the wave's elements of digit d
        if (valid && inRound == 0) {                 // group leader, after every lane's reads (program order)
            match[w][d] = 0ull;
            waveRun[w][d] = before + (uint32_t)__popcll(peers);
        }
    }
    __syncthreads();
    {   // thread tid < 256 owns digit tid: block histogram, wave offsets, block and global bases
        const bool owner = tid < 256;
        uint32_t c[NW], cs = 0;
        if (owner) {
#pragma unroll
            for (int k = 0; k < NW; k++) { c[k] = waveRun[k][tid]; cs += c[k]; }
        }
        uint32_t tot;
        uint32_t ls = block_excl_scan(cs, sm, &tot);
        if (owner) {
            blockStart[tid] = ls;
#pragma unroll
            for (int k = 0; k < NW; k++) { waveRun[k][tid] = ls; ls += c[k]; }
        }
        if (SMALL) {
            uint32_t before = 0, total = 0;
            const int nb = (int)gridDim.x;
            if (owner) {
#pragma unroll 8
                for (int b = 0; b < nb; b++) {
                    const uint32_t x = hist[b * 256 + tid];
                    total += x;
                    before += b < (int)tile ? x : 0u;
                }
            }
            const uint32_t gs = block_excl_scan(total, sm, &tot);
            if (owner) digitBase[tid] = gs + before;
        } else {
            const uint32_t gs = block_excl_scan(owner ? rowTotal[tid] : 0u, sm, &tot);
            if (owner) digitBase[tid] = gs + hist[tid * nbCap + tile];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t i = w * PER_WAVE + r * 64 + lane;
        if (i < cnt) {
            const uint32_t d = (key[r] >> shift) & 255u;
            const uint32_t pos = waveRun[w][d] + rank[r];
            keyS[pos] = key[r];
            if (HAS_VALS) valS[pos] = val[r];
        }
    }
    __syncthreads();
    for (uint32_t p = tid; p < cnt; p += THREADS) {
        const uint32_t k = keyS[p];
        const uint32_t d = (k >> shift) & 255u;
        const uint32_t dst = digitBase[d] + (p - blockStart[d]);
        // a destination is a rank among nMax elements: one beyond the buffers can only come from a corrupted rank or
        // histogram, and must show up as a failed parity test (and the overflow word), not as a store out of bounds
        if (dst >= nMax) { if (oob) *oob = 1u; continue; }
        keysOut[dst] = k;
        if (HAS_VALS) valsOut[dst] = valS[p];
    }
    __syncthreads();      // keyS doubles as the next tile's match tables
    }
}

// ---- the depth sort of at most GS_TINY_SORT_MAX records: ONE workgroup, one launch ---------------------------
// 10 k Gaussians are three sort tiles: the eight launches of the passes above were eight kernel boundaries with three
// busy CUs each (58 us of the 10 k / 400x400 forward's 294).  Here sixteen waves keep the records in registers (wave w
// owns the contiguous records [w T/16, (w+1) T/16), as in radix_scatter_kernel), rank them per pass with the same
// wave-private match tables, exchange them through one LDS copy and read them back in place; bytes that are the same
// in every real key are skipped (their pass is the identity on a stable sort).  Same order as the multi-launch path,
// bit for bit (tests/test_gpu_parity.py::test_tile_bin_bit_exact runs N = 3000 and 6000 through it).
constexpr int GS_TINY_THREADS = 1024, GS_TINY_ITEMS = 16, GS_TINY_SORT_MAX = GS_TINY_THREADS * GS_TINY_ITEMS;
__global__ __launch_bounds__(GS_TINY_THREADS) void radix_sort_tiny_kernel(const uint32_t* __restrict__ keysIn,
                                                                           const uint32_t* __restrict__ valsIn,
                                                                           uint32_t* __restrict__ keysOut,
                                                                           uint32_t* __restrict__ valsOut, uint32_t n)
{
    constexpr int NW = GS_TINY_THREADS / 64, PER_WAVE = GS_TINY_SORT_MAX / NW;        // 16 waves x 1024 records
    __shared__ __attribute__((aligned(16))) uint32_t keyS[GS_TINY_SORT_MAX];          // doubles as the match tables
    __shared__ uint32_t valS[GS_TINY_SORT_MAX];
    __shared__ uint32_t waveRun[NW][256];          // per wave: running count of digit d, then its first position
    __shared__ uint32_t sm[4];
    __shared__ uint32_t sBits[2];
    unsigned long long (*match)[256] = reinterpret_cast<unsigned long long (*)[256]>(keyS);
    static_assert(sizeof(unsigned long long) * NW * 256 <= sizeof(keyS), "match tables must fit in keyS");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long myBit = 1ull << lane;
    uint32_t key[GS_TINY_ITEMS], val[GS_TINY_ITEMS], rank[GS_TINY_ITEMS];
    const uint32_t lastIdx = n - 1u;
    uint32_t a = 0xFFFFFFFFu, o = 0u;
#pragma unroll
    for (int r = 0; r < GS_TINY_ITEMS; r++) {      // unconditional loads from clamped addresses
        const uint32_t i = (uint32_t)(w * PER_WAVE + r * 64 + lane);
        key[r] = keysIn[min(i, lastIdx)];
        val[r] = valsIn[min(i, lastIdx)];
        if (i < n && key[r] != GS_SORT_NO_KEY) { a &= key[r]; o |= key[r]; }
    }
    if (tid == 0) { sBits[0] = 0xFFFFFFFFu; sBits[1] = 0u; }
    __syncthreads();
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a &= (uint32_t)__shfl_xor((int)a, d, 64); o |= (uint32_t)__shfl_xor((int)o, d, 64); }
    if (lane == 0) { atomicAnd(&sBits[0], a); atomicOr(&sBits[1], o); }
    __syncthreads();
    const uint32_t varying = sBits[0] ^ sBits[1];          // bits that differ among the keys that have pairs
    for (int shift = 0; shift < 32; shift += 8) {
        if (shift != 0 && ((varying >> shift) & 255u) == 0u) continue;       // block-uniform
        for (int d = tid; d < NW * 256; d += GS_TINY_THREADS) { (&waveRun[0][0])[d] = 0u; (&match[0][0])[d] = 0ull; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < GS_TINY_ITEMS; r++) {
            const uint32_t i = (uint32_t)(w * PER_WAVE + r * 64 + lane);
            const bool valid = i < n;
            const uint32_t d = valid ? (key[r] >> shift) & 255u : 0u;
            if (valid) atomicOr(&match[w][d], myBit);
            const unsigned long long peers = valid ? reinterpret_cast<volatile unsigned long long*>(&match[w][0])[d] : 0ull;
            const uint32_t before = valid ? reinterpret_cast<volatile uint32_t*>(&waveRun[w][0])[d] : 0u;
            const uint32_t inRound = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
            rank[r] = before + inRound;                  // rank among the wave's records of digit d
            if (valid && inRound == 0) {                 // group leader, after every lane's reads (program order)
                match[w][d] = 0ull;
                waveRun[w][d] = before + (uint32_t)__popcll(peers);
            }
        }
        __syncthreads();
        // thread d < 256 (the first four waves): first position of digit d, then of every wave's share of it
        uint32_t c[NW], tot = 0, incl = 0;
        if (tid < 256) {
#pragma unroll
            for (int k = 0; k < NW; k++) { c[k] = waveRun[k][tid]; tot += c[k]; }
            incl = wave_incl_scan(tot);
            if (lane == 63) sm[w] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t run = incl - tot;
            for (int k = 0; k < w; k++) run += sm[k];
#pragma unroll
            for (int k = 0; k < NW; k++) { waveRun[k][tid] = run; run += c[k]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < GS_TINY_ITEMS; r++) {
            const uint32_t i = (uint32_t)(w * PER_WAVE + r * 64 + lane);
            if (i < n) {
                const uint32_t d = (key[r] >> shift) & 255u;
                const uint32_t pos = waveRun[w][d] + rank[r];
                keyS[pos] = key[r];
                valS[pos] = val[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < GS_TINY_ITEMS; r++) {
            const uint32_t i = (uint32_t)(w * PER_WAVE + r * 64 + lane);
            if (i < n) { key[r] = keyS[i]; val[r] = valS[i]; }
        }
        __syncthreads();          // keyS doubles as the next pass's match tables
    }
#pragma unroll
    for (int r = 0; r < GS_TINY_ITEMS; r++) {
        const uint32_t i = (uint32_t)(w * PER_WAVE + r * 64 + lane);
        if (i < n) { keysOut[i] = key[r]; valsOut[i] = val[r]; }
    }
}

// ---- the depth sort of at most GS_TINY_SORT_MAX records by RANK (round 4): one launch on the whole chip --------------------
// The position of record i in the stable sort is the number of records j with key_j < key_i, or key_j == key_i and j < i.
// N^2 compares are nothing at these sizes (10 k records: 2 x 10^8 / 64 wave-instructions = 5 us of the chip's issue slots)
// and, unlike the passes of radix_sort_tiny_kernel (26 us on ONE CU, each pass a chain of LDS round trips), they spread:
// workgroup b owns the 64 records [64 b, 64 b + 64), keeps ALL keys in LDS, and its sixteen waves count over a sixteenth of
// the keys each -- the keys come as broadcast 16-byte LDS reads, the tie-break only where j can be below i (the slice that
// holds the workgroup's own records; before it every key counts if <=, behind it if <) --, sum in LDS and store the 64
// records at their ranks.  Same order as the LSD passes, bit for bit (the binning tests run N = 1 ... 16384 through it).
constexpr int GS_RANK_THREADS = 1024;
__global__ __launch_bounds__(GS_RANK_THREADS) void rank_sort_kernel(const uint32_t* __restrict__ keysIn,
                                                                     const uint32_t* __restrict__ valsIn,
                                                                     uint32_t* __restrict__ keysOut,
                                                                     uint32_t* __restrict__ valsOut, uint32_t n)
{
    __shared__ __attribute__((aligned(16))) uint32_t keyS[GS_TINY_SORT_MAX];
    __shared__ uint32_t cnt[GS_RANK_THREADS / 64][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t n4 = (n + 3u) & ~3u;
    for (uint32_t j = (uint32_t)tid; j < n4; j += GS_RANK_THREADS) keyS[j] = j < n ? keysIn[j] : 0xFFFFFFFFu;   // (pads count for nobody)
    __syncthreads();
    const uint32_t i0 = blockIdx.x * 64u, i = i0 + (uint32_t)lane;
    const uint32_t ki = i < n ? keyS[i] : 0u;
    const uint32_t L = (((n4 + 15u) / 16u) + 3u) & ~3u;              // keys per wave, whole uint4
    // (the wave's range as scalars: the loops below branch and address on the scalar unit)
    const uint32_t jb = __builtin_amdgcn_readfirstlane(min((uint32_t)w * L, n4)), je = __builtin_amdgcn_readfirstlane(min(jb + L, n4));
    uint32_t c = 0;
    const uint4* k4 = reinterpret_cast<const uint4*>(keyS);
    // [jb, m0): every j below every i of the workgroup -- the key counts if <=;  [m0, m1): the workgroup's own records -- the
    // tie-break by index;  [m1, je): every j above -- the key counts if <
    const uint32_t m0 = min(max(i0 & ~3u, jb), je), m1 = min(max((i0 + 64u + 3u) & ~3u, jb), je);
    uint32_t j = jb;
    for (; j < m0; j += 4) {
        const uint4 k = k4[j >> 2];
        c += (uint32_t)(k.x <= ki); c += (uint32_t)(k.y <= ki); c += (uint32_t)(k.z <= ki); c += (uint32_t)(k.w <= ki);
    }
    for (; j < m1; j += 4) {
        const uint4 k = k4[j >> 2];
        c += (uint32_t)(k.x < ki || (k.x == ki && j < i)); c += (uint32_t)(k.y < ki || (k.y == ki && j + 1u < i));
        c += (uint32_t)(k.z < ki || (k.z == ki && j + 2u < i)); c += (uint32_t)(k.w < ki || (k.w == ki && j + 3u < i));
    }
    for (; j < je; j += 4) {
        const uint4 k = k4[j >> 2];
        c += (uint32_t)(k.x < ki); c += (uint32_t)(k.y < ki); c += (uint32_t)(k.z < ki); c += (uint32_t)(k.w < ki);
    }
    cnt[w][lane] = c;
    __syncthreads();
    if (w == 0 && i < n) {
        uint32_t r = 0;
#pragma unroll
        for (int q = 0; q < GS_RANK_THREADS / 64; q++) r += cnt[q][lane];
        keysOut[r] = ki;
        valsOut[r] = valsIn[i];
    }
}

// ---- the depth sort of 16385 .. 655 k records, round 3: splitter buckets + one local sort per bucket ------------------
// Four LSD passes are eight dependent launches (74 us for 322 k keys: every launch a chain of ranking rounds on a few
// dozen CUs).  Here: three.  127 SPLITTERS (keys of the previous depth sort's result at the ranks j N / 128) cut the key
// range into 255 buckets -- the 128 open intervals between consecutive splitters (even ids) and the 127 "equal to
// splitter j" classes (odd ids: whatever their size they need no sorting) --;
//   ss_hist_kernel      bucket of every record (binary search over the splitters in LDS), histogram per sort tile
//   ss_scatter_kernel   the stable scatter by bucket (radix_scatter_kernel's ranking), + the first record of every bucket
//   bucket_sort_kernel  one workgroup per bucket: a stable LSD sort of its records by their full key in registers and
//                       LDS (bytes that are the same in the whole bucket skipped; up to 8192 records, beyond that the
//                       same passes streamed through global memory by the one workgroup), and the splitters for the
//                       NEXT sort, read off the sorted result
// The bucket function is monotone in the key whatever the splitters are, and both steps are stable: the result is the
// order of the stable 32-bit LSD sort bit for bit for ANY splitters -- they only decide how even the buckets are.  (A
// digit cut out of the key's bits does not work for float depths: the highest bit in which two depths differ is an
// exponent bit, and one octave's bucket then holds most of the scene -- measured 0.20 -> 0.56 ms)
// A context's first depth sort has no splitters yet and takes the LSD passes, followed by ss_refresh_kernel.
// NS splitters make NB = 2 (NS + 1) buckets: 127 / 256 up to GS_SMALL_SORT_BLOCKS sort tiles (655 k records), 255 / 512 above
// (round 4: a bucket should hold a few thousand records -- at 1 M records 256 buckets of 7.8 k each overran the local sort's
// register path, at 2 M they would hold 15.6 k).
template <int NS>
__device__ __forceinline__ uint32_t ss_bucket_of(const uint32_t* __restrict__ sp /*LDS, [NS + 1], sp[NS] = 0xFFFFFFFF*/, uint32_t key)
{
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t step = (NS + 1) / 2; step >= 1; step >>= 1)
        if (sp[lo + step - 1] < key) lo += step;              // lo = number of splitters below the key, 0 .. NS
    return 2u * lo + ((lo < (uint32_t)NS && sp[lo] == key) ? 1u : 0u);
}

template <int NS> struct SsBucketId { typedef unsigned char type; };
template <> struct SsBucketId<255> { typedef unsigned short type; };

template <int ITEMS, int NS>
__global__ __launch_bounds__(GS_SORT_THREADS) void ss_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                                  const uint32_t* __restrict__ splitters,
                                                                  typename SsBucketId<NS>::type* __restrict__ bucketId,
                                                                  uint32_t* __restrict__ histB, ColourRider rider, int ownBlocks)
{
    if ((int)blockIdx.x >= ownBlocks) { colour_rider_block(rider, (int)blockIdx.x - ownBlocks); return; }   // gs_rider.h
    constexpr int NB = 2 * (NS + 1);
    __shared__ uint32_t h[NB];
    __shared__ uint32_t sp[NS + 1];
    GS_PROBE_MARK(3072 + blockIdx.x, 0);
    for (int i = threadIdx.x; i <= NS; i += GS_SORT_THREADS) sp[i] = i < NS ? splitters[i] : 0xFFFFFFFFu;
    for (int i = threadIdx.x; i < NB; i += GS_SORT_THREADS) h[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (GS_SORT_THREADS * ITEMS);
    // every key of the thread first, then their searches side by side: a block has its CU to itself (74 blocks at 300 k
    // records), so a thread's ITEMS binary searches -- seven dependent LDS reads each -- only overlap if the code lets them
    // (tools/probe_read.py: 9.0 us with four in flight at a time)
    uint32_t k[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t i = base + r * GS_SORT_THREADS + threadIdx.x;
        k[r] = keys[min(i, n - 1u)];
    }
    uint32_t lo[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) lo[r] = 0;
#pragma unroll
    for (uint32_t step = (NS + 1) / 2; step >= 1; step >>= 1)
#pragma unroll
        for (int r = 0; r < ITEMS; r++)
            if (sp[lo[r] + step - 1] < k[r]) lo[r] += step;          // (ss_bucket_of, interleaved)
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t i = base + r * GS_SORT_THREADS + threadIdx.x;
        if (i < n) {
            const uint32_t b = 2u * lo[r] + ((lo[r] < (uint32_t)NS && sp[lo[r]] == k[r]) ? 1u : 0u);
            bucketId[i] = (typename SsBucketId<NS>::type)b;
            atomicAdd(&h[b], 1u);
        }
    }
    __syncthreads();
    GS_PROBE_MARK(3072 + blockIdx.x, 1);
    for (int i = threadIdx.x; i < NB; i += GS_SORT_THREADS) histB[blockIdx.x * NB + i] = h[i];
    GS_PROBE_MARK(3072 + blockIdx.x, 2);
}

// Many sort tiles (BIG, more than GS_SMALL_SORT_BLOCKS): every scatter block summing the histogram rows of ALL the blocks
// before it is quadratic in the tile count (245 rows at 1 M records, 489 at 2 M; round 3 measured binning 0.31 -> 0.52 ms
// with it).  One small launch in between: block c turns the rows of its GS_SS_CHUNK tiles into exclusive prefixes inside
// the chunk (in place) and writes the chunk's totals; a scatter block then sums the totals of the chunks before its own
// -- 1/16 of the rows -- and adds its row.
constexpr int GS_SS_CHUNK = 16;
template <int NB>
__global__ __launch_bounds__(256) void ss_colscan_kernel(int nb, uint32_t* __restrict__ histB, uint32_t* __restrict__ chunkTot)
{
    const int b0 = blockIdx.x * GS_SS_CHUNK;
    const int d = blockIdx.y * 256 + threadIdx.x;
    uint32_t v[GS_SS_CHUNK];
#pragma unroll
    for (int k = 0; k < GS_SS_CHUNK; k++) v[k] = b0 + k < nb ? histB[(size_t)(b0 + k) * NB + d] : 0u;
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < GS_SS_CHUNK; k++) {
        if (b0 + k < nb) histB[(size_t)(b0 + k) * NB + d] = run;
        run += v[k];
    }
    chunkTot[(size_t)blockIdx.x * NB + d] = run;
}

// radix_scatter_kernel<true, true, ITEMS> with the digit read from bucketId instead of cut out of the key
template <int ITEMS, int NS, bool BIG>
__global__ __launch_bounds__(GS_SORT_THREADS) void ss_scatter_kernel(
    const uint32_t* __restrict__ keysIn, const uint32_t* __restrict__ valsIn, const typename SsBucketId<NS>::type* __restrict__ bucketId,
    uint32_t* __restrict__ keysOut, uint32_t* __restrict__ valsOut, uint32_t n, const uint32_t* __restrict__ histB,
    const uint32_t* __restrict__ chunkTot, uint32_t* __restrict__ bucketStart, uint32_t* __restrict__ oob, ColourRider rider,
    int ownBlocks)
{
    if ((int)blockIdx.x >= ownBlocks) { colour_rider_block(rider, (int)blockIdx.x - ownBlocks); return; }   // gs_rider.h
    constexpr int NB = 2 * (NS + 1), DPT = NB / GS_SORT_THREADS;       // buckets, buckets owned per thread (1 or 2)
    constexpr int TILE = GS_SORT_THREADS * ITEMS, PER_WAVE = TILE / 4;
    typedef typename SsBucketId<NS>::type BId;
    __shared__ uint32_t digitBase[NB];
    __shared__ uint32_t blockStart[NB];
    __shared__ uint32_t waveRun[4][NB];
    __shared__ __attribute__((aligned(16))) uint32_t keyS[TILE < 2 * 4 * NB ? 2 * 4 * NB : TILE];     // at least the match tables
    unsigned long long (*match)[NB] = reinterpret_cast<unsigned long long (*)[NB]>(keyS);
    static_assert(sizeof(unsigned long long) * 4 * NB <= sizeof(keyS), "match tables must fit in keyS");
    __shared__ uint32_t valS[TILE];
    __shared__ BId digS[TILE];
    __shared__ uint32_t sm[8];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t tile = blockIdx.x, base = tile * TILE;
    if (base >= n) return;
    const uint32_t cnt = min((uint32_t)TILE, n - base);
    GS_PROBE_MARK(3328 + blockIdx.x, 0);
    for (int d = tid; d < NB; d += GS_SORT_THREADS) {
        waveRun[0][d] = 0; waveRun[1][d] = 0; waveRun[2][d] = 0; waveRun[3][d] = 0;
        match[0][d] = 0ull; match[1][d] = 0ull; match[2][d] = 0ull; match[3][d] = 0ull;
    }
    __syncthreads();
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS], dig[ITEMS];
    const unsigned long long myBit = 1ull << lane;
    const uint32_t lastIdx = base + cnt - 1u;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {      // unconditional loads from clamped addresses (see wide_scatter_kernel)
        const uint32_t i = base + w * PER_WAVE + r * 64 + lane;
        key[r] = keysIn[min(i, lastIdx)];
        val[r] = valsIn[min(i, lastIdx)];
        dig[r] = bucketId[min(i, lastIdx)];
    }
    // records of every bucket in all the sort tiles (total) and in the tiles before this one (before): rows of global
    // memory that are ready when the kernel starts -- summed HERE, under the latency of the key loads above and ahead of the
    // ranking chain, not between the ranking and the stores (12.3 -> 10.x us per block alone on its CU)
    uint32_t before[DPT], total[DPT];
#pragma unroll
    for (int j = 0; j < DPT; j++) { before[j] = 0; total[j] = 0; }
    if (BIG) {
        const int nc = (ownBlocks + GS_SS_CHUNK - 1) / GS_SS_CHUNK, myc = (int)tile / GS_SS_CHUNK;
#pragma unroll 4
        for (int cc = 0; cc < nc; cc++)
#pragma unroll
            for (int j = 0; j < DPT; j++) {
                const uint32_t x = chunkTot[(size_t)cc * NB + tid * DPT + j];
                total[j] += x;
                before[j] += cc < myc ? x : 0u;
            }
#pragma unroll
        for (int j = 0; j < DPT; j++) before[j] += histB[(size_t)tile * NB + tid * DPT + j];      // prefix inside the chunk
    } else {
        const int nb = ownBlocks;
#pragma unroll 16
        for (int b = 0; b < nb; b++)
#pragma unroll
            for (int j = 0; j < DPT; j++) {
                const uint32_t x = histB[(size_t)b * NB + tid * DPT + j];
                total[j] += x;
                before[j] += b < (int)tile ? x : 0u;
            }
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t i = w * PER_WAVE + r * 64 + lane;
        const bool valid = i < cnt;
        const uint32_t d = valid ? dig[r] : 0u;
        if (valid) atomicOr(&match[w][d], myBit);
        const unsigned long long peers = valid ? reinterpret_cast<volatile unsigned long long*>(&match[w][0])[d] : 0ull;
        const uint32_t before = valid ? reinterpret_cast<volatile uint32_t*>(&waveRun[w][0])[d] : 0u;
        const uint32_t inRound = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        rank[r] = before + inRound;
        if (valid && inRound == 0) {
            match[w][d] = 0ull;
            waveRun[w][d] = before + (uint32_t)__popcll(peers);
        }
    }
    __syncthreads();
    GS_PROBE_MARK(3328 + blockIdx.x, 1);
    {   // thread tid owns the DPT consecutive buckets from DPT tid on: block histogram, wave offsets, block and global bases
        uint32_t c[DPT][4], cs[DPT], sum = 0;
#pragma unroll
        for (int j = 0; j < DPT; j++) {
            const int d = tid * DPT + j;
            c[j][0] = waveRun[0][d]; c[j][1] = waveRun[1][d]; c[j][2] = waveRun[2][d]; c[j][3] = waveRun[3][d];
            cs[j] = c[j][0] + c[j][1] + c[j][2] + c[j][3];
            sum += cs[j];
        }
        uint32_t tot;
        uint32_t ls = block_excl_scan(sum, sm, &tot);
#pragma unroll
        for (int j = 0; j < DPT; j++) {
            const int d = tid * DPT + j;
            blockStart[d] = ls;
            waveRun[0][d] = ls; waveRun[1][d] = ls + c[j][0]; waveRun[2][d] = ls + c[j][0] + c[j][1];
            waveRun[3][d] = ls + c[j][0] + c[j][1] + c[j][2];
            ls += cs[j];
        }
        uint32_t tsum = 0;
#pragma unroll
        for (int j = 0; j < DPT; j++) tsum += total[j];
        uint32_t gs = block_excl_scan(tsum, sm, &tot);
#pragma unroll
        for (int j = 0; j < DPT; j++) {
            const int d = tid * DPT + j;
            digitBase[d] = gs + before[j];
            if (tile == 0) bucketStart[d] = gs;     // first record of every bucket, for the local sorts behind this pass
            gs += total[j];
        }
        if (tile == 0 && tid == GS_SORT_THREADS - 1) bucketStart[NB] = n;
    }
    __syncthreads();
    GS_PROBE_MARK(3328 + blockIdx.x, 2);
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t i = w * PER_WAVE + r * 64 + lane;
        if (i < cnt) {
            const uint32_t pos = waveRun[w][dig[r]] + rank[r];
            keyS[pos] = key[r]; valS[pos] = val[r]; digS[pos] = (BId)dig[r];
        }
    }
    __syncthreads();
    for (uint32_t p = tid; p < cnt; p += GS_SORT_THREADS) {
        const uint32_t d = digS[p];
        const uint32_t dst = digitBase[d] + (p - blockStart[d]);
        if (dst >= n) { *oob = 1u; continue; }      // see radix_scatter_kernel: never a store out of bounds
        keysOut[dst] = keyS[p];
        valsOut[dst] = valS[p];
    }
    GS_PROBE_MARK(3328 + blockIdx.x, 3);
}

// workgroup size of the local sort: 512 threads keep up to 8192 records of a bucket in registers and LDS (73 KB), 1024
// threads 16384 (144 KB; the 2 M-record sorts, whose buckets average 7.8 k)
constexpr int GS_BUCKET_ITEMS = 16;

// One stable LSD pass over the block's `n` records (wave w owns the contiguous records [w perWave, (w + 1) perWave), `rounds`
// = perWave / 64 register slots per lane): ranks with the wave-private match tables of radix_scatter_kernel, then the
// digit's first position and every wave's share of it.  Leaves in pos[] every record's position in block-sorted order,
// in waveRun[0][d] the first position of digit d and in digitCount (thread d < 256) the block's count of digit d.
template <int GS_BUCKET_THREADS>
__device__ __forceinline__ void bucket_rank_pass(const uint32_t (&key)[GS_BUCKET_ITEMS], uint32_t n, uint32_t perWave, int rounds,
                                                 int shift, unsigned long long (*match)[256], uint32_t (*waveRun)[256],
                                                 uint32_t* sm, uint32_t (&pos)[GS_BUCKET_ITEMS], uint32_t& digitCount)
{
    constexpr int GS_BUCKET_NW = GS_BUCKET_THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int d = tid; d < GS_BUCKET_NW * 256; d += GS_BUCKET_THREADS) { (&waveRun[0][0])[d] = 0u; (&match[0][0])[d] = 0ull; }
    __syncthreads();
    const unsigned long long myBit = 1ull << lane;
#pragma unroll
    for (int r = 0; r < GS_BUCKET_ITEMS; r++) {
        if (r < rounds) {                               // block-uniform
            const uint32_t i = (uint32_t)w * perWave + (uint32_t)(r * 64 + lane);
            const bool valid = i < n;
            const uint32_t d = valid ? (key[r] >> shift) & 255u : 0u;
            if (valid) atomicOr(&match[w][d], myBit);
            const unsigned long long peers = valid ? reinterpret_cast<volatile unsigned long long*>(&match[w][0])[d] : 0ull;
            const uint32_t before = valid ? reinterpret_cast<volatile uint32_t*>(&waveRun[w][0])[d] : 0u;
            const uint32_t inRound = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
            pos[r] = before + inRound;                  // rank among the wave's records of digit d
            if (valid && inRound == 0) {                // group leader, after every lane's reads (program order)
                match[w][d] = 0ull;
                waveRun[w][d] = before + (uint32_t)__popcll(peers);
            }
        }
    }
    __syncthreads();
    uint32_t c[GS_BUCKET_NW], tot = 0, incl = 0;
    if (tid < 256) {
#pragma unroll
        for (int k = 0; k < GS_BUCKET_NW; k++) { c[k] = waveRun[k][tid]; tot += c[k]; }
        incl = wave_incl_scan(tot);
        if (lane == 63) sm[w] = incl;
    }
    digitCount = tot;
    __syncthreads();
    if (tid < 256) {
        uint32_t run = incl - tot;
        for (int k = 0; k < w; k++) run += sm[k];
#pragma unroll
        for (int k = 0; k < GS_BUCKET_NW; k++) { waveRun[k][tid] = run; run += c[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < GS_BUCKET_ITEMS; r++) {
        if (r < rounds) {
            const uint32_t i = (uint32_t)w * perWave + (uint32_t)(r * 64 + lane);
            if (i < n) pos[r] += waveRun[w][(key[r] >> shift) & 255u];
        }
    }
}

// the splitters whose ranks fall into [s0, s0 + n), read from `sorted` (the bucket's records in order; LDS or global).
// Splitter j = the key at rank ceil(j N / (NS + 1)) - 1 of this sort's result.
// (One 64-bit division per splitter instead of two per record.)
template <int NS, class Sorted>
__device__ __forceinline__ void ss_emit_range(uint32_t* __restrict__ splitNext, uint32_t N, uint32_t s0, uint32_t n, Sorted sorted)
{
    for (uint32_t j = threadIdx.x + 1u; j <= (uint32_t)NS; j += blockDim.x) {
        const uint32_t r = (uint32_t)(((unsigned long long)j * N + (unsigned long long)NS) / (unsigned long long)(NS + 1)) - 1u;
        if (r >= s0 && r < s0 + n) splitNext[j - 1u] = sorted(r - s0);
    }
}

template <int GS_BUCKET_THREADS, int NS>
__global__ __launch_bounds__(GS_BUCKET_THREADS) void bucket_sort_kernel(uint32_t* __restrict__ keysA, uint32_t* __restrict__ valsA,
                                                                        uint32_t* __restrict__ keysB, uint32_t* __restrict__ valsB,
                                                                        const uint32_t* __restrict__ bucketStart,
                                                                        uint32_t* __restrict__ splitNext)
{
    constexpr int GS_BUCKET_NW = GS_BUCKET_THREADS / 64, GS_BUCKET_MAX = GS_BUCKET_THREADS * GS_BUCKET_ITEMS;
    __shared__ __attribute__((aligned(16))) uint32_t keyS[GS_BUCKET_MAX];              // doubles as the match tables
    __shared__ uint32_t valS[GS_BUCKET_MAX];
    __shared__ uint32_t waveRun[GS_BUCKET_NW][256];
    __shared__ uint32_t base[256];
    __shared__ uint32_t sm[GS_BUCKET_NW];
    __shared__ uint32_t sBits[2];
    unsigned long long (*match)[256] = reinterpret_cast<unsigned long long (*)[256]>(keyS);
    static_assert(sizeof(unsigned long long) * GS_BUCKET_NW * 256 <= sizeof(keyS), "match tables must fit in keyS");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t s0 = bucketStart[blockIdx.x], n = bucketStart[blockIdx.x + 1] - s0, N = bucketStart[2 * (NS + 1)];
    if (n == 0u) return;                                         // (block-uniform)
    GS_PROBE_MARK(3584 + blockIdx.x, 0);
    GS_PROBE_VAL(3584 + blockIdx.x, 8, n);
    const bool allEqual = (blockIdx.x & 1u) != 0u;               // an "equal to splitter" class: in place already
    uint32_t key[GS_BUCKET_ITEMS], val[GS_BUCKET_ITEMS], pos[GS_BUCKET_ITEMS];
    if (allEqual || n == 1u) {
        // every record of the bucket has the same key: the splitters whose ranks fall into it, without walking it (an
        // "equal" class can be most of the array: the Gaussians without a pair all carry the key 0xFFFFFFFF)
        const uint32_t k0 = keysA[s0];
        ss_emit_range<NS>(splitNext, N, s0, n, [=](uint32_t) { return k0; });
        return;
    }
    if (tid == 0) { sBits[0] = 0xFFFFFFFFu; sBits[1] = 0u; }
    __syncthreads();
    if (n <= (uint32_t)GS_BUCKET_MAX) {
        // the records in registers, spread evenly over the waves (a bucket is ~2500 records on the bench scene: five
        // rounds per wave, not sixteen in the first and none in the last)
        const uint32_t perWave = ((n + GS_BUCKET_NW - 1) / GS_BUCKET_NW + 63u) & ~63u;
        const int rounds = (int)(perWave >> 6);
        const uint32_t last = s0 + n - 1u;
        uint32_t a = 0xFFFFFFFFu, o = 0u;
#pragma unroll
        for (int r = 0; r < GS_BUCKET_ITEMS; r++) {
            const uint32_t i = (uint32_t)w * perWave + (uint32_t)(r * 64 + lane);
            key[r] = keysA[min(s0 + i, last)];      // unconditional loads from clamped addresses
            val[r] = valsA[min(s0 + i, last)];
            a &= key[r]; o |= key[r];               // (clamped duplicates are records of the bucket too)
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { a &= (uint32_t)__shfl_xor((int)a, d, 64); o |= (uint32_t)__shfl_xor((int)o, d, 64); }
        if (lane == 0) { atomicAnd(&sBits[0], a); atomicOr(&sBits[1], o); }
        __syncthreads();
        const uint32_t varying = sBits[0] ^ sBits[1];
        GS_PROBE_MARK(3584 + blockIdx.x, 1);
        GS_PROBE_VAL(3584 + blockIdx.x, 9, varying);
        bool inLds = false;       // keyS holds the bucket's keys in sorted order (after the first pass that runs)
        for (int shift = 0; shift < 32; shift += 8) {
            if (((varying >> shift) & 255u) == 0u) continue;       // block-uniform: the byte is the same in the whole bucket
            inLds = true;
            uint32_t dc;
            bucket_rank_pass<GS_BUCKET_THREADS>(key, n, perWave, rounds, shift, match, waveRun, sm, pos, dc);
            __syncthreads();      // the match tables (in keyS) are dead
#pragma unroll
            for (int r = 0; r < GS_BUCKET_ITEMS; r++) {
                const uint32_t i = (uint32_t)w * perWave + (uint32_t)(r * 64 + lane);
                if (r < rounds && i < n) { keyS[pos[r]] = key[r]; valS[pos[r]] = val[r]; }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < GS_BUCKET_ITEMS; r++) {
                const uint32_t i = (uint32_t)w * perWave + (uint32_t)(r * 64 + lane);
                if (r < rounds && i < n) { key[r] = keyS[i]; val[r] = valS[i]; }
            }
            __syncthreads();      // keyS doubles as the next pass's match tables
        }
        GS_PROBE_MARK(3584 + blockIdx.x, 2);
#pragma unroll
        for (int r = 0; r < GS_BUCKET_ITEMS; r++) {
            const uint32_t i = (uint32_t)w * perWave + (uint32_t)(r * 64 + lane);
            if (r < rounds && i < n) { keysA[s0 + i] = key[r]; valsA[s0 + i] = val[r]; }
        }
        GS_PROBE_MARK(3584 + blockIdx.x, 3);
        // (no pass ran: every key of the bucket is the same, sBits[0])
        const uint32_t kAll = sBits[0];
        ss_emit_range<NS>(splitNext, N, s0, n, [=](uint32_t i) { return inLds ? keyS[i] : kAll; });
        GS_PROBE_MARK(3584 + blockIdx.x, 4);
        return;
    }
    // A bucket beyond the register path (the splitters are stale, or thousands of records lie between two of them): the
    // same passes, streamed by this one workgroup through global memory in tiles of GS_BUCKET_MAX records, buffer A <->
    // buffer B (the bucket's own stretch of each).  Slow and rare; correct for any size.
    {
        uint32_t a = 0xFFFFFFFFu, o = 0u;
        for (uint32_t i = tid; i < n; i += GS_BUCKET_THREADS) { const uint32_t k = keysA[s0 + i]; a &= k; o |= k; }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { a &= (uint32_t)__shfl_xor((int)a, d, 64); o |= (uint32_t)__shfl_xor((int)o, d, 64); }
        if (lane == 0) { atomicAnd(&sBits[0], a); atomicOr(&sBits[1], o); }
        __syncthreads();
    }
    const uint32_t varying = sBits[0] ^ sBits[1];
    uint32_t* kin = keysA; uint32_t* vin = valsA; uint32_t* kout = keysB; uint32_t* vout = valsB;
    const uint32_t perWaveT = GS_BUCKET_MAX / GS_BUCKET_NW;
    for (int shift = 0; shift < 32; shift += 8) {
        if (((varying >> shift) & 255u) == 0u) continue;
        if (tid < 256) base[tid] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += GS_BUCKET_THREADS) atomicAdd(&base[(kin[s0 + i] >> shift) & 255u], 1u);
        __syncthreads();
        {   // exclusive scan of the 256 digit counts
            uint32_t v = 0, incl = 0;
            if (tid < 256) { v = base[tid]; incl = wave_incl_scan(v); if (lane == 63) sm[w] = incl; }
            __syncthreads();
            if (tid < 256) {
                uint32_t run = incl - v;
                for (int k = 0; k < w; k++) run += sm[k];
                base[tid] = run;
            }
            __syncthreads();
        }
        for (uint32_t t0 = 0; t0 < n; t0 += GS_BUCKET_MAX) {
            const uint32_t nt = min((uint32_t)GS_BUCKET_MAX, n - t0);
            const uint32_t lastT = s0 + t0 + nt - 1u;
#pragma unroll
            for (int r = 0; r < GS_BUCKET_ITEMS; r++) {
                const uint32_t i = (uint32_t)w * perWaveT + (uint32_t)(r * 64 + lane);
                key[r] = kin[min(s0 + t0 + i, lastT)];
                val[r] = vin[min(s0 + t0 + i, lastT)];
            }
            uint32_t dc;
            bucket_rank_pass<GS_BUCKET_THREADS>(key, nt, perWaveT, GS_BUCKET_ITEMS, shift, match, waveRun, sm, pos, dc);
            // pos = position in tile-sorted order; the tile's records of digit d start at waveRun[0][d] there and go to
            // base[d] onwards in the bucket
#pragma unroll
            for (int r = 0; r < GS_BUCKET_ITEMS; r++) {
                const uint32_t i = (uint32_t)w * perWaveT + (uint32_t)(r * 64 + lane);
                if (i < nt) {
                    const uint32_t d = (key[r] >> shift) & 255u;
                    const uint32_t dst = s0 + base[d] + (pos[r] - waveRun[0][d]);
                    kout[dst] = key[r]; vout[dst] = val[r];
                }
            }
            __syncthreads();
            if (tid < 256) base[tid] += dc;
            __syncthreads();
        }
        uint32_t* t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
        __syncthreads();
    }
    if (kin != keysA)       // an odd number of passes: the result lies in B's stretch
        for (uint32_t i = tid; i < n; i += GS_BUCKET_THREADS) { keysA[s0 + i] = kin[s0 + i]; valsA[s0 + i] = vin[s0 + i]; }
    ss_emit_range<NS>(splitNext, N, s0, n, [=](uint32_t i) { return kin[s0 + i]; });
}

// splitters for the next depth sort, read off a sorted key array (after the LSD passes of a context's first sort)
__global__ void ss_refresh_kernel(const uint32_t* __restrict__ sortedKeys, uint32_t N, uint32_t* __restrict__ splitNext, int NS)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x + 1u;       // splitter j = 1 .. NS at rank ceil(j N / (NS + 1))
    if (j > (uint32_t)NS) return;
    const uint32_t rank = (uint32_t)(((unsigned long long)j * N + (unsigned long long)NS) / (unsigned long long)(NS + 1)) - 1u;      // as ss_emit_range
    splitNext[j - 1u] = sortedKeys[min(rank, N - 1u)];
}

// true if the depth sort of N records is going to take the splitter buckets (radix_sort below): the kernels the colour
// riders travel with (projection.hip decides on it before the binning is queued)
// splitters / buckets of the depth sort of N records: 127 / 256 up to GS_SMALL_SORT_BLOCKS sort tiles, 255 / 512 above
static inline int ss_splitters_for(int N) { return gs_small_depth_sort((long long)N) ? 127 : 255; }
constexpr int GS_SPLIT_MAX_TILES = 1024;        // sort tiles (4 M records) up to which the splitter sort is taken (ssChunk's size)

static bool ss_fits(const gs_ctx* c, int N)
{
    // Above GS_SMALL_SORT_BLOCKS sort tiles (655 k records) the 512-bucket form below is bit-exact and SLOWER than the twelve
    // launches of the LSD passes it replaces (MI355X, round 4: 2 M records 184 us against 115 -- the local sort of 256
    // buckets of 7.8 k records by 1024-thread workgroups alone takes 132 us --; 1 M records: binning 0.395 against 0.361 ms;
    // and at these sizes the sort kernels fill the chip, so colour riders only stretch them: EXPERIMENTS.md).  It is kept
    // behind GS_TUNE_SPLITTER_DEPTH_SORT = 2 for the tests and for A/B runs; 1 (default) takes splitters up to 655 k.
    if (!gs_small_depth_sort((long long)N) && c->splitterSort < 2) return false;
    const int nbSmall = gs_div_up(N, GS_SORT_THREADS * GS_SMALL_SORT_ITEMS);
    const int NB = 2 * (ss_splitters_for(N) + 1);
    return N > GS_TINY_SORT_MAX && nbSmall <= GS_SPLIT_MAX_TILES && (long long)nbSmall * NB <= 256LL * c->nbCap && c->splitterSort;
}

bool depth_sort_takes_splitters(const gs_ctx* c, int N)
{
    return ss_fits(c, N) && c->haveSplitters && c->splitterNS == ss_splitters_for(N);
}

// A host kernel's share of the forward's outstanding colour units (gs_rider.h): `permille` of all of them, as rider
// workgroups of `threads` threads behind the kernel's own.
static ColourRider rider_take(gs_ctx* c, int slot, int threads, int* blocks, int waves = GS_RIDER_WAVES)
{
    ColourRider a = c->rider.args;
    a.units = 0; a.unit0 = 0; *blocks = 0;
    if (!c->rider.on || c->colourRiders != 1) return a;
    const int left = c->rider.total - c->rider.next;
    int want = (int)(((long long)c->rider.total * c->riderShare[slot] + 999) / 1000);
    if (want > left || left - want < c->rider.total / 50) want = left;      // no launch of its own for a 2 % remainder
    if (c->riderShare[slot] <= 0 || want <= 0) return a;
    a.unit0 = c->rider.next; a.units = want;
    c->rider.next += want;
    *blocks = rider_blocks(want, threads, waves);
    return a;
}

// sorts key[0] (and val[0] if hasVals) over key bits [bitLo, bitHi); *resultBuf = index (0/1) of the result buffers
static int radix_sort(gs_ctx* c, uint32_t* key[2], uint32_t* val[2], bool hasVals, const uint32_t* nPtr, uint32_t nMax,
                      int bitLo, int bitHi, int* resultBuf)
{
    int src = 0;
    const int nbAll = gs_div_up(nMax, GS_SORT_TILE);
    if (nbAll == 0) { *resultBuf = 0; return GS_OK; }
    // count known on the host and few tiles (the depth sort of the Gaussians): two launches per pass, constant bytes skipped
    const int nbSmall = gs_div_up(nMax, GS_SORT_THREADS * GS_SMALL_SORT_ITEMS);
    if (!nPtr && hasVals && bitLo == 0 && bitHi == 32 && nMax <= (uint32_t)GS_TINY_SORT_MAX) {
        if (c->rankSort)
            hipLaunchKernelGGL(rank_sort_kernel, dim3(gs_div_up(nMax, 64)), dim3(GS_RANK_THREADS), 0, c->stream, key[0], val[0], key[1], val[1], nMax);
        else
            hipLaunchKernelGGL(radix_sort_tiny_kernel, dim3(1), dim3(GS_TINY_THREADS), 0, c->stream, key[0], val[0], key[1], val[1], nMax);
        GS_HIP_CHECK(c, hipGetLastError());
        *resultBuf = 1;
        return GS_OK;
    }
    const bool depthSort = !nPtr && hasVals && bitLo == 0 && bitHi == 32;
    const bool depthSmall = depthSort && gs_small_depth_sort((long long)nMax) && nbSmall <= c->nbCap;
    if (depthSort && depth_sort_takes_splitters(c, (int)nMax)) {
        const uint32_t* split = c->sortSplit[c->splitCur];
        uint32_t* splitNext = c->sortSplit[c->splitCur ^ 1];
        const bool big = !gs_small_depth_sort((long long)nMax);
        int rb = 0;
        ColourRider ra = rider_take(c, GS_RIDE_SS_HIST, GS_SORT_THREADS, &rb);
        if (!big) {
            hipLaunchKernelGGL((ss_hist_kernel<GS_SMALL_SORT_ITEMS, 127>), dim3(nbSmall + rb), dim3(GS_SORT_THREADS), 0, c->stream, key[0], nMax, split,
                               reinterpret_cast<unsigned char*>(c->bucketId), c->hist, ra, nbSmall);
            ra = rider_take(c, GS_RIDE_SS_SCATTER, GS_SORT_THREADS, &rb);
            hipLaunchKernelGGL((ss_scatter_kernel<GS_SMALL_SORT_ITEMS, 127, false>), dim3(nbSmall + rb), dim3(GS_SORT_THREADS), 0, c->stream, key[0], val[0],
                               reinterpret_cast<const unsigned char*>(c->bucketId), key[1], val[1], nMax, c->hist, nullptr, c->bucketStart,
                               c->counters + GS_CNT_OVERFLOW, ra, nbSmall);
            hipLaunchKernelGGL((bucket_sort_kernel<512, 127>), dim3(255), dim3(512), 0, c->stream, key[1], val[1], key[0], val[0],
                               c->bucketStart, splitNext);
        } else {
            // more than GS_SMALL_SORT_BLOCKS sort tiles: 512 buckets, and a chunk scan of the histogram rows between the two
            // passes (four launches; the LSD passes of this size are twelve)
            hipLaunchKernelGGL((ss_hist_kernel<GS_SMALL_SORT_ITEMS, 255>), dim3(nbSmall + rb), dim3(GS_SORT_THREADS), 0, c->stream, key[0], nMax, split,
                               reinterpret_cast<unsigned short*>(c->bucketId), c->hist, ra, nbSmall);
            hipLaunchKernelGGL((ss_colscan_kernel<512>), dim3(gs_div_up(nbSmall, GS_SS_CHUNK), 2), dim3(256), 0, c->stream, nbSmall, c->hist, c->ssChunk);
            ra = rider_take(c, GS_RIDE_SS_SCATTER, GS_SORT_THREADS, &rb);
            hipLaunchKernelGGL((ss_scatter_kernel<GS_SMALL_SORT_ITEMS, 255, true>), dim3(nbSmall + rb), dim3(GS_SORT_THREADS), 0, c->stream, key[0], val[0],
                               reinterpret_cast<const unsigned short*>(c->bucketId), key[1], val[1], nMax, c->hist, c->ssChunk, c->bucketStart,
                               c->counters + GS_CNT_OVERFLOW, ra, nbSmall);
            if (nMax <= 1400000u)        // buckets of ~N / 256 records: 8192 hold them with room up to 1.4 M, 16384 beyond
                hipLaunchKernelGGL((bucket_sort_kernel<512, 255>), dim3(511), dim3(512), 0, c->stream, key[1], val[1], key[0], val[0],
                                   c->bucketStart, splitNext);
            else
                hipLaunchKernelGGL((bucket_sort_kernel<1024, 255>), dim3(511), dim3(1024), 0, c->stream, key[1], val[1], key[0], val[0],
                                   c->bucketStart, splitNext);
        }
        GS_HIP_CHECK(c, hipGetLastError());
        c->splitCur ^= 1;
        *resultBuf = 1;
        return GS_OK;
    }
    // no splitters yet (a context's first depth sort, or the record count has crossed the 127 / 255 boundary): the LSD
    // passes, then the splitters for the sorts behind it read off the sorted keys
    const bool wantSplitters = depthSort && ss_fits(c, (int)nMax);
    if (!nPtr && hasVals && gs_small_depth_sort((long long)nMax) && nbSmall <= c->nbCap) {
        for (int shift = bitLo; shift < bitHi; shift += 8) {
            const bool first = shift == bitLo;
            if (first)
                hipLaunchKernelGGL((radix_hist_small_kernel<true, GS_SMALL_SORT_ITEMS>), dim3(nbSmall), dim3(GS_SORT_THREADS), 0,
                                   c->stream, key[src], nMax, shift, c->hist, c->sortBits);
            else
                hipLaunchKernelGGL((radix_hist_small_kernel<false, GS_SMALL_SORT_ITEMS>), dim3(nbSmall), dim3(GS_SORT_THREADS), 0,
                                   c->stream, key[src], nMax, shift, c->hist, c->sortBits);
            hipLaunchKernelGGL((radix_scatter_kernel<true, true, GS_SMALL_SORT_ITEMS>), dim3(nbSmall), dim3(GS_SORT_THREADS), 0,
                               c->stream, key[src], val[src], key[src ^ 1], val[src ^ 1], nullptr, nMax, shift, c->nbCap, c->hist,
                               nullptr, c->sortBits, first ? 1 : 0, c->counters + GS_CNT_OVERFLOW);
            src ^= 1;
        }
        if (depthSmall && wantSplitters) {
            hipLaunchKernelGGL(ss_refresh_kernel, dim3(1), dim3(256), 0, c->stream, key[src], nMax, c->sortSplit[c->splitCur], 127);
            c->haveSplitters = true;
            c->splitterNS = 127;
        }
        GS_HIP_CHECK(c, hipGetLastError());
        *resultBuf = src;
        return GS_OK;
    }
    const int nb = nbAll < GS_SORT_MAX_GRID ? nbAll : GS_SORT_MAX_GRID;     // the kernels walk the tiles beyond the grid
    // sixteen waves per tile while there is at most one tile per CU (82 KB of LDS: one block per CU).  MI355X: 1 M records /
    // 245 tiles 12.7 -> 10.7 us per pass, 2 M / 489 tiles 15.3 -> 18.6
    const bool wideLsd = depthSort && c->lsdThreads != 256 && (c->lsdThreads == 1024 || nbAll <= c->numCUs);
    for (int shift = bitLo; shift < bitHi; shift += 8) {
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nb), dim3(GS_SORT_THREADS), 0, c->stream, key[src], nPtr, nMax,
                           shift, c->nbCap, c->hist);
        hipLaunchKernelGGL(radix_rowscan_kernel, dim3(256), dim3(256), 0, c->stream, nPtr, nMax, c->nbCap, c->hist,
                           c->rowTotal);
        if (hasVals && wideLsd)
            hipLaunchKernelGGL((radix_scatter_kernel<true, false, GS_SORT_ITEMS / 4, 1024>), dim3(nb), dim3(1024), 0, c->stream, key[src],
                               val[src], key[src ^ 1], val[src ^ 1], nPtr, nMax, shift, c->nbCap, c->hist, c->rowTotal,
                               nullptr, 0, c->counters + GS_CNT_OVERFLOW);
        else if (hasVals)
            hipLaunchKernelGGL((radix_scatter_kernel<true, false, GS_SORT_ITEMS>), dim3(nb), dim3(GS_SORT_THREADS), 0, c->stream, key[src],
                               val[src], key[src ^ 1], val[src ^ 1], nPtr, nMax, shift, c->nbCap, c->hist, c->rowTotal,
                               nullptr, 0, c->counters + GS_CNT_OVERFLOW);
        else
            hipLaunchKernelGGL((radix_scatter_kernel<false, false, GS_SORT_ITEMS>), dim3(nb), dim3(GS_SORT_THREADS), 0, c->stream, key[src],
                               nullptr, key[src ^ 1], nullptr, nPtr, nMax, shift, c->nbCap, c->hist, c->rowTotal, nullptr, 0,
                               c->counters + GS_CNT_OVERFLOW);
        src ^= 1;
    }
    if (wantSplitters) {
        hipLaunchKernelGGL(ss_refresh_kernel, dim3(1), dim3(256), 0, c->stream, key[src], nMax, c->sortSplit[c->splitCur], 255);
        c->haveSplitters = true;
        c->splitterNS = 255;
    }
    GS_HIP_CHECK(c, hipGetLastError());
    *resultBuf = src;
    return GS_OK;
}


// ---------------------------------------------------------------------------------------------
// Tile sort in ONE pass (T <= 4096 tiles: the whole tile id is one 12-bit digit).
//
// The two 8-bit LSD passes above move every pair twice and need seven launches (2 x histogram, row scan, scatter, then
// the range kernel).  Here: per sort tile (4096 pairs) a 4096-bin histogram (wide_hist_kernel, u16 counts); prefixes of
// the counts over the sort tiles in two levels (wide_chunk_kernel inside chunks of 16 tiles, wide_tile_kernel over the
// chunks, one thread per tile id, coalesced rows); then wide_scatter_kernel sorts its 4096 pairs by tile id LOCALLY in
// LDS (two stable steps with the match-table ranking of radix_scatter_kernel: low 8 bits, high 4 bits), finds the runs
// of equal tile ids and streams every run to tile start + pairs of that tile in earlier sort tiles.  Pairs move once;
// four launches; the tile ranges fall out of the per-tile totals (no range kernel).  Stable, so each tile's list stays
// in (depth bits, Gaussian index) order.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GS_SORT_THREADS) void wide_hist_kernel(const uint32_t* __restrict__ keys,
                                                                    const uint32_t* __restrict__ nPtr, uint32_t nMax, int shift,
                                                                    uint16_t* __restrict__ cnt)
{
    __shared__ uint32_t h[GS_WIDE_BINS];
    uint32_t n = *nPtr;
    if (n > nMax) n = nMax;
    for (uint32_t tile = blockIdx.x; (unsigned long long)tile * GS_SORT_TILE < n; tile += gridDim.x) {
        const uint32_t base = tile * GS_SORT_TILE;
        for (int i = threadIdx.x; i < GS_WIDE_BINS; i += GS_SORT_THREADS) h[i] = 0;
        __syncthreads();
        uint32_t k[GS_SORT_ITEMS];          // every load of the thread in flight before the first LDS atomic
#pragma unroll
        for (int r = 0; r < GS_SORT_ITEMS; r++) k[r] = keys[min(base + r * GS_SORT_THREADS + threadIdx.x, n - 1u)];
#pragma unroll
        for (int r = 0; r < GS_SORT_ITEMS; r++)
            if (base + r * GS_SORT_THREADS + threadIdx.x < n) atomicAdd(&h[(k[r] >> shift) & (GS_WIDE_BINS - 1)], 1u);
        __syncthreads();
        uint32_t* out = reinterpret_cast<uint32_t*>(cnt + (size_t)tile * GS_WIDE_BINS);     // a count is at most 4096
        for (int i = threadIdx.x; i < GS_WIDE_BINS / 2; i += GS_SORT_THREADS) out[i] = h[2 * i] | (h[2 * i + 1] << 16);
        __syncthreads();
    }
}

// block (c, y): chunk c of GS_WIDE_CHUNK sort tiles, tile ids [256 y, 256 y + 256): counts -> exclusive prefixes inside
// the chunk (in place, < 15 * 4096 < 2^16), chunk totals out
__global__ __launch_bounds__(256) void wide_chunk_kernel(const uint32_t* __restrict__ nPtr, uint32_t nMax, uint16_t* __restrict__ cnt,
                                                         uint32_t* __restrict__ chunkSum)
{
    uint32_t n = *nPtr;
    if (n > nMax) n = nMax;
    const uint32_t nb = (n + GS_SORT_TILE - 1) / GS_SORT_TILE;
    const uint32_t b0 = blockIdx.x * GS_WIDE_CHUNK;
    if (b0 >= nb) return;
    const uint32_t t = blockIdx.y * 256 + threadIdx.x;
    uint32_t v[GS_WIDE_CHUNK];
#pragma unroll
    for (int k = 0; k < GS_WIDE_CHUNK; k++) v[k] = b0 + k < nb ? cnt[(size_t)(b0 + k) * GS_WIDE_BINS + t] : 0u;
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < GS_WIDE_CHUNK; k++) {
        if (b0 + k < nb) cnt[(size_t)(b0 + k) * GS_WIDE_BINS + t] = (uint16_t)run;
        run += v[k];
    }
    chunkSum[(size_t)blockIdx.x * GS_WIDE_BINS + t] = run;
}

// one thread per tile id: chunk totals -> exclusive prefixes over the chunks (in place), pairs of the tile out
__global__ __launch_bounds__(256) void wide_tile_kernel(const uint32_t* __restrict__ nPtr, uint32_t nMax, uint32_t* __restrict__ chunkSum,
                                                        uint32_t* __restrict__ tileTotal, ColourRider rider, int ownBlocks)
{
    if ((int)blockIdx.x >= ownBlocks) { colour_rider_block(rider, (int)blockIdx.x - ownBlocks); return; }   // gs_rider.h
    uint32_t n = *nPtr;
    if (n > nMax) n = nMax;
    const uint32_t nb = (n + GS_SORT_TILE - 1) / GS_SORT_TILE;
    const uint32_t nChunks = (nb + GS_WIDE_CHUNK - 1) / GS_WIDE_CHUNK;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    uint32_t run = 0;
    uint32_t c = 0;
    for (; c + 32 <= nChunks; c += 32) {    // 32 independent loads in flight: sixteen workgroups walk the chunks alone, and a
        uint32_t v[32];                     // batch costs one memory latency however wide it is (eight: 2 M Gaussians 20 us)
#pragma unroll
        for (int k = 0; k < 32; k++) v[k] = chunkSum[(size_t)(c + k) * GS_WIDE_BINS + t];
#pragma unroll
        for (int k = 0; k < 32; k++) { chunkSum[(size_t)(c + k) * GS_WIDE_BINS + t] = run; run += v[k]; }
    }
    for (; c + 8 <= nChunks; c += 8) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = chunkSum[(size_t)(c + k) * GS_WIDE_BINS + t];
#pragma unroll
        for (int k = 0; k < 8; k++) { chunkSum[(size_t)(c + k) * GS_WIDE_BINS + t] = run; run += v[k]; }
    }
    for (; c < nChunks; c++) {
        const uint32_t v = chunkSum[(size_t)c * GS_WIDE_BINS + t];
        chunkSum[(size_t)c * GS_WIDE_BINS + t] = run;
        run += v;
    }
    tileTotal[t] = run;
}

// stable ranking of the block's elements by an (up to) 8-bit digit, as in radix_scatter_kernel: wave-private match
// tables, no workgroup barrier per round.  Returns the LDS position of every element in block-sorted order.
// BALLOT_BITS > 0: the digit has that few bits and whole waves share one value of it (the high bits of the tile id:
// 64 consecutive pairs lie in one or two rows of tiles) -- the LDS match table would take 64 same-address atomics per
// round; the lanes of equal digit are found with one ballot per bit instead.
template <int NW, int BALLOT_BITS, class DigitOf>
__device__ __forceinline__ void local_rank_pass(const uint32_t (&key)[GS_SORT_TILE / (NW * 64)], uint32_t cnt, DigitOf digit_of,
                                                unsigned long long (*match)[256], uint32_t (*waveRun)[256], uint32_t* sm,
                                                uint32_t (&pos)[GS_SORT_TILE / (NW * 64)])
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int PER_WAVE = GS_SORT_TILE / NW, ITEMS = PER_WAVE / 64;
#pragma unroll
    for (int k = 0; k < 4; k++) { (&waveRun[0][0])[tid + k * NW * 64] = 0u; (&match[0][0])[tid + k * NW * 64] = 0ull; }
    __syncthreads();
    const unsigned long long myBit = 1ull << lane;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t i = w * PER_WAVE + r * 64 + lane;
        const bool valid = i < cnt;
        const uint32_t d = valid ? digit_of(key[r]) : 0u;
        unsigned long long peers;
        if (BALLOT_BITS > 0) {
            peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < BALLOT_BITS; b++) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long m = __ballot(valid && bit);
                peers &= bit ? m : ~m;
            }
        } else {
            if (valid) atomicOr(&match[w][d], myBit);
            peers = valid ? reinterpret_cast<volatile unsigned long long*>(&match[w][0])[d] : 0ull;
        }
        const uint32_t before = valid ? reinterpret_cast<volatile uint32_t*>(&waveRun[w][0])[d] : 0u;
        const uint32_t inRound = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        pos[r] = before + inRound;
        if (valid && inRound == 0) {
            if (BALLOT_BITS == 0) match[w][d] = 0ull;
            waveRun[w][d] = before + (uint32_t)__popcll(peers);
        }
    }
    __syncthreads();
    {   // thread d < 256: first position of digit d, then of every wave's share of it
        uint32_t c[NW], sum = 0;
        if (tid < 256) {
#pragma unroll
            for (int k = 0; k < NW; k++) { c[k] = waveRun[k][tid]; sum += c[k]; }
        }
        uint32_t tot;
        uint32_t run = block_excl_scan(sum, sm, &tot);
        if (tid < 256) {
#pragma unroll
            for (int k = 0; k < NW; k++) { waveRun[k][tid] = run; run += c[k]; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t i = w * PER_WAVE + r * 64 + lane;
        if (i < cnt) pos[r] += waveRun[w][digit_of(key[r])];
    }
}

// THREADS: 256 (sixteen elements per thread) or 1024 (four).  The ranking rounds of a wave are a dependent chain through its
// LDS tables -- ~0.3 us each for a wave that has its SIMD to itself (tools/probe_read.py: 4.4 + 5.1 us for the two ranking steps of
// a 256-thread block, of its 20) --, so the same sort tile on sixteen waves ranks in four rounds instead of sixteen and
// gathers its runs' bases in one batch instead of four.
template <bool HAS_VALS, int THREADS>
__global__ __launch_bounds__(THREADS, HAS_VALS ? 1 : THREADS == 1024 ? 8 : THREADS == 512 ? 6 : 1) void wide_scatter_kernel(
    const uint32_t* __restrict__ keysIn, const uint32_t* __restrict__ valsIn, uint32_t* __restrict__ keysOut,
    uint32_t* __restrict__ valsOut, const uint32_t* __restrict__ nPtr, uint32_t nMax, int shift,
    const uint16_t* __restrict__ cnt, const uint32_t* __restrict__ chunkSum, const uint32_t* __restrict__ tileTotal,
    uint32_t* __restrict__ tileRanges, int T, SegBaseArgs seg, int withSeg, uint32_t* __restrict__ oob)
{
    constexpr int NW = THREADS / 64, ITEMS = GS_SORT_TILE / THREADS;
    __shared__ uint32_t waveRun[NW][256];
    // the match tables of the two ranking steps live in keyS while it holds nothing else (as in radix_scatter_kernel)
    __shared__ __attribute__((aligned(16))) uint32_t keyS[GS_SORT_TILE > NW * 512 ? GS_SORT_TILE : NW * 512];
    unsigned long long (*match)[256] = reinterpret_cast<unsigned long long (*)[256]>(keyS);
    __shared__ uint32_t valS[HAS_VALS ? GS_SORT_TILE : 1];
    // destination of LDS position 0 of the run of tile id D (so that position p of the run goes to baseS[D] + p)
    __shared__ uint32_t baseS[GS_WIDE_BINS];
    __shared__ uint32_t sm[NW + 1];
    GS_PROBE_MARK(blockIdx.x < 3000u ? blockIdx.x : 4094u, 0);
    uint32_t n = *nPtr;
    if (n > nMax) n = nMax;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int PER_WAVE = GS_SORT_TILE / NW;
    // Workgroups go to the eight XCDs round-robin, and each XCD has its own L2.  A tile's list is appended to by
    // consecutive sort tiles, a few pairs (one run) each: XCD x owns a CONTIGUOUS eighth of the sort tiles, so the runs
    // that share a 128-B line meet in one L2.
    // The first eight workgroups sort nothing (eight, so that the sort tiles keep their XCDs): block 0 publishes the tile
    // ranges and block 1, when the fused blend forward follows, does that kernel's bookkeeping from the per-tile totals
    // (seg_base_body: one launch less in front of the blend).
    if (blockIdx.x == 1 && withSeg) {
        seg_base_body<GS_SEG_LEN>(seg, baseS);
        GS_PROBE_MARK(blockIdx.x < 3000u ? blockIdx.x : 4094u, 8);
        return;
    }
    if (blockIdx.x >= 1 && blockIdx.x < 8) return;
    const uint32_t bid = blockIdx.x - 8u;           // (block 0: wraps; hasWork is false for it below)
    const uint32_t nbActive = (n + GS_SORT_TILE - 1) / GS_SORT_TILE;
    const uint32_t perXcd = (nbActive + 7u) / 8u;
    const uint32_t tile = (bid & 7u) * perXcd + (bid >> 3);
    const bool hasWork = blockIdx.x >= 8 && (bid >> 3) < perXcd && tile < nbActive;
    const uint32_t base = tile * GS_SORT_TILE;
    if (!hasWork && blockIdx.x != 0) return;        // block 0 always publishes the tile ranges
    const uint32_t cntHere = hasWork ? min((uint32_t)GS_SORT_TILE, n - base) : 0u;
    const uint32_t mask = GS_WIDE_BINS - 1;

    {   // first pair of every tile id = exclusive scan of the per-tile totals (ITEMS consecutive ids per thread)
        static_assert(GS_WIDE_BINS == GS_SORT_TILE && ITEMS % 4 == 0, "one tile id per element slot");
        uint32_t v[ITEMS], sum = 0;
        const uint4* tt = reinterpret_cast<const uint4*>(tileTotal) + tid * (ITEMS / 4);
#pragma unroll
        for (int k = 0; k < ITEMS / 4; k++) { const uint4 q = tt[k]; v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w; }
#pragma unroll
        for (int k = 0; k < ITEMS; k++) sum += v[k];
        uint32_t tot;
        uint32_t run = block_excl_scan(sum, sm, &tot);
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const int t = tid * ITEMS + k;
            baseS[t] = run;
            // compute_tile_ranges (:314-344): [first, last + 1) of the tiles that have pairs, (0, 0) otherwise
            if (blockIdx.x == 0 && t < T) {
                tileRanges[2 * t] = v[k] ? run : 0u;
                tileRanges[2 * t + 1] = v[k] ? run + v[k] : 0u;
            }
            run += v[k];
        }
    }
    GS_PROBE_MARK(blockIdx.x < 3000u ? blockIdx.x : 4094u, 1);
    if (!hasWork) return;

    {
        uint32_t key[ITEMS], val[ITEMS], pos[ITEMS];
        // (unconditional loads from clamped addresses: a load under `if (i < cnt)` into a register array makes the
        // compiler merge the whole array at every branch and wait for each load before the next)
        const uint32_t lastIdx = base + cntHere - 1u;
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t i = base + w * PER_WAVE + r * 64 + lane;
            key[r] = keysIn[min(i, lastIdx)];
            val[r] = HAS_VALS ? valsIn[min(i, lastIdx)] : 0u;
        }
        // step 1: by the low 8 bits of the tile id
        local_rank_pass<NW, 0>(key, cntHere, [=](uint32_t k) { return (k >> shift) & 255u; }, match, waveRun, sm, pos);
        __syncthreads();      // the match tables (in keyS) are dead
        GS_PROBE_MARK(blockIdx.x < 3000u ? blockIdx.x : 4094u, 2);
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t i = w * PER_WAVE + r * 64 + lane;
            if (i < cntHere) { keyS[pos[r]] = key[r]; if (HAS_VALS) valS[pos[r]] = val[r]; }
        }
        __syncthreads();
        // step 2: by the high 4 bits, reading the elements back in step-1 order
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t i = min((uint32_t)(w * PER_WAVE + r * 64 + lane), cntHere - 1u);
            key[r] = keyS[i];
            val[r] = HAS_VALS ? valS[i] : 0u;
        }
        __syncthreads();      // every element is in registers: keyS turns into the match tables again
        GS_PROBE_MARK(blockIdx.x < 3000u ? blockIdx.x : 4094u, 3);
        local_rank_pass<NW, 4>(key, cntHere, [=](uint32_t k) { return (k >> (shift + 8)) & 15u; }, match, waveRun, sm, pos);
        __syncthreads();
        GS_PROBE_MARK(blockIdx.x < 3000u ? blockIdx.x : 4094u, 4);
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t i = w * PER_WAVE + r * 64 + lane;
            if (i < cntHere) { keyS[pos[r]] = key[r]; if (HAS_VALS) valS[pos[r]] = val[r]; }
        }
    }
    __syncthreads();          // keyS sorted by tile id (stable)
    GS_PROBE_MARK(blockIdx.x < 3000u ? blockIdx.x : 4094u, 5);

    // the first element of every run adds what precedes the run's tile in earlier sort tiles and takes its own position
    // off; a few gathers in flight per thread at a time (one run per tile id in a sorted block: no two writers)
    const uint32_t chunk = tile / GS_WIDE_CHUNK;
    const uint32_t* __restrict__ chunkRow = chunkSum + (size_t)chunk * GS_WIDE_BINS;
    const uint16_t* __restrict__ cntRow = cnt + (size_t)tile * GS_WIDE_BINS;
#pragma unroll 1
    for (int k0 = 0; k0 < ITEMS; k0 += 4) {
        uint32_t runD[4], add[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t p = tid + (k0 + k) * THREADS;
            runD[k] = 0xFFFFFFFFu;
            add[k] = 0;
            if (p < cntHere) {
                const uint32_t D = (keyS[p] >> shift) & mask;
                if (p == 0 || ((keyS[p - 1] >> shift) & mask) != D) {
                    runD[k] = D;
                    add[k] = chunkRow[D] + cntRow[D] - p;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (runD[k] != 0xFFFFFFFFu) baseS[runD[k]] += add[k];
    }
    __syncthreads();
    GS_PROBE_MARK(blockIdx.x < 3000u ? blockIdx.x : 4094u, 6);
    for (uint32_t p = tid; p < cntHere; p += THREADS) {
        const uint32_t k = keyS[p];
        const uint32_t dst = baseS[(k >> shift) & mask] + p;
        if (dst >= nMax) { *oob = 1u; continue; }      // see radix_scatter_kernel: never a store out of bounds
        keysOut[dst] = k;
        if (HAS_VALS) valsOut[dst] = valS[p];
    }
    GS_PROBE_MARK(blockIdx.x < 3000u ? blockIdx.x : 4094u, 7);
}

// ---------------------------------------------------------------------------------------------
// tile ranges / counts / dense table  (compute_tile_ranges :314-344, ..._counts :353-367, build_packed :377-404)
// ---------------------------------------------------------------------------------------------
__global__ void tile_ranges_kernel(const uint32_t* __restrict__ sortedKeys, int idxBits,
                                   const uint32_t* __restrict__ counters, uint32_t* __restrict__ tileRanges)
{
    const uint32_t M = counters[GS_CNT_M];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        const uint32_t cur = sortedKeys[i] >> idxBits;
        if (i == 0) tileRanges[cur * 2] = 0;
        else {
            const uint32_t prev = sortedKeys[i - 1] >> idxBits;
            if (cur != prev) { tileRanges[prev * 2 + 1] = i; tileRanges[cur * 2] = i; }
        }
        if (i == M - 1) tileRanges[cur * 2 + 1] = M;
    }
}

__global__ void unpack_idx_kernel(const uint32_t* __restrict__ packed, uint32_t mask,
                                  const uint32_t* __restrict__ counters, uint32_t* __restrict__ out)
{
    const uint32_t M = counters[GS_CNT_M];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) out[i] = packed[i] & mask;
}

__global__ void tile_counts_kernel(int T, const uint32_t* __restrict__ tileRanges, uint32_t* __restrict__ tileCounts,
                                   uint32_t* __restrict__ counters, const uint32_t* __restrict__ visPerBlock, int visBlocks)
{
    if (blockIdx.x == 0) {       // statistics only: visible Gaussians = sum of the projection's per-block counts
        uint32_t v = 0;
        for (int i = threadIdx.x; i < visBlocks; i += blockDim.x) v += visPerBlock[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&counters[GS_CNT_NVIS], v);
    }
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cnt = 0;
    if (t < T) {
        const uint32_t s = tileRanges[2 * t], e = tileRanges[2 * t + 1];
        cnt = e > s ? e - s : 0u;
        tileCounts[t] = cnt;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt = max(cnt, (uint32_t)__shfl_xor((int)cnt, d, 64));
    if ((threadIdx.x & 63) == 0 && cnt) atomicMax(&counters[GS_CNT_B], cnt);
}

__global__ void build_packed_tile_indices_kernel(uint32_t T, uint32_t B, const uint32_t* __restrict__ sortedIdx,
                                                 const uint32_t* __restrict__ tileRanges, int32_t* __restrict__ out)
{
    const size_t total = (size_t)T * B;
    for (size_t lin = (size_t)blockIdx.x * blockDim.x + threadIdx.x; lin < total; lin += (size_t)gridDim.x * blockDim.x) {
        const uint32_t tile = (uint32_t)(lin / B), slot = (uint32_t)(lin - (size_t)tile * B);
        const uint32_t s = tileRanges[2 * tile], e = tileRanges[2 * tile + 1];
        const uint32_t cnt = e > s ? e - s : 0u;
        out[lin] = slot < cnt ? (int32_t)sortedIdx[s + slot] : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
int launch_bin_prep(gs_ctx* c, int N, const float* rectMin, const float* rectMax, const float* radii,
                    const float* depths)
{
    c->piecesValid = false;       // (the op-level rects are the reference's)
    if (N == 0) return GS_OK;
    hipLaunchKernelGGL(bin_prep_kernel, dim3(gs_div_up(N, 256)), dim3(256), 0, c->stream, N, c->tileW, c->tileH,
                       c->gridW, c->gridH, rectMin, rectMax, radii, depths, c->tileRect, c->tilesTouched,
                       c->depthKey[0], c->depthVal[0], c->visPerBlock, c->counters, gs_small_depth_sort(N) ? 1 : 0);
    c->visBlocks = gs_div_up(N, 256);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

// Expects tileRect / tilesTouched / depthKey[0] / depthVal[0] filled for N Gaussians and counters zeroed
// (except NVIS).  Leaves the sorted list, tileRanges and counters[M] for the blend kernels.
// When tile bits + index bits fit in 32, a pair is ONE packed word (tile << idxBits | index): the two tile passes
// move 4 B per pair instead of 8 and the list is read through c->sortedRaw & c->idxMask; c->sortedIdx (plain
// indices) is produced only when wantPlain (op-level entry points, generic blend kernels) or on export.
int cut_super_width(const gs_ctx* c) { return gs_div_up(c->gridW, GS_CUT_SUPER); }

int launch_cut_super(gs_ctx* c, const uint32_t* cuts)
{
    const int superW = cut_super_width(c), superH = gs_div_up(c->gridH, GS_CUT_SUPER);
    hipLaunchKernelGGL(cut_super_kernel, dim3(gs_div_up(superW * superH, 256)), dim3(256), 0, c->stream, c->gridW, c->gridH, superW,
                       superH, cuts, c->superCut);
    GS_HIP_CHECK(c, hipGetLastError());
    c->superCutReady = true;
    return GS_OK;
}

int launch_binning(gs_ctx* c, int N, bool wantPlain)
{
    if (N == 0) GS_HIP_CHECK(c, hipMemsetAsync(c->tileRanges, 0, sizeof(uint32_t) * 2 * c->T, c->stream));
    int idxBits = 1;
    while ((1LL << idxBits) < (long long)N) idxBits++;
    const bool packed = idxBits + c->tileBits <= 32;
    c->idxBits = packed ? idxBits : 0;
    c->idxMask = packed ? (uint32_t)((1ull << idxBits) - 1ull) : 0xFFFFFFFFu;
    c->sortedRaw = packed ? c->pairKey[0] : c->pairVal[0];
    c->sortedIdx = c->pairVal[0];
    c->sortedPlainValid = !packed;
    if (N == 0) { c->sortedPlainValid = true; return GS_OK; }
    // 1. depth sort over N (n known on the host: nPtr == nullptr means n = nMax)
    int res = 0;
    int rc = radix_sort(c, c->depthKey, c->depthVal, true, nullptr, (uint32_t)N, 0, 32, &res);
    if (rc) return rc;
    const uint32_t* sortedG = c->depthVal[res];
    // 2. scan
    const int nb = gs_div_up(N, GS_SCAN_BLOCK);
    // set by gs_render_forward for this forward only, or by gs_tile_bin_cut for this call only
    const uint32_t* cuts = c->fwd.cutsActive ? c->fwd.cutStore : c->opCuts;
    const uint32_t* sortedKey = c->depthKey[res];
    hipLaunchKernelGGL(scan_blocksum_kernel, dim3(nb), dim3(GS_SCAN_BLOCK), 0, c->stream, N, sortedG, c->tilesTouched,
                       c->blockSums);
    // 3. expand
    const bool bigScan = nb > GS_FUSED_SCAN_MAX;
    // large inputs: every wave's positions in slices (grid y), for the blocks that have many.  Small inputs too: 10 k
    // Gaussians are 40 blocks on 256 CUs, each wave walking its 64 Gaussians' ~4.6 k positions alone (37 us of the
    // 10 k / 400x400 forward's 320; 8 us sliced; 100 k: 21 -> 13 us; at 300 k -- five blocks per CU -- slicing costs
    // 12 us instead: every slice repeats the block's prologue)
    const bool fewBlocks = nb < 2 * c->numCUs;
    const int slices = (bigScan || fewBlocks) ? GS_EXPAND_SLICES : 1;
    const uint32_t sliceMinPairs = bigScan ? GS_SLICE_MIN_PAIRS : 2048u;
    if (bigScan) launch_prefix(c, nb, c->blockSums, 1, c->scanPrefix);
    const uint4* pieces = c->piecesValid ? c->tilePieces : nullptr;      // (the fused projection of this forward trimmed its rects into row groups)
    auto expand = cuts ? (pieces ? expand_kernel<true, true> : expand_kernel<true, false>)
                       : (pieces ? expand_kernel<false, true> : expand_kernel<false, false>);
    const int superW = cut_super_width(c);
    const uint32_t* superCut = nullptr;
    if (cuts && c->superCut && c->cutSuper) {
        if (!c->superCutReady) { const int src = launch_cut_super(c, cuts); if (src) return src; }     // (the fused projection has it already)
        superCut = c->superCut;
    }
    c->superCutReady = false;
    hipLaunchKernelGGL(expand, dim3(nb, slices), dim3(GS_SCAN_BLOCK), 0, c->stream, N, c->gridW, c->idxBits, sortedG,
                       c->tilesTouched, c->tileRect, c->blockSums, c->counters, (unsigned long long)c->capM, c->tileRanges,
                       2 * c->T, c->pairKey[0], c->pairVal[0], sortedKey, cuts, c->waveSeg,
                       bigScan ? c->scanPrefix : nullptr, c->missDev, sliceMinPairs, superCut, superW, pieces);
    uint32_t* pk[2] = {c->pairKey[0], c->pairKey[1]};
    uint32_t* pv[2] = {c->pairVal[0], c->pairVal[1]};
    if (cuts) {     // the cut expansion left gaps: the compacted pairs are in the second buffers, the sort starts there
        const int nSeg = nb * (GS_SCAN_BLOCK / 64) * slices;
        if (bigScan)        // the expansion is done with the buffer by now (stream order)
            launch_prefix(c, nSeg, reinterpret_cast<const uint32_t*>(c->waveSeg) + 1, 2, c->scanPrefix);
        hipLaunchKernelGGL(compact_pairs_kernel, dim3(gs_div_up(nSeg, GS_SCAN_BLOCK / 64)), dim3(GS_SCAN_BLOCK), 0, c->stream, nSeg,
                           c->waveSeg, c->pairKey[0], packed ? nullptr : c->pairVal[0], c->pairKey[1], c->pairVal[1],
                           c->counters, c->missDev, bigScan ? c->scanPrefix : nullptr, c->dropPerBlock, c->dropBlocks);
        pk[0] = c->pairKey[1]; pk[1] = c->pairKey[0];
        pv[0] = c->pairVal[1]; pv[1] = c->pairVal[0];
    }
    GS_HIP_CHECK(c, hipGetLastError());
    // 4. tile sort over M (device-resident count), tile bits only
    if (c->T <= GS_WIDE_BINS && c->wideCnt && c->wideTileSort) {
        // one pass: histogram per sort tile, two-level prefix, local sort + scatter; the ranges come with it
        const uint32_t* mPtr = c->counters + GS_CNT_M;
        const int nbAll = gs_div_up(c->capM, GS_SORT_TILE);
        if (nbAll > 0) {
            const int shift = c->idxBits;          // packed: tile id above the index bits; key + value: the key is the tile id
            hipLaunchKernelGGL(wide_hist_kernel, dim3(nbAll < GS_SORT_MAX_GRID ? nbAll : GS_SORT_MAX_GRID), dim3(GS_SORT_THREADS), 0,
                               c->stream, pk[0], mPtr, (uint32_t)c->capM, shift, c->wideCnt);
            hipLaunchKernelGGL(wide_chunk_kernel, dim3(gs_div_up(nbAll, GS_WIDE_CHUNK), GS_WIDE_BINS / 256), dim3(256), 0, c->stream,
                               mPtr, (uint32_t)c->capM, c->wideCnt, c->wideChunk);
            int rb = 0;
            const ColourRider ra = rider_take(c, GS_RIDE_WIDE_TILE, 256, &rb);
            hipLaunchKernelGGL(wide_tile_kernel, dim3(GS_WIDE_BINS / 256 + rb), dim3(256), 0, c->stream, mPtr, (uint32_t)c->capM,
                               c->wideChunk, c->wideTotal, ra, GS_WIDE_BINS / 256);
            SegBaseArgs seg = {};
            const int withSeg = c->segBaseWanted ? 1 : 0;
            if (withSeg) {
                fill_seg_base(c, seg);
                seg.tileRanges = nullptr;           // the ranges are being written by block 0 of the same launch
                seg.tileTotal = c->wideTotal;
                c->segBaseDone = true;
            }
            // sort tile t of XCD x is block 8 + 8 (t mod perXcd) + x with perXcd = ceil(active tiles / 8): the grid has to
            // reach 8 ceil(nbAll / 8) blocks behind the eight spare ones, or the last tiles of a reserve that is not a
            // multiple of 8 sort tiles would find no block when M comes within 7 tiles of it
            const int scatterGrid = 8 * gs_div_up(nbAll, 8) + 8;
            // waves per sort tile: sixteen (1024 threads, four elements each) while the sort tiles are fewer than the CUs -- a
            // block then has its CU to itself and its time is the ranking chain's --, eight beyond that (three blocks of 41 KB
            // per CU).  MI355X, kernel trace: 10 k Gaussians / 176 sort tiles 21.4 (four waves) -> 14.8 (eight) -> 12.5 us
            // (sixteen); 300 k / 1751 tiles 54.4 -> 48.7 -> 57.0; 1 M / 5200 tiles 92.1 -> 83.3 -> 104.4.  The tile count is on
            // the device; the Gaussian count stands in for it.
            const int st = c->scatterThreads ? c->scatterThreads : N <= 65536 ? 1024 : 512;
            auto launch_scatter = [&](auto kern, int threads, const uint32_t* vIn, uint32_t* vOut) {
                hipLaunchKernelGGL(kern, dim3(scatterGrid), dim3(threads), 0, c->stream, pk[0], vIn, pk[1], vOut, mPtr, (uint32_t)c->capM,
                                   shift, c->wideCnt, c->wideChunk, c->wideTotal, c->tileRanges, c->T, seg, withSeg,
                                   c->counters + GS_CNT_OVERFLOW);
            };
            if (packed) {
                if (st == 1024) launch_scatter(wide_scatter_kernel<false, 1024>, 1024, nullptr, nullptr);
                else if (st == 512) launch_scatter(wide_scatter_kernel<false, 512>, 512, nullptr, nullptr);
                else launch_scatter(wide_scatter_kernel<false, 256>, 256, nullptr, nullptr);
            } else {
                if (st == 1024) launch_scatter(wide_scatter_kernel<true, 1024>, 1024, pv[0], pv[1]);
                else if (st == 512) launch_scatter(wide_scatter_kernel<true, 512>, 512, pv[0], pv[1]);
                else launch_scatter(wide_scatter_kernel<true, 256>, 256, pv[0], pv[1]);
            }
        }
        GS_HIP_CHECK(c, hipGetLastError());
        c->sortedRaw = packed ? pk[1] : pv[1];
        if (!packed) c->sortedIdx = pv[1];
    } else {
        rc = radix_sort(c, pk, pv, !packed, c->counters + GS_CNT_M, (uint32_t)c->capM, c->idxBits,
                        c->idxBits + c->tileBits, &res);
        if (rc) return rc;
        c->sortedRaw = packed ? pk[res] : pv[res];
        if (!packed) c->sortedIdx = pv[res];
        // 5. ranges
        const int rb = (int)((c->capM + 255) / 256 < 2048 ? (c->capM + 255) / 256 : 2048);
        hipLaunchKernelGGL(tile_ranges_kernel, dim3(rb > 0 ? rb : 1), dim3(256), 0, c->stream, pk[res], c->idxBits,
                           c->counters, c->tileRanges);
        GS_HIP_CHECK(c, hipGetLastError());
    }
    if (wantPlain) return ensure_plain_sorted(c);
    return GS_OK;
}

// plain Gaussian indices of the sorted list (c->sortedIdx), unpacked on demand
int ensure_plain_sorted(gs_ctx* c)
{
    if (c->sortedPlainValid) return GS_OK;
    uint32_t* dst = (c->sortedRaw == c->pairKey[0]) ? c->pairKey[1] : c->pairKey[0];   // the other key buffer is free
    const int rb = (int)((c->capM + 255) / 256 < 2048 ? (c->capM + 255) / 256 : 2048);
    hipLaunchKernelGGL(unpack_idx_kernel, dim3(rb > 0 ? rb : 1), dim3(256), 0, c->stream, c->sortedRaw, c->idxMask,
                       c->counters, dst);
    GS_HIP_CHECK(c, hipGetLastError());
    c->sortedIdx = dst;
    c->sortedPlainValid = true;
    return GS_OK;
}

int launch_tile_counts(gs_ctx* c)
{
    GS_HIP_CHECK(c, hipMemsetAsync(c->counters + GS_CNT_B, 0, sizeof(uint32_t), c->stream));
    GS_HIP_CHECK(c, hipMemsetAsync(c->counters + GS_CNT_NVIS, 0, sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(tile_counts_kernel, dim3(gs_div_up(c->T, 256)), dim3(256), 0, c->stream, c->T, c->tileRanges,
                       c->tileCounts, c->counters, c->visPerBlock, c->visBlocks);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int launch_build_packed_tile_indices(gs_ctx* c, uint32_t B, int32_t* out)
{
    if (B == 0 || c->T == 0) return GS_OK;
    const size_t total = (size_t)c->T * B;
    const int nb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(build_packed_tile_indices_kernel, dim3(nb), dim3(256), 0, c->stream, (uint32_t)c->T, B,
                       c->sortedIdx, c->tileRanges, out);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

}  // namespace gs
