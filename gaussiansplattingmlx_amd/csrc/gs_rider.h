// gs_rider.h -- the SH colour of the fused forward as RIDER workgroups.
//
// The fused forward's projection kernel was geometry (44 B in, 64 B out per Gaussian) and SH colour (12 K bytes in, 16 out) in
// one launch, 37 us on the bench scene, three quarters of it the SH rows' HBM time -- and only the blend needs the colours.
// Between the projection and the blend lies the binning chain: a dozen small dependent kernels (depth sort, scan, tile sort)
// that are bound by their own latency and leave most of the chip's CUs and nearly all of its HBM bandwidth idle.  A second
// stream cannot use that (one fork + join costs ~20 us on this stack, tools/forkjoin_cost.py); extra workgroups in the SAME
// launches can: every binning kernel that has room takes a share of the colour work as workgroups behind its own
// (blockIdx.x >= ownBlocks), which the dispatcher places on the CUs the kernel leaves empty.  Whatever is left when the
// binning is done runs as a kernel of its own in front of the blend (colour_rest_kernel, projection.hip).
//
// A rider unit is 64 consecutive Gaussians = one wave: the wave's SH-rest rows go through LDS in two halves of 12
// coefficients exactly as in proj_fwd_fused_kernel (same helpers, same order of the sum: the colours are the same bits), rows
// of Gaussians that touch no tile are not fetched at all (nothing reads their colour; on the 2 M garden scene that is 48 %
// of the rows), and the three colour floats + the gate word go into the packed record the geometry kernel has written.
// K = 25 only (the app's SH degree 4); other K keep the one-kernel projection.
#pragma once
#include "gs_ctx.h"
#include "gs_math.h"

namespace gs {

constexpr int SH_HALF_MAX4 = 9;       // float4 per lane: 64 rows x 36 floats / 64 lanes / 4

// Half-row staging: columns [c0, c0 + LH) of the wave's rows, LH % 4 == 0, LH <= 36.  The loads go to registers first
// (sh_half_load), so both halves can be in flight from the top while only ONE half-sized LDS buffer exists: 9.5 KB per
// wave instead of 18.7, twice the resident waves for an HBM-bound kernel.  rowMask: bit r = fetch row r.
__device__ __forceinline__ void sh_half_load(const float* __restrict__ g, int rows, int L, int c0, int LH, int lane,
                                             float4 (&regs)[SH_HALF_MAX4], unsigned long long rowMask = ~0ull)
{
    const int per4 = LH >> 2, total4 = rows * per4;
#pragma unroll
    for (int i = 0; i < SH_HALF_MAX4; i++) {
        const int e = lane + 64 * i;
        regs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < total4) {
            const int r = e / per4, c = (e - r * per4) * 4;
            if ((rowMask >> r) & 1ull) regs[i] = *reinterpret_cast<const float4*>(g + (size_t)r * L + c0 + c);
        }
    }
}
__device__ __forceinline__ void sh_half_to_lds(float* __restrict__ lds, int rows, int LH, int lane,
                                               const float4 (&regs)[SH_HALF_MAX4])
{
    const int per4 = LH >> 2, total4 = rows * per4;
#pragma unroll
    for (int i = 0; i < SH_HALF_MAX4; i++) {
        const int e = lane + 64 * i;
        if (e < total4) {
            const int r = e / per4, c = (e - r * per4) * 4;
            float* d = lds + r * (LH + 1) + c;
            d[0] = regs[i].x; d[1] = regs[i].y; d[2] = regs[i].z; d[3] = regs[i].w;
        }
    }
}

typedef ::GsColourRider ColourRider;

constexpr int GS_RIDER_K = 25, GS_RIDER_L = (GS_RIDER_K - 1) * 3, GS_RIDER_LH = GS_RIDER_L / 2;
constexpr int GS_RIDER_ROW = GS_RIDER_LH + 1;                   // LDS row pitch in floats (odd: conflict-free columns)
constexpr int GS_RIDER_WAVES = 4;                               // waves of a rider workgroup that work (38 KB of LDS)

// the colour of one max(., 0)-gated channel triple and its gate word (2 bits per channel: 0 below, 1 tie, 2 above), as
// proj_fwd_fused_kernel writes them
__device__ __forceinline__ uint32_t colour_gate(float& c0, float& c1, float& c2)
{
    const uint32_t gate = (c0 > 0.f ? 2u : (c0 == 0.f ? 1u : 0u)) | (c1 > 0.f ? 8u : (c1 == 0.f ? 4u : 0u)) |
                          (c2 > 0.f ? 32u : (c2 == 0.f ? 16u : 0u));
    c0 = c0 > 0.f ? c0 : 0.f; c1 = c1 > 0.f ? c1 : 0.f; c2 = c2 > 0.f ? c2 : 0.f;
    return gate;
}

// one wave, one unit.  myRows: 64 x GS_RIDER_ROW floats of LDS private to the wave.
// touched: >= 0 when the caller has the lane's tiles-touched count at hand (the projection kernel's own wave: no read-back
// of a word the lane has just stored); < 0: read it.
__device__ __forceinline__ void colour_rider_wave(const ColourRider& r, int unit, float* __restrict__ myRows, int lane,
                                                  int touched = -1)
{
    constexpr int L = GS_RIDER_L, LH = GS_RIDER_LH, kSplit = 1 + (GS_RIDER_K - 1) / 2;
    const int row0 = unit * 64;
    const int rows = min(64, r.N - row0);
    if (rows <= 0) return;
    const int p = row0 + lane;
    const bool want = p < r.N && (touched >= 0 ? touched != 0 : r.tilesTouched[p] != 0u);
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return;                       // wave-uniform
    float4 halfA[SH_HALF_MAX4], halfB[SH_HALF_MAX4];
    sh_half_load(r.frest + (size_t)row0 * L, rows, L, 0, LH, lane, halfA, mask);
    sh_half_load(r.frest + (size_t)row0 * L, rows, L, LH, LH, lane, halfB, mask);
    float x = 0.f, y = 0.f, z = 0.f, d0[3] = {0.f, 0.f, 0.f};
    if (want) {
        x = r.xyz[3 * p] - r.cam[0]; y = r.xyz[3 * p + 1] - r.cam[1]; z = r.xyz[3 * p + 2] - r.cam[2];
        d0[0] = r.fdc[(size_t)p * 3]; d0[1] = r.fdc[(size_t)p * 3 + 1]; d0[2] = r.fdc[(size_t)p * 3 + 2];
    }
    sh_half_to_lds(myRows, rows, LH, lane, halfA);
    // each wave reads back only what it staged itself: DS operations of one wave complete in order
    const float* rest = myRows + lane * GS_RIDER_ROW;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (want)        // same sum, same order (k ascending) as the one-kernel projection
        sh_foreach(r.degree, x, y, z, [&](int k, float b, float, float, float) {
            if (k == 0) { c0 = b * d0[0]; c1 = b * d0[1]; c2 = b * d0[2]; }
            else if (k < kSplit) {
                const float* q = rest + (k - 1) * 3;
                c0 += b * q[0]; c1 += b * q[1]; c2 += b * q[2];
            }
        });
    sh_half_to_lds(myRows, rows, LH, lane, halfB);       // the first half is consumed
    if (want) {
        sh_foreach(r.degree, x, y, z, [&](int k, float b, float, float, float) {
            if (k >= kSplit) {
                const float* q = rest + (k - kSplit) * 3;
                c0 += b * q[0]; c1 += b * q[1]; c2 += b * q[2];
            }
        });
        c0 += 0.5f; c1 += 0.5f; c2 += 0.5f;
        const uint32_t gate = colour_gate(c0, c1, c2);
        float* rec = r.packed12 + (size_t)p * 12;
        *reinterpret_cast<float2*>(rec + 6) = make_float2(c0, c1);
        rec[8] = c2;
        rec[11] = __uint_as_float(gate);
    }
}

// a rider workgroup of any size >= 64 threads: its first `waves` waves take one unit each.  rows: waves x 64 x GS_RIDER_ROW
// floats of LDS -- the host kernel's own arrays where it has them (a rider block never runs the host's code), so that
// riding costs the host no LDS and no occupancy
__device__ __forceinline__ void colour_rider_block(const ColourRider& r, int riderBlock, float* rows, int waves)
{
    const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int nW = min((int)(blockDim.x >> 6), waves);
    if (wv >= nW) return;
    const int u = riderBlock * nW + wv;
    if (u < r.units) colour_rider_wave(r, r.unit0 + u, rows + wv * 64 * GS_RIDER_ROW, lane);
}
// ... with LDS of its own (hosts that have none to lend)
__device__ __forceinline__ void colour_rider_block(const ColourRider& r, int riderBlock)
{
    __shared__ float riderRows[GS_RIDER_WAVES * 64 * GS_RIDER_ROW];
    colour_rider_block(r, riderBlock, riderRows, GS_RIDER_WAVES);
}

// workgroups of `threads` threads that `units` units need
static inline int rider_blocks(int units, int threads, int waves = GS_RIDER_WAVES)
{
    const int nW = threads / 64 < waves ? threads / 64 : waves;
    return (units + nW - 1) / nW;
}

}  // namespace gs
