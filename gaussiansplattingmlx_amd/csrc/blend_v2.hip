// blend_v2.hip -- the fused path's alpha-blend kernels (gs_render_forward / gs_render_backward*).
//
// Same arithmetic as blend.hip (which keeps serving the op-level entry points), different mapping.  What the
// measurements on MI355X said (tools/microbench.hip, tools/fwd_trace.py, tools/pmc_fwd.sh):
//
//  * Issue rates, cycles per wave64 instruction per SIMD: v_mul/add/sub/fmac_f32, v_mov, integer add/and/shift
//    2.6-3.0; v_min/max/med3_f32, v_cmp, v_cndmask, v_bfi 4.2-4.5; v_exp_f32 / v_rcp_f32 8.3; v_pk_*_f32 4.8 (no
//    packed advantage); v_readlane_b32 12; DPP add 4.2; v_permlane{16,32}_swap 13.6.  Both kernels end up bound by
//    f32 instruction issue (SQ counters: forward ~100 %, backward 96 % VALU-busy), so the levers are instruction
//    count and not doing work.  The file is built with -fno-slp-vectorize: the SLP vectoriser's v_pk_* forms buy
//    nothing here and cost ~8 v_mov per splat in operand shuffles.
//  * Every wavefront is autonomous: it gathers 64 list entries at a time, one per lane (a coalesced index burst +
//    three 16-B loads per lane), parks them in a wave-private LDS slot and reads entry j back as three broadcast
//    ds_read_b128.  No workgroup barriers; the next 64 entries are in flight while the current ones are blended.
//    (v_readlane broadcast: 131 cycles per record; scalar loads: two dependent scalar-cache misses per group.)
//  * Staging COMPACTS: lane j evaluates, in closed form, the minimum of entry j's quadratic form over the wave's
//    pixel rectangle (rect_min_q) and drops the entry if exp(-q/2) is below 2^-29 = 2e-9 on every pixel (CULL_QMIN).
//    The reference blends everything in its bounding squares, which are much larger than the ellipses; what is
//    dropped moves no rendered value by more than ~5e-8 per entry, and the image's distance to the oracle does not
//    change in the third digit (measured for thresholds 2^-42 ... 2^-24.5; at 2^-20 it does).  Dropped entries cost
//    no LDS broadcast and no VALU.
//  * Forward (blend_fwd_v2q_kernel): one wavefront per 8x8 quadrant of a 16x16 block, one pixel per lane, four
//    splats per trip (exponents and exps first, the short serial chain through T second), branch-free termination
//    (a finished pixel keeps blending with alpha = 0), wave-uniform exit.  Persistent waves pull items from a
//    device queue; the first item of a wave is its blockIdx.x (thousands of simultaneous pops on one counter take
//    ~6 ns each to resolve: 15 % of the kernel).  With a per-view block-work buffer (gs_set_block_work_buffer) the
//    queue is ordered deepest-first, which takes the slowest wave from 1.55x to 1.16x the mean.  Every SEG list
//    positions the running state (T, C, D) is saved per pixel.  blend_fwd_v2_kernel is the older 16x8, two pixels
//    per lane, packed-f32 variant (gs_ctx_set_tuning(GS_TUNE_FWD_QUADRANTS, 0) selects it): same results, ~25 % slower.
//  * Backward (blend_bwd_v2_kernel): the saved states make a tile's list SEGMENT-parallel.  The work items are
//    (pixel block, segment) pairs of at most SEG list positions, pulled from a device queue by persistent
//    single-wave workgroups.  Inside an item the sweep runs FORWARD (T by multiplication, as the forward pass), with
//    the cotangent of T in closed form,
//        c_i = (sum_{j>i} T_j a_j S_j + T_n cT_n) / T_{i+1},
//    the sum being (final colour - running colour) . cotColour.  Each lane owns 4 pixels, so the per-splat wave
//    reduction is paid once per 256 pixel-splats; it is a TRANSPOSED reduction (wave_sum10_transposed): lane swaps
//    and bank-masked DPP adds halve the live registers level by level, 23 instructions instead of 60.
//  * The reference rebuilds T from the rounded output alpha (T_n' = 1 - outAlpha) and divides its way back;
//    every T of a pixel is therefore off by the factor s = T_n' / T_n.  The same factor is applied here, so
//    the gradients match the reference's arithmetic, not just the exact calculus.
//
// Compiled with -ffp-contract=off and explicit fmaf so that forward and backward evaluate the running sums
// bit-identically (the closed form needs the backward's recomputed colour to meet the forward's).
#include "gs_ctx.h"
#include "gs_bwd_prep.h"
#include "gs_cull.h"
#include "gs_wavesum.h"

namespace gs {

constexpr int BLK = 16;
constexpr uint32_t CKPT_POOL_MAX = 8;   // checkpoint slots a forward wave takes from the arena's shared part per atomic (blend_fwd_v2*_kernel):
                                        // this many when the arena is roomy, fewer when a part of it holds less than that per wave
                                        // (ckpt_pool below) -- a tiny reserve (capM of a few hundred thousand pairs: an arena of ~10 k
                                        // slots for 4 - 5 k persistent waves) was used up by the waves' unused remainders, every
                                        // forward reported an arena overflow until the regrow (round 5: tools/dp_overflow_rehearsal.py
                                        // one_view, never re-run after the four-wave forward had multiplied the waves by four)
static inline uint32_t ckpt_pool(uint32_t partSlots, uint32_t waves)
{
    const uint32_t perWave = partSlots / (waves / 8u + 1u);
    return perWave >= CKPT_POOL_MAX ? CKPT_POOL_MAX : (perWave < 1u ? 1u : perWave);
}

struct Rec {
    float mx, my, c00, c01, c10, c11, r, g, b, op, depth;
};

#ifndef GS_FWD_PREFETCH
#define GS_FWD_PREFETCH 1      // register sets of the one-wave forward's record pipeline (blend_fwd_v2q_kernel)
#endif

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// one record per lane (lane j of a chunk holds splat chunkStart + j of the list)
struct RecV {
    f4 a, b, c;
};

__device__ __forceinline__ RecV load_chunk(const float4* __restrict__ packed12, const uint32_t* __restrict__ idx,
                                           uint32_t idxMask, uint32_t i0, uint32_t iEnd, int lane)
{
    RecV v;
    v.a = v.b = v.c = (f4){0.f, 0.f, 0.f, 0.f};
    if (i0 + lane < iEnd) {
        const f4* p = reinterpret_cast<const f4*>(packed12) + (size_t)(idx[i0 + lane] & idxMask) * 3;
        v.a = p[0]; v.b = p[1]; v.c = p[2];
    }
    return v;
}

// the same in two steps, so that the list index of chunk c + 2 can be in flight while the records of chunk c + 1 are
// gathered: otherwise the gather waits for its index first, two memory latencies in a row per chunk, which is what a
// deep quadrant that runs alone at the end of the kernel spends its time on
__device__ __forceinline__ uint32_t load_chunk_index(const uint32_t* __restrict__ idx, uint32_t idxMask, uint32_t i0,
                                                     uint32_t iEnd, int lane)
{
    return i0 + lane < iEnd ? idx[i0 + lane] & idxMask : 0xFFFFFFFFu;
}
__device__ __forceinline__ RecV gather_chunk(const float4* __restrict__ packed12, uint32_t g)
{
    RecV v;
    v.a = v.b = v.c = (f4){0.f, 0.f, 0.f, 0.f};
    if (g != 0xFFFFFFFFu) {
        const f4* p = reinterpret_cast<const f4*>(packed12) + (size_t)g * 3;
        v.a = p[0]; v.b = p[1]; v.c = p[2];
    }
    return v;
}

// park a chunk in the wave's LDS slot; DS operations of one wave complete in order, so the broadcast reads
// that follow need no barrier
__device__ __forceinline__ void stage_chunk(f4* slot, const RecV& v, int lane)
{
    slot[lane * 3] = v.a; slot[lane * 3 + 1] = v.b; slot[lane * 3 + 2] = v.c;
}

__device__ __forceinline__ Rec unpack(const f4 a, const f4 b, const f4 c)
{
    Rec r;
    r.mx = a.x; r.my = a.y; r.c00 = a.z; r.c01 = a.w;
    r.c10 = b.x; r.c11 = b.y; r.r = b.z; r.g = b.w;
    r.b = c.x; r.op = c.y; r.depth = c.z;
    return r;
}

// exp(-q/2) = 2^(q C), C = -1/(2 ln 2).  In f32 the product q C1 (C1 = fl(C)) is rounded to |q C1| 2^-24: a relative
// error of the result of up to ~1e-6 for the exponents that still matter (|q C| ~ 10-20), which is what separated this
// forward from the oracle's libm exp by 3.4e-4 on the bench scene's colours of magnitude ~27.
// The forward therefore carries the product's rounding error lo = fma(q, C1, -hi) (exact) and C's own tail C2 along:
//   exp(-q/2) = 2^hi (1 + (lo + q C2) ln 2),   four more VALU instructions per pixel-splat
// (measured: L-inf 3.41e-4 -> 4.3e-5 against the float32 oracle, the same as with the device math library's expf at a
// third of its cost; blend forward 0.195 -> 0.207 ms).  -DGS_EXP_PLAIN builds the uncompensated forward.  The backward
// only needs alpha to ~1e-6 (gradients are held to 1e-3): it stays plain, and evaluates the quadratic form with a
// pre-multiplied conic instead of in the reference's order (RecB, pair_exponent_bwd).
constexpr float EXP_C1 = -0.72134751081466675f;      // fl(-1 / (2 ln 2))
constexpr float EXP_C2LN2 = -6.674879e-09f;          // (C - C1) ln 2
constexpr float EXP_LN2 = 0.69314718055994531f;

// opacity * exp(-q/2)
__device__ __forceinline__ float gauss_alpha_raw(float q, float op)
{
    const float hi = q * EXP_C1;
#ifndef GS_EXP_PLAIN
    const float lo = fmaf(q, EXP_C1, -hi);
    const float d = fmaf(q, EXP_C2LN2, lo * EXP_LN2);
    const float g = __builtin_amdgcn_exp2f(hi);
    return op * fmaf(g, d, g);
#else
    return op * __builtin_amdgcn_exp2f(hi);
#endif
}

// (cull bound CULL_QMIN and rect_min_q: gs_cull.h)

// ---------------------------------------------------------------------------------------------
// segment bookkeeping
// ---------------------------------------------------------------------------------------------
// segBase[b] = index of pixel block b's first saved state (state before splat SEG); exclusive scan of
// max(ceil(count/SEG) - 1, 0).  One workgroup.
template <int SEG>
__global__ __launch_bounds__(1024) void seg_base_kernel(SegBaseArgs a)
{
    __shared__ uint32_t lds[GS_SEGBASE_LDS];
    seg_base_body<SEG>(a, lds);
}

// work items of the backward: (block, segment) for segment < ceil(blockWork/SEG).  One workgroup.
// It also renews the view's depth cuts (binning.hip) when the caller keeps them (cutStore != nullptr; 16x16 tiles, so
// tile = block): the cut of a tile whose sweep stopped `work` entries into a list of `len` is the depth key of entry
// 2 work + 128 if the list goes on beyond that (on the bench scene sweeps grow by up to 1.85x between two visits of a
// view; with work + work/4 + 64 a handful of the 2500 tiles missed in nearly every forward); a cut that was in force
// this forward (cutsInForce) and still leaves 1.5 work + 64 entries is kept; anything else means "bin everything
// next time".
template <int SEG>
__global__ __launch_bounds__(1024) void bwd_items_kernel(BwdPrepArgs prep, int cutBlocks)
{
    __shared__ uint32_t sm[17];
    // blocks 0..GS_ITEM_PARTS-1: the item list; the next cutBlocks: the view's depth cuts, one tile per thread; the blocks
    // behind: the accumulator clear (hidden under the item scan).  The loss kernel carries the same three along when the loss
    // of a fused forward is taken through the library (ssim.hip); this launch is for hosts with their own loss.
    constexpr int IP = GS_ITEM_PARTS;
    if ((int)blockIdx.x >= IP + cutBlocks) {
        bwd_clear_part(prep, blockIdx.x - IP - cutBlocks, gridDim.x - IP - cutBlocks);
        return;
    }
    if ((int)blockIdx.x >= IP) {
        bwd_cut_renew(prep, (int)(blockIdx.x - IP) * 1024 + (int)threadIdx.x);
        return;
    }
    bwd_items_scan<SEG>(prep, sm, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// packed-f32 helpers.  The f32 VALU of gfx950 retires one wave64 instruction per 4 cycles per SIMD (measured:
// both blend kernels ran at >90 % of that issue rate while "using" 25 % of the 157 TFLOP/s headline, which
// counts v_pk_fma_f32).  Each lane therefore owns pixel PAIRS (x, x+8) on one image row and does the pair's
// arithmetic with v_pk_mul/add/fma_f32; the row term dy is shared by the pair and stays scalar per lane.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f2 splat2(float v) { return (f2){v, v}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// ---------------------------------------------------------------------------------------------
// forward: one wavefront per 8x8 quadrant (h, k) of a 16x16 block, one pixel per lane, scalar f32.  (Rounds 1-2 also
// kept a 16x8 variant, two pixels per lane in packed f32: the same instruction slots -- v_pk_* has no throughput
// advantage -- but twice the per-splat dependent chain and a cull over 128 pixels instead of 64; 25 % slower, retired.)
//
// Checkpoints.  Every SEG list entries a quadrant that still has a live pixel saves its running state (T, R, G, B[, D])
// for the segment-parallel backward.  Round 2 addressed the slots by list position (slot = first slot of the block +
// segment), which sizes the arena for every list being swept to its end: 1.97 GB at the bench reserve, of which 0.1 GB
// was ever touched.  Now a slot is ALLOCATED when it is written: every persistent wave keeps a private pool of
// a few quadrant slots (one atomicAdd per CKPT_POOL_MAX boundaries, ckpt_pool), the slot's id goes to a small table indexed by
// (block's first segment + segment, quadrant) -- 16 B per 64 list entries -- and the backward's item kernel copies the
// four ids of a (block, segment) item into the item list, so the backward itself does no table look-up.  Stale table
// entries are never read: the backward loads a quadrant's state only for pixels that were live at the boundary, and
// then this forward wrote the entry.  An arena that runs out raises the overflow word like a pair reserve that does
// (the render itself is complete; the step is gated, the host regrows: gs_ctx_reserve).
// ---------------------------------------------------------------------------------------------
// DEPTH = false: the caller takes no depth image (gs_render_forward with out_depth NULL): no depth sum in the sweep (one of
// its ~30 vector instructions per splat), none in the checkpoints, nothing stored.
template <int SEG, bool DEPTH>
__global__ __launch_bounds__(256) void blend_fwd_v2q_kernel(
    int W, int H, int tileW, int tileH, int gridW, int blocksX, int nItems, int whiteBg,
    const float4* __restrict__ rec12, const uint32_t* __restrict__ sortedIdx, uint32_t idxMask,
    const uint32_t* __restrict__ tileRanges, const uint32_t* __restrict__ segBase, uint32_t segCap, int statePlanes,
    float* __restrict__ outColor, float* __restrict__ outDepth,
    float* __restrict__ outAlpha, uint32_t* __restrict__ lastContrib, float* __restrict__ finalT,
    float* __restrict__ segState, uint32_t* __restrict__ segSlot, uint32_t qslotCap, uint32_t qslotOwn, uint32_t qslotPart, uint32_t ckptPool,
    uint32_t* __restrict__ blockWork, uint32_t* __restrict__ counters, uint32_t* __restrict__ fwdQueue, uint32_t nq,
    const uint32_t* __restrict__ blockOrder, unsigned long long* __restrict__ trace,
    const uint32_t* __restrict__ cutStore, uint32_t* __restrict__ hostWords, uint32_t slowSlot, GsVirtGeom vg)
{
    static_assert(SEG % 64 == 0, "segment length must be a multiple of the 64-record chunk");
    // Round 6: the launch is workgroups of FOUR independent waves (no barrier between them, every wave its own LDS slots) instead
    // of one-wave workgroups.  The dispatcher spreads a workgroup's waves over the CU's four SIMDs, so every SIMD holds exactly
    // waves-per-SIMD of them; with 16 single-wave workgroups per CU it put five on some SIMDs and three on others (73 of 1024
    // each, tools/fwd_trace.py), and a fifth wave runs at a quarter of the first one's pace (below).  The four waves of a
    // workgroup take the four quadrants of one block as their first items -- the same records, now through one CU's L1.
    __shared__ f4 sgAll[4][2][192];      // per wave: two 64-record slots
    const int wvg = (int)(threadIdx.x >> 6);          // wave of the workgroup
    f4 (*sg)[192] = sgAll[wvg];
    // Round 6 (tools/fwd_trace.py, cycles per blended entry by HW_ID wave slot on c3: slot 0 282, slot 1 314, slot 2 402,
    // slot 3 582, the fifth wave of a SIMD that got five 994): the SIMD's issue arbiter favours its lower wave slots, a
    // wave's slot is fixed for its life, and the launch's last third was the waves of slot 3 finishing a SECOND item
    // (popped at ~60 % of the span, 130 k cycles at their pace) while the slots 0-2 had run out of work at 270 k of 346 k
    // cycles.  Three waves saturate a SIMD's issue (two do), so a wave in a slow slot takes its static first item and no
    // other: what it leaves in the queues the fast slots take, faster once it has gone.  Which wave blends which item
    // never changed a bit of the result (test_forward_queue_count_leaves_the_same_bits).
    const uint32_t hwSlot = (uint32_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11));      // HW_ID[3:0]: wave slot in the SIMD
    const int lane = threadIdx.x & 63;
    const uint32_t nWaves = gridDim.x * 4u;
    const uint32_t dw = blockIdx.x * 4u + (uint32_t)wvg;      // dense wave id (checkpoint pools, trace)
    // this wave's pool of checkpoint slots (wave-uniform): qslotOwn slots of the arena are its own from the start -- every
    // wave reaches its first boundary at about the same time, and that many pops on one counter would take ~6 ns each
    // to resolve (measured: blend forward 0.19 -> 0.30 ms with a shared counter only) -- and only a wave that uses them
    // up draws ckptPool (up to 8) more at a time from the shared part behind them, which is split in eight with a counter each
    // (workgroups go round-robin over the eight XCDs: a wave's counter lives in its own L2).  With ONE counter behind
    // a static share of 12 slots the 100 k / 800x800 config, whose waves need ~20, lost 54 us of its 175 (blend forward).
    uint32_t poolNext = dw * qslotOwn, poolEnd = poolNext + qslotOwn;
    const uint32_t part = blockIdx.x & 7u;
    uint32_t partsEmpty = 0;
    // Work distribution (round 4).  An item is one 8x8 quadrant of the pixel block at position p of the launch order
    // (blockOrder, deepest first).  The four quadrant waves of a block gather the same records; workgroups go to the eight
    // XCDs round-robin, each XCD with an L2 of its own, and rounds 1-3 gave the four items of a block to four consecutive
    // waves = four XCDs: every record was fetched from the fabric four times (PMC: 3.9x the algorithmic bytes).  Now block
    // position p belongs to XCD p mod 8: the wave with blockIdx.x = 8 s + x starts on quadrant s mod 4 of position
    // 8 (s / 4) + x, and the positions behind the statically assigned ones are handed out by EIGHT queues, one per XCD
    // (four consecutive pops = one block), so the quadrants of a block meet in one L2 -- and the pops, which resolve at
    // ~6 ns each on one address, spread over eight.  A wave whose XCD's queue has run dry takes from the others'.
    // nq = 8 queues (default) / 1 (rounds 1-3's mapping: position p = blockIdx.x / 4, one queue; GS_TUNE_FWD_QUEUES, A/B)
    // (workgroup g sits on XCD g mod nq; its wave w is "slot" 4 (g / nq) + w of that XCD: first item = quadrant w of position
    // nq (g / nq) + g mod nq)
    const uint32_t xcd = blockIdx.x % nq, slot = (blockIdx.x / nq) * 4u + (uint32_t)wvg;
    const uint32_t nPos = nq * ((((uint32_t)nItems >> 2) + nq - 1u) / nq);      // positions of the launch order: the pixel blocks,
                                                                                // padded to whole rows of nq (gs_bwd_prep.h, seg_base_body)
    const uint32_t staticRows = nWaves / (4u * nq);           // rows of nq positions covered by the waves' first items
    uint32_t dead = 0;                                        // queues found empty
    for (bool first = true;; first = false) {
        uint32_t pos = 0xFFFFFFFFu, quad = 0;
        if (first && (slot >> 2) < staticRows) { pos = nq * (slot >> 2) + xcd; quad = slot & 3u; }
        else if (hwSlot >= slowSlot) break;       // (a slow slot: no item beyond the static one)
        else {
            for (uint32_t t = 0; t < nq && pos == 0xFFFFFFFFu; t++) {
                const uint32_t y = (xcd + t) % nq;
                if ((dead >> y) & 1u) continue;
                // (no look before the pop: a load of the line the pops hammer queues behind them -- measured, one queue:
                // blend forward 0.19 -> 0.49 ms with a relaxed load in front of every atomicAdd)
                uint32_t k = 0;
                if (lane == 0) k = atomicAdd(&fwdQueue[y * 32u], 1u);
                k = __builtin_amdgcn_readfirstlane(k);
                const uint32_t p = nq * (staticRows + (k >> 2)) + y;
                if (p < nPos) { pos = p; quad = k & 3u; }
                else dead |= 1u << y;
            }
        }
        if (pos >= nPos) {
            if (first) continue;          // (a wave beyond the static rows, or an image smaller than the grid: try the queues)
            break;                        // every queue is empty; they only grow: every wave reaches this exit
        }
        const uint32_t item = pos * 4u + quad;
        const unsigned long long tStart = trace ? clock64() : 0ull;
        uint32_t itersDone = 0, chunksDone = 0;
        const uint32_t bRaw = __builtin_amdgcn_readfirstlane(blockOrder[pos]);
        if (bRaw == 0xFFFFFFFFu) continue;        // (a position a short last stripe leaves over)
        const int b = (int)bRaw;
        const int h = (int)((quad >> 1) & 1u), k = (int)(quad & 1u);
        const int by = b / blocksX, bx = b - by * blocksX;
        const int tile = ((by * BLK) / tileH) * gridW + (bx * BLK) / tileW;
        const uint32_t start = __builtin_amdgcn_readfirstlane(tileRanges[2 * tile]);
        const uint32_t end = __builtin_amdgcn_readfirstlane(tileRanges[2 * tile + 1]);
        const uint32_t count = end > start ? end - start : 0u;
        const uint32_t sbase = __builtin_amdgcn_readfirstlane(segBase[b]);

        int X0, Y0, XL, YL;       // the block's pixel origin and limits (block lists: blocks are enumerated per tile)
        gs_block_pixels(vg, bx, by, W, H, X0, Y0, XL, YL);
        const int x = X0 + k * 8 + (lane & 7), y = Y0 + h * 8 + (lane >> 3);
        const bool in = x < XL && y < YL;
        const float px = (float)x, py = (float)y;
        // the quadrant's pixel-centre rectangle, for the lane-private reach test at staging time
        const float qx0 = (float)(X0 + k * 8), qx1 = qx0 + 7.0f, qy0 = (float)(Y0 + h * 8), qy1 = qy0 + 7.0f;
        float T = in ? 1.0f : 0.0f;               // pixels outside the image start dead and are never stored
        // Round 6: the colour (depth) sums are kept as BASE + CHUNK -- the entries of the current 64-position chunk accumulate
        // from zero, and the chunk's sum is added to the base at the chunk's end (one rounding of the size of the total per
        // chunk instead of one per entry).  The backward takes what the rest of a list still owes from the difference between
        // the final image and a checkpoint (a base); deep in a list that difference is small against both, and its error was
        // the forward's per-entry roundings at the size of the TOTAL.  Same instruction count per entry.
        float cr = 0.f, cg = 0.f, cb = 0.f, dd = 0.f;          // this chunk's sums
        float br = 0.f, bgr = 0.f, bb = 0.f, bd = 0.f;         // the sums of the chunks in front of it
        uint32_t nc = 0;

        const uint32_t* __restrict__ idx = sortedIdx + start;
        auto save_state = [&](uint32_t i) {       // (called only while some pixel of the quadrant is live)
            if (poolNext == poolEnd) {
                // the wave's own eighth of the shared part first, then the others': since round 4 an XCD works on one stripe
                // of the image, and a deep stripe needs more slots than its eighth holds while a shallow one leaves its
                // own unused (grown bench scene: the arena "ran out" with most of it free).  A part found empty is not
                // asked again (every failed draw still advances its counter: gs_ctx_reserve sizes a regrow from their sum).
                poolNext = qslotCap;      // (all parts used up: positions beyond the arena, which the test below reports)
                for (uint32_t t = 0; t < 8u && poolNext == qslotCap; t++) {
                    const uint32_t y = (part + t) & 7u;
                    if ((partsEmpty >> y) & 1u) continue;
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&counters[GS_CNT_QSLOTS + y], ckptPool);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base + ckptPool <= qslotPart) poolNext = nWaves * qslotOwn + y * qslotPart + base;
                    else {      // (give the failed draw back: the counters' sum stays what was drawn + what was wanted and not had)
                        partsEmpty |= 1u << y;
                        if (lane == 0) atomicSub(&counters[GS_CNT_QSLOTS + y], ckptPool);
                    }
                }
                // nothing left anywhere: keep counting what would have been drawn (the size of the regrow)
                if (poolNext == qslotCap && partsEmpty == 0xFFu && lane == 0) atomicAdd(&counters[GS_CNT_QSLOTS + part], ckptPool);
                poolEnd = poolNext + ckptPool;
            }
            const uint32_t phys = poolNext++;
            const uint32_t vslot = sbase + i / SEG - 1;
            if (phys < qslotCap && vslot < segCap) {
                if (lane == 0) segSlot[(size_t)vslot * 4 + (h * 2 + k)] = phys;
                // a pixel that has terminated is never read back (the backward loads state only where nContrib > i0)
                float* st = segState + (size_t)phys * (statePlanes * 64) + lane;
                if (T >= 1e-4f) {
                    st[0] = T; st[64] = br; st[128] = bgr; st[192] = bb;
                    if (DEPTH && statePlanes == 5) st[256] = bd;   // wave-uniform: the depth sum only when a depth cotangent may come
                }
            } else if (lane == 0) {
                // out of checkpoint slots: the image is still complete, but no backward can be taken from this forward
                counters[GS_CNT_OVERFLOW] = 1u;
                hostWords[4] = 2u;        // 2: "the checkpoint arena", 1: "the pair reserve" (binning.hip)
            }
        };
        struct Pre {
            float aclamp, r, g, b, depth;
            uint32_t ncv;              // list position + 1 of the splat: nContrib of a pixel that is live when it arrives
        };
        // Staging compacts the chunk: lane j holds list entry c0 + j and tests it against the whole quadrant -- the
        // minimum of its quadratic form over the pixel rectangle (rect_min_q) -- and only entries that can reach a
        // pixel are parked in LDS, each with its list position.  An entry whose weight is below 2^-29 everywhere
        // moves no pixel's state by more than ~5e-8 (CULL_QMIN), so skipping it changes nothing but the work: no LDS
        // broadcast (the CU's LDS port is what saturates first with one pixel per lane), no exponent, no exp.
        auto stage_compact = [&](f4* slot, const RecV& v, uint32_t c0) -> uint32_t {
            const float qmin = rect_min_q(v.a.z, v.a.w, v.b.x, v.b.y, qx0 - v.a.x, qx1 - v.a.x, qy0 - v.a.y, qy1 - v.a.y);
            const bool keep = (c0 + lane < count) && !(qmin > CULL_QMIN);
            const unsigned long long m = __ballot(keep);
            const uint32_t pos = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (keep) {
                slot[pos * 3] = v.a; slot[pos * 3 + 1] = v.b;
                slot[pos * 3 + 2] = (f4){v.c.x, v.c.y, v.c.z, __uint_as_float(c0 + (uint32_t)lane + 1u)};
            }
            // pad to a multiple of four with null splats (opacity 0: alpha = 0, state untouched) so that the loop runs
            // whole groups only; a pixel still live at a pad entry is live at the end of the chunk, hence its nContrib
            const uint32_t n = (uint32_t)__popcll(m), n4 = (n + 3u) & ~3u;
            if ((uint32_t)lane < n4 - n) {
                const f4 z = (f4){0.f, 0.f, 0.f, 0.f};
                slot[(n + lane) * 3] = z; slot[(n + lane) * 3 + 1] = z;
                slot[(n + lane) * 3 + 2] = (f4){0.f, 0.f, 0.f, __uint_as_float(min(c0 + 64u, count))};
            }
            return n4;
        };
        auto pre = [&](const f4* slot, uint32_t j, Pre& o) {
            const Rec s = unpack(slot[j * 3], slot[j * 3 + 1], slot[j * 3 + 2]);
            const float dx = px - s.mx, dy = py - s.my;
            const float dxdy = dx * dy, dx2 = dx * dx, dy2 = dy * dy;
            const float q = ((dx2 * s.c00 + dy2 * s.c11) + dxdy * s.c01) + dxdy * s.c10;
#ifdef GS_FWD_LIBM_EXP   // experiment: the oracle's exp(-0.5 q) through the device math library instead of v_exp_f32
            o.aclamp = fminf(s.op * expf(-0.5f * q), 0.99f);
#else
            o.aclamp = fminf(gauss_alpha_raw(q, s.op), 0.99f);
#endif
            o.r = s.r; o.g = s.g; o.b = s.b; o.depth = s.depth;
            o.ncv = __float_as_uint(slot[j * 3 + 2].w);
        };
        auto post = [&](const Pre& o) {
            const bool a = T >= 1e-4f;
            nc = a ? o.ncv : nc;
            const float alpha = a ? o.aclamp : 0.0f;
            const float w = T * alpha;
#ifdef GS_FWD_UNFUSED   // experiment ("the 1e-4 bar"): the reference's two-rounding C + w c
            cr = cr + w * o.r; cg = cg + w * o.g; cb = cb + w * o.b; if (DEPTH) dd = dd + w * o.depth;
#else
            cr = fmaf(w, o.r, cr); cg = fmaf(w, o.g, cg); cb = fmaf(w, o.b, cb); if (DEPTH) dd = fmaf(w, o.depth, dd);
#endif
            T = T * (1.0f - alpha);
        };
        auto any_live = [&]() { return __any(T >= 1e-4f); };

        // The list runs through GS_FWD_PREFETCH register sets in turn: the set that holds chunk c's records is refilled, right after
        // they are staged, with the records of chunk c + D from indices loaded D chunks earlier, and then takes the indices of chunk
        // c + 2 D.  D = 1 (shipped): indices two chunks ahead, records one.  D = 2, 3 (round 6, tried because the last 15 % of the
        // launch's span move 1.4 % of its entries -- tools/fwd_trace.py -- as if every chunk waited for its gather): 114 / 128 VGPRs
        // against 105, blend forward on c3 0.179 / 0.181 ms against 0.178, c2 0.162 / 0.221 against 0.162 (EXPERIMENTS.md): the
        // tail is not the gather's latency.
        constexpr int D = GS_FWD_PREFETCH;
        RecV recs[D];
        uint32_t gis[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            recs[d] = load_chunk(rec12, idx, idxMask, 64u * d, count, lane);
            gis[d] = load_chunk_index(idx, idxMask, 64u * (D + d), count, lane);
        }
        auto chunk = [&](uint32_t c0, RecV& rec, uint32_t& gi) -> bool {      // false: the quadrant has no live pixel left
            f4* slot = sg[(c0 >> 6) & 1];
            const uint32_t n = stage_compact(slot, rec, c0);
            if (c0 + 64u * D < count) {
                rec = gather_chunk(rec12, gi);
                gi = load_chunk_index(idx, idxMask, c0 + 128u * D, count, lane);
            }
            br += cr; bgr += cg; bb += cb; cr = 0.f; cg = 0.f; cb = 0.f;          // the chunk behind us joins the base
            if (DEPTH) { bd += dd; dd = 0.f; }
            if (statePlanes != 0 && c0 != 0 && (c0 % SEG) == 0) save_state(c0);
            bool live = true;
            uint32_t j = 0;
            for (; j < n; j += 4) {      // n is a multiple of 4
                Pre p0, p1, p2, p3;
                pre(slot, j, p0); pre(slot, j + 1, p1); pre(slot, j + 2, p2); pre(slot, j + 3, p3);
                post(p0); post(p1); post(p2); post(p3);
                if (!any_live()) { live = false; break; }
            }
            itersDone += j; chunksDone++;
            if (!live) return false;
            if (T >= 1e-4f) nc = min(c0 + 64u, count);      // still live: went through the whole chunk
            return any_live();
        };
        for (uint32_t c0 = 0; c0 < count; c0 += 64u * D) {
            bool go = true;
#pragma unroll
            for (int d = 0; d < D; d++)
                if (go && c0 + 64u * d < count) go = chunk(c0 + 64u * d, recs[d], gis[d]);
            if (!go) break;
        }
        if (trace && lane == 0 && item < (uint32_t)nItems) {
            trace[(size_t)item * 4 + 0] = tStart;
            trace[(size_t)item * 4 + 1] = clock64();
            trace[(size_t)item * 4 + 2] = (unsigned long long)itersDone | ((unsigned long long)chunksDone << 32);      // blended entries (with pads) | chunks
            // blockIdx.x | XCC_ID (4 bits) << 32 | HW_ID[15:0] (wave, SIMD, pipe, CU, SH, SE) << 36
            trace[(size_t)item * 4 + 3] = (unsigned long long)dw |
                                          ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32) |
                                          ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (15 << 11)) << 36);
        }
        // depth cuts: live pixels at the end of a list that was cut short -- this forward has to be repeated without
        // cuts (pixels outside the image carry T = 0).  The flag word lives in host memory.
        if (cutStore && any_live() && lane == 0 && cutStore[tile] != 0u) hostWords[0] = 1u;
        if (in) {
            const size_t pix = (size_t)y * W + x;
            const float bg = whiteBg ? T : 0.0f;
            outColor[3 * pix] = (br + cr) + bg; outColor[3 * pix + 1] = (bgr + cg) + bg; outColor[3 * pix + 2] = (bb + cb) + bg;
            if (DEPTH) outDepth[pix] = bd + dd;
            outAlpha[pix] = 1.0f - T; lastContrib[pix] = nc; finalT[pix] = T;
        }
        uint32_t m = in ? nc : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
        if (lane == 0 && m) atomicMax(&blockWork[b], m);
    }
}

// ---------------------------------------------------------------------------------------------
// forward with a STAGING WAVE beside every sweeping wave (round 6; GS_TUNE_FWD_PAIR).
//
// What paces a deep quadrant (EXPERIMENTS.md, "where the forward's 31 % go"): its list keeps one entry in six, so a 64-position
// chunk is ~11 blended entries (~360 vector instructions) behind ~275 instructions that do not depend on the running state at
// all -- the record gather's address arithmetic, the reach test, the ballot / compaction into LDS, the pad.  In the one-wave
// kernel both sit on ONE wave's in-order instruction stream, and the launch ends when the deepest such chain does (c3: one
// quadrant's ~3000 positions; no launch order shortens a single item).  Here a workgroup is two waves: wave 1 stages --
// chunk c + 1 goes through load, cull and compaction into the other LDS slot while wave 0 blends chunk c, exactly the
// one-wave kernel's arithmetic in exactly its order (same image, same nContrib, same checkpoints, bit for bit) --, and the
// two meet at one workgroup barrier per chunk.  The chain per chunk is max(staging, blending) instead of their sum; the
// instruction total is the same, on six waves per SIMD (three pairs) instead of four.
// ---------------------------------------------------------------------------------------------
// a workgroup barrier that makes the LDS writes in front of it visible behind it, usable from wave-uniform branches
#define V2P_BARRIER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); \
                           __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
#ifndef GS_V2P_WAVES
#define GS_V2P_WAVES 6     // waves per SIMD the kernel is compiled for (the second __launch_bounds__ argument): <= 80 VGPRs
#endif
#define GS_V2P_WGS (2 * GS_V2P_WAVES)      // workgroups (pairs) per CU at that occupancy
template <int SEG, bool DEPTH>
__global__ __launch_bounds__(128, GS_V2P_WAVES) void blend_fwd_v2p_kernel(
    int W, int H, int tileW, int tileH, int gridW, int blocksX, int nItems, int whiteBg,
    const float4* __restrict__ rec12, const uint32_t* __restrict__ sortedIdx, uint32_t idxMask,
    const uint32_t* __restrict__ tileRanges, const uint32_t* __restrict__ segBase, uint32_t segCap, int statePlanes,
    float* __restrict__ outColor, float* __restrict__ outDepth,
    float* __restrict__ outAlpha, uint32_t* __restrict__ lastContrib, float* __restrict__ finalT,
    float* __restrict__ segState, uint32_t* __restrict__ segSlot, uint32_t qslotCap, uint32_t qslotOwn, uint32_t qslotPart, uint32_t ckptPool,
    uint32_t* __restrict__ blockWork, uint32_t* __restrict__ counters, uint32_t* __restrict__ fwdQueue, uint32_t nq,
    const uint32_t* __restrict__ blockOrder, const uint32_t* __restrict__ cutStore, uint32_t* __restrict__ hostWords, GsVirtGeom vg)
{
    static_assert(SEG % 64 == 0, "segment length must be a multiple of the 64-record chunk");
    __shared__ f4 sg[2][192];          // two 64-record slots: one being blended, one being staged
    __shared__ uint32_t sN[2];         // entries (padded to four) the staging wave left in each slot
    __shared__ uint32_t sItem[2];      // the workgroup's item: position of the launch order, quadrant
    __shared__ uint32_t sStop;         // set by the sweeping wave when the item is finished
    const int lane = threadIdx.x & 63;
    const bool stager = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) != 0;      // wave-uniform by construction
    uint32_t poolNext = blockIdx.x * qslotOwn, poolEnd = poolNext + qslotOwn;
    const uint32_t part = blockIdx.x & 7u;
    uint32_t partsEmpty = 0;
    const uint32_t xcd = blockIdx.x % nq, slot = blockIdx.x / nq;
    const uint32_t nPos = nq * ((((uint32_t)nItems >> 2) + nq - 1u) / nq);
    const uint32_t staticRows = gridDim.x / (4u * nq);
    uint32_t dead = 0;                                        // (wave 0) queues found empty
    for (bool first = true;; first = false) {
        if (!stager) {      // the item of the workgroup: as in blend_fwd_v2q_kernel, popped by the sweeping wave
            uint32_t pos = 0xFFFFFFFFu, quad = 0;
            if (first && (slot >> 2) < staticRows) { pos = nq * (slot >> 2) + xcd; quad = slot & 3u; }
            else {
                for (uint32_t t = 0; t < nq && pos == 0xFFFFFFFFu; t++) {
                    const uint32_t y = (xcd + t) % nq;
                    if ((dead >> y) & 1u) continue;
                    uint32_t k = 0;
                    if (lane == 0) k = atomicAdd(&fwdQueue[y * 32u], 1u);
                    k = __builtin_amdgcn_readfirstlane(k);
                    const uint32_t p = nq * (staticRows + (k >> 2)) + y;
                    if (p < nPos) { pos = p; quad = k & 3u; }
                    else dead |= 1u << y;
                }
            }
            if (lane == 0) { sItem[0] = pos; sItem[1] = quad; sStop = 0u; }
        }
        __syncthreads();
        const uint32_t pos = sItem[0], quad = sItem[1];
        if (pos >= nPos) {
            __syncthreads();              // (sItem is written again by the next round's pop)
            if (first) continue;
            break;
        }
        const uint32_t bRaw = __builtin_amdgcn_readfirstlane(blockOrder[pos]);
        const int b = (int)bRaw;
        const int h = (int)((quad >> 1) & 1u), k = (int)(quad & 1u);
        uint32_t start = 0, count = 0, sbase = 0;
        int tile = 0, X0 = 0, Y0 = 0, XL = 0, YL = 0;
        if (bRaw != 0xFFFFFFFFu) {
            const int by = b / blocksX, bx = b - by * blocksX;
            tile = ((by * BLK) / tileH) * gridW + (bx * BLK) / tileW;
            start = __builtin_amdgcn_readfirstlane(tileRanges[2 * tile]);
            const uint32_t end = __builtin_amdgcn_readfirstlane(tileRanges[2 * tile + 1]);
            count = end > start ? end - start : 0u;
            sbase = __builtin_amdgcn_readfirstlane(segBase[b]);
            gs_block_pixels(vg, bx, by, W, H, X0, Y0, XL, YL);
        }
        const int x = X0 + k * 8 + (lane & 7), y = Y0 + h * 8 + (lane >> 3);
        const bool in = bRaw != 0xFFFFFFFFu && x < XL && y < YL;
        const float px = (float)x, py = (float)y;
        const float qx0 = (float)(X0 + k * 8), qx1 = qx0 + 7.0f, qy0 = (float)(Y0 + h * 8), qy1 = qy0 + 7.0f;
        const uint32_t* __restrict__ idx = sortedIdx + start;

        // ---- the staging wave's half: load, reach test, compaction (blend_fwd_v2q_kernel, stage_compact) ----
        auto stage_compact = [&](f4* sl, const RecV& v, uint32_t c0) -> uint32_t {
            const float qmin = rect_min_q(v.a.z, v.a.w, v.b.x, v.b.y, qx0 - v.a.x, qx1 - v.a.x, qy0 - v.a.y, qy1 - v.a.y);
            const bool keep = (c0 + lane < count) && !(qmin > CULL_QMIN);
            const unsigned long long m = __ballot(keep);
            const uint32_t ps = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (keep) {
                sl[ps * 3] = v.a; sl[ps * 3 + 1] = v.b;
                sl[ps * 3 + 2] = (f4){v.c.x, v.c.y, v.c.z, __uint_as_float(c0 + (uint32_t)lane + 1u)};
            }
            const uint32_t n = (uint32_t)__popcll(m), n4 = (n + 3u) & ~3u;
            if ((uint32_t)lane < n4 - n) {
                const f4 z = (f4){0.f, 0.f, 0.f, 0.f};
                sl[(n + lane) * 3] = z; sl[(n + lane) * 3 + 1] = z;
                sl[(n + lane) * 3 + 2] = (f4){0.f, 0.f, 0.f, __uint_as_float(min(c0 + 64u, count))};
            }
            return n4;
        };
        // The two roles run SEPARATE loops with the same number of barriers (a barrier counts arriving waves, not program
        // locations; `stager` is wave-uniform): in one shared loop the staging wave's prefetch registers and the sweeping wave's
        // pixel state were all loop-carried together (100 VGPRs: four waves per SIMD instead of six).
        if (stager) {
            RecV nxt;
            nxt.a = nxt.b = nxt.c = (f4){0.f, 0.f, 0.f, 0.f};
            uint32_t gNext = 0xFFFFFFFFu;
            if (count > 0) {          // chunk 0 into slot 0 before the sweep starts; chunk 1's records and chunk 2's indices in flight
                gNext = load_chunk_index(idx, idxMask, 64, count, lane);
                nxt = load_chunk(rec12, idx, idxMask, 0, count, lane);
                const uint32_t n = stage_compact(sg[0], nxt, 0);
                if (lane == 0) sN[0] = n;
                if (64 < count) {
                    nxt = gather_chunk(rec12, gNext);
                    gNext = load_chunk_index(idx, idxMask, 128, count, lane);
                }
            }
            V2P_BARRIER();                // slot 0 is staged (and sItem has been read by both waves)
            for (uint32_t c0 = 0; c0 < count; c0 += 64) {
                const uint32_t cur = (c0 >> 6) & 1u;
                if (c0 + 64 < count) {    // the next chunk into the other slot while the sweeping wave blends this one
                    const uint32_t n = stage_compact(sg[cur ^ 1u], nxt, c0 + 64);
                    if (lane == 0) sN[cur ^ 1u] = n;
                    if (c0 + 128 < count) {
                        nxt = gather_chunk(rec12, gNext);
                        gNext = load_chunk_index(idx, idxMask, c0 + 192, count, lane);
                    }
                }
                V2P_BARRIER();
                if (*(volatile uint32_t*)&sStop) break;          // (both waves read the same word behind the same barrier)
            }
            V2P_BARRIER();                // (sStop / sItem / the slots are written again by the next item)
            continue;
        }
        // ---- the sweeping wave's half: the running state and everything that depends on it ----
        float T = in ? 1.0f : 0.0f;
        float cr = 0.f, cg = 0.f, cb = 0.f, dd = 0.f;          // this chunk's sums; the chunks in front of it (blend_fwd_v2q_kernel):
        float br = 0.f, bgr = 0.f, bb = 0.f, bd = 0.f;
        uint32_t nc = 0;
        auto save_state = [&](uint32_t i) {       // (blend_fwd_v2q_kernel, save_state)
            if (poolNext == poolEnd) {
                poolNext = qslotCap;
                for (uint32_t t = 0; t < 8u && poolNext == qslotCap; t++) {
                    const uint32_t yy = (part + t) & 7u;
                    if ((partsEmpty >> yy) & 1u) continue;
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&counters[GS_CNT_QSLOTS + yy], ckptPool);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base + ckptPool <= qslotPart) poolNext = gridDim.x * qslotOwn + yy * qslotPart + base;
                    else {
                        partsEmpty |= 1u << yy;
                        if (lane == 0) atomicSub(&counters[GS_CNT_QSLOTS + yy], ckptPool);
                    }
                }
                if (poolNext == qslotCap && partsEmpty == 0xFFu && lane == 0) atomicAdd(&counters[GS_CNT_QSLOTS + part], ckptPool);
                poolEnd = poolNext + ckptPool;
            }
            const uint32_t phys = poolNext++;
            const uint32_t vslot = sbase + i / SEG - 1;
            if (phys < qslotCap && vslot < segCap) {
                if (lane == 0) segSlot[(size_t)vslot * 4 + (h * 2 + k)] = phys;
                float* st = segState + (size_t)phys * (statePlanes * 64) + lane;
                if (T >= 1e-4f) {
                    st[0] = T; st[64] = br; st[128] = bgr; st[192] = bb;
                    if (DEPTH && statePlanes == 5) st[256] = bd;
                }
            } else if (lane == 0) {
                counters[GS_CNT_OVERFLOW] = 1u;
                hostWords[4] = 2u;
            }
        };
        struct Pre {
            float aclamp, r, g, b, depth;
            uint32_t ncv;
        };
        auto pre = [&](const f4* sl, uint32_t j, Pre& o) {
            const Rec s = unpack(sl[j * 3], sl[j * 3 + 1], sl[j * 3 + 2]);
            const float dx = px - s.mx, dy = py - s.my;
            const float dxdy = dx * dy, dx2 = dx * dx, dy2 = dy * dy;
            const float q = ((dx2 * s.c00 + dy2 * s.c11) + dxdy * s.c01) + dxdy * s.c10;
            o.aclamp = fminf(gauss_alpha_raw(q, s.op), 0.99f);
            o.r = s.r; o.g = s.g; o.b = s.b; o.depth = s.depth;
            o.ncv = __float_as_uint(sl[j * 3 + 2].w);
        };
        auto post = [&](const Pre& o) {
            const bool a = T >= 1e-4f;
            nc = a ? o.ncv : nc;
            const float alpha = a ? o.aclamp : 0.0f;
            const float w = T * alpha;
            cr = fmaf(w, o.r, cr); cg = fmaf(w, o.g, cg); cb = fmaf(w, o.b, cb); if (DEPTH) dd = fmaf(w, o.depth, dd);
            T = T * (1.0f - alpha);
        };
        auto any_live = [&]() { return __any(T >= 1e-4f); };

        V2P_BARRIER();                    // slot 0 is staged
        for (uint32_t c0 = 0; c0 < count; c0 += 64) {
            const uint32_t cur = (c0 >> 6) & 1u;
            const f4* sl = sg[cur];
            const uint32_t n = __builtin_amdgcn_readfirstlane(*(volatile uint32_t*)&sN[cur]);
            br += cr; bgr += cg; bb += cb; cr = 0.f; cg = 0.f; cb = 0.f;          // the chunk behind us joins the base
            if (DEPTH) { bd += dd; dd = 0.f; }
            if (statePlanes != 0 && c0 != 0 && (c0 % SEG) == 0) save_state(c0);
            bool live = true;
            for (uint32_t j = 0; j < n; j += 4) {      // n is a multiple of 4
                Pre p0, p1, p2, p3;
                pre(sl, j, p0); pre(sl, j + 1, p1); pre(sl, j + 2, p2); pre(sl, j + 3, p3);
                post(p0); post(p1); post(p2); post(p3);
                if (!any_live()) { live = false; break; }
            }
            if (live) {
                if (T >= 1e-4f) nc = min(c0 + 64u, count);      // still live: went through the whole chunk
                live = any_live();
            }
            if (!live && lane == 0) sStop = 1u;
            V2P_BARRIER();
            if (*(volatile uint32_t*)&sStop) break;
        }
        V2P_BARRIER();                    // (sStop / sItem / the slots are written again by the next item)
        if (bRaw != 0xFFFFFFFFu) {
            if (cutStore && any_live() && lane == 0 && cutStore[tile] != 0u) hostWords[0] = 1u;
            if (in) {
                const size_t pix = (size_t)y * W + x;
                const float bg = whiteBg ? T : 0.0f;
                outColor[3 * pix] = (br + cr) + bg; outColor[3 * pix + 1] = (bgr + cg) + bg; outColor[3 * pix + 2] = (bb + cb) + bg;
                if (DEPTH) outDepth[pix] = bd + dd;
                outAlpha[pix] = 1.0f - T; lastContrib[pix] = nc; finalT[pix] = T;
            }
            uint32_t m = in ? nc : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
            if (lane == 0 && m) atomicMax(&blockWork[b], m);
        }
    }
}
#undef V2P_BARRIER

// ---------------------------------------------------------------------------------------------
// forward for launches with FEWER ITEMS THAN WAVE SLOTS (small images): four waves per quadrant, along the list.
//
// What the trace of the 10 k / 400x400 scene says (tools/fwd_trace.py, FWD_TRACE_CONFIG=c1_10k_400): a gfx950 wave issues a
// vector instruction every ~8 - 9 cycles at best -- one entry per ~280 cycles alone on its SIMD, 236 beside one other wave,
// and only from two waves on does the SIMD issue every ~4 cycles --, the 2500 quadrant items are two or three to a SIMD, and
// the kernel's 173 us are its deepest item's 1660 entries x 236 cycles while the chip-wide floor is 114.  More independent
// work INSIDE a wave does not help (two streams per wave: measured slower, EXPERIMENTS.md); more waves do.  Here a workgroup
// of four waves (one per SIMD of its CU) owns the quadrant.  The list goes by rounds of four 64-entry chunks, one per wave
// (which wave takes which turns with the round: a list's first chunk is the one every item has): part 0 from the
// quadrant's running state S (exact, the one-wave kernel's trips), parts 1-3 from T = 1, C = 0.  The end states meet in LDS and every wave folds
// them:  C += T_p C_w,  T *= T_w  for a pixel that is live from the part's first entry to its last (T only falls, so
// T_p T_w >= 1e-4 at the end says so); a pixel that is dead at the part's start keeps its state; a pixel that CROSSES 1e-4
// inside part w has sums that hold entries the reference does not blend -- wave w takes its (still staged) chunk again from
// the folded prefix, exactly as the one-wave kernel would have, and publishes that pixel's state; the fold then goes on
// behind part w from THAT state (round 5: a pixel on the threshold can come back live, the composed and the sequential
// product round differently; it is folded -- and re-swept, if it crosses again -- through the parts behind).  A part's
// checkpoint is its prefix, saved once the fold has settled it for every pixel.  The part boundaries depend on the list position only: hinted / unhinted / cut / uncut forwards of a view stay
// the same bits, and a list of at most 64 entries is swept by one wave with the one-wave kernel's arithmetic.
// ---------------------------------------------------------------------------------------------
// workgroups (= waves per SIMD) a CU holds: 102 VGPRs, 23 KB of LDS each.  10 k / 400x400 scene, blend forward: 0.137 ms with
// four, 0.130 with five (six does not launch); two staging slots per wave and 36 KB: four at most
#define GS_V2W_WGS 5
template <int SEG, bool DEPTH>
__global__ __launch_bounds__(256, GS_V2W_WGS) void blend_fwd_v2w_kernel(
    int W, int H, int tileW, int tileH, int gridW, int blocksX, int nItems, int whiteBg,
    const float4* __restrict__ rec12, const uint32_t* __restrict__ sortedIdx, uint32_t idxMask,
    const uint32_t* __restrict__ tileRanges, const uint32_t* __restrict__ segBase, uint32_t segCap, int statePlanes,
    float* __restrict__ outColor, float* __restrict__ outDepth,
    float* __restrict__ outAlpha, uint32_t* __restrict__ lastContrib, float* __restrict__ finalT,
    float* __restrict__ segState, uint32_t* __restrict__ segSlot, uint32_t qslotCap, uint32_t qslotOwn, uint32_t qslotPart, uint32_t ckptPool,
    uint32_t* __restrict__ blockWork, uint32_t* __restrict__ counters, uint32_t* __restrict__ fwdQueue, uint32_t nq,
    const uint32_t* __restrict__ blockOrder, const uint32_t* __restrict__ cutStore, uint32_t* __restrict__ hostWords, float foldScale,
    GsVirtGeom vg)
{
    static_assert(SEG == 64, "a part is one 64-entry chunk = one segment: its only checkpoint is its prefix");
    __shared__ f4 sgAll[4][192];               // per wave: one 64-record slot (DS operations of a wave complete in order)
    __shared__ float xEnd[4][5][64];           // end state of every part (wave 0: absolute, waves 1-3: relative)
    __shared__ float xDead[4][6][64];          // state (and nContrib) of the pixels that finished inside a part swept again
    __shared__ float xPre[4][4][64];           // per wave: the colour (depth) sums in front of its own part = its checkpoint
    __shared__ uint32_t sPop[2];
    const int lane = threadIdx.x & 63, hw = threadIdx.x >> 6;
    f4* sg = sgAll[hw];
    uint32_t poolNext = (blockIdx.x * 4u + (uint32_t)hw) * qslotOwn, poolEnd = poolNext + qslotOwn;
    const uint32_t part = blockIdx.x & 7u;
    uint32_t partsEmpty = 0;
    const uint32_t xcd = blockIdx.x % nq, slot = blockIdx.x / nq;
    const uint32_t nPos = nq * ((((uint32_t)nItems >> 2) + nq - 1u) / nq);
    const uint32_t staticRows = gridDim.x / (4u * nq);
    uint32_t dead = 0;                                        // (thread 0) queues found empty
    for (bool first = true;; first = false) {
        // the item of the workgroup: as in blend_fwd_v2q_kernel, popped by one thread
        if (threadIdx.x == 0) {
            uint32_t pos = 0xFFFFFFFFu, quad = 0;
            if (first && (slot >> 2) < staticRows) { pos = nq * (slot >> 2) + xcd; quad = slot & 3u; }
            else {
                for (uint32_t t = 0; t < nq && pos == 0xFFFFFFFFu; t++) {
                    const uint32_t y = (xcd + t) % nq;
                    if ((dead >> y) & 1u) continue;
                    const uint32_t k = atomicAdd(&fwdQueue[y * 32u], 1u);
                    const uint32_t p = nq * (staticRows + (k >> 2)) + y;
                    if (p < nPos) { pos = p; quad = k & 3u; }
                    else dead |= 1u << y;
                }
            }
            sPop[0] = pos; sPop[1] = quad;
        }
        __syncthreads();
        const uint32_t pos = sPop[0], quad = sPop[1];
        __syncthreads();
        if (pos >= nPos) {
            if (first) continue;
            break;
        }
        const uint32_t bRaw = __builtin_amdgcn_readfirstlane(blockOrder[pos]);
        if (bRaw == 0xFFFFFFFFu) continue;
        const int b = (int)bRaw;
        const int h = (int)((quad >> 1) & 1u), k = (int)(quad & 1u);
        const int by = b / blocksX, bx = b - by * blocksX;
        const int tile = ((by * BLK) / tileH) * gridW + (bx * BLK) / tileW;
        const uint32_t start = __builtin_amdgcn_readfirstlane(tileRanges[2 * tile]);
        const uint32_t end = __builtin_amdgcn_readfirstlane(tileRanges[2 * tile + 1]);
        const uint32_t count = end > start ? end - start : 0u;
        const uint32_t sbase = __builtin_amdgcn_readfirstlane(segBase[b]);

        int X0, Y0, XL, YL;
        gs_block_pixels(vg, bx, by, W, H, X0, Y0, XL, YL);
        const int x = X0 + k * 8 + (lane & 7), y = Y0 + h * 8 + (lane >> 3);
        const bool in = x < XL && y < YL;
        const float px = (float)x, py = (float)y;
        const float qx0 = (float)(X0 + k * 8), qx1 = qx0 + 7.0f, qy0 = (float)(Y0 + h * 8), qy1 = qy0 + 7.0f;
        // the quadrant's running state S: the same bits in all four waves
        float T = in ? 1.0f : 0.0f;
        float cr = 0.f, cg = 0.f, cb = 0.f, dd = 0.f;
        uint32_t nc = 0;

        const uint32_t* __restrict__ idx = sortedIdx + start;
        // (sT .. sd: the state before splat i, absolute; called only while some pixel of the quadrant is live there)
        auto save_state_vals = [&](uint32_t i, float sT, float sr, float sgr, float sb, float sd) {
            if (poolNext == poolEnd) {
                poolNext = qslotCap;
                for (uint32_t t = 0; t < 8u && poolNext == qslotCap; t++) {
                    const uint32_t yy = (part + t) & 7u;
                    if ((partsEmpty >> yy) & 1u) continue;
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&counters[GS_CNT_QSLOTS + yy], ckptPool);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base + ckptPool <= qslotPart) poolNext = gridDim.x * 4u * qslotOwn + yy * qslotPart + base;
                    else {
                        partsEmpty |= 1u << yy;
                        if (lane == 0) atomicSub(&counters[GS_CNT_QSLOTS + yy], ckptPool);
                    }
                }
                if (poolNext == qslotCap && partsEmpty == 0xFFu && lane == 0) atomicAdd(&counters[GS_CNT_QSLOTS + part], ckptPool);
                poolEnd = poolNext + ckptPool;
            }
            const uint32_t phys = poolNext++;
            const uint32_t vslot = sbase + i / SEG - 1;
            if (phys < qslotCap && vslot < segCap) {
                if (lane == 0) segSlot[(size_t)vslot * 4 + (h * 2 + k)] = phys;
                float* st = segState + (size_t)phys * (statePlanes * 64) + lane;
                if (sT >= 1e-4f) {
                    st[0] = sT; st[64] = sr; st[128] = sgr; st[192] = sb;
                    if (DEPTH && statePlanes == 5) st[256] = sd;
                }
            } else if (lane == 0) {
                counters[GS_CNT_OVERFLOW] = 1u;
                hostWords[4] = 2u;
            }
        };
        auto save_state = [&](uint32_t i) { save_state_vals(i, T, cr, cg, cb, dd); };
        struct Pre {
            float aclamp, r, g, b, depth;
            uint32_t ncv;
        };
        auto stage_compact = [&](f4* sl, const RecV& v, uint32_t c0) -> uint32_t {
            const float qmin = rect_min_q(v.a.z, v.a.w, v.b.x, v.b.y, qx0 - v.a.x, qx1 - v.a.x, qy0 - v.a.y, qy1 - v.a.y);
            const bool keep = (c0 + lane < count) && !(qmin > CULL_QMIN);
            const unsigned long long m = __ballot(keep);
            const uint32_t ps = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (keep) {
                sl[ps * 3] = v.a; sl[ps * 3 + 1] = v.b;
                sl[ps * 3 + 2] = (f4){v.c.x, v.c.y, v.c.z, __uint_as_float(c0 + (uint32_t)lane + 1u)};
            }
            const uint32_t n = (uint32_t)__popcll(m), n4 = (n + 3u) & ~3u;
            if ((uint32_t)lane < n4 - n) {
                const f4 z = (f4){0.f, 0.f, 0.f, 0.f};
                sl[(n + lane) * 3] = z; sl[(n + lane) * 3 + 1] = z;
                sl[(n + lane) * 3 + 2] = (f4){0.f, 0.f, 0.f, __uint_as_float(min(c0 + 64u, count))};
            }
            return n4;
        };
        auto pre = [&](const f4* sl, uint32_t j, Pre& o) {
            const Rec s = unpack(sl[j * 3], sl[j * 3 + 1], sl[j * 3 + 2]);
            const float dx = px - s.mx, dy = py - s.my;
            const float dxdy = dx * dy, dx2 = dx * dx, dy2 = dy * dy;
            const float q = ((dx2 * s.c00 + dy2 * s.c11) + dxdy * s.c01) + dxdy * s.c10;
            o.aclamp = fminf(gauss_alpha_raw(q, s.op), 0.99f);
            o.r = s.r; o.g = s.g; o.b = s.b; o.depth = s.depth;
            o.ncv = __float_as_uint(sl[j * 3 + 2].w);
        };
        auto post = [&](const Pre& o) {
            const bool a = T >= 1e-4f;
            nc = a ? o.ncv : nc;
            const float alpha = a ? o.aclamp : 0.0f;
            const float wgt = T * alpha;
            cr = fmaf(wgt, o.r, cr); cg = fmaf(wgt, o.g, cg); cb = fmaf(wgt, o.b, cb); if (DEPTH) dd = fmaf(wgt, o.depth, dd);
            T = T * (1.0f - alpha);
        };
        auto any_live = [&]() { return __any(T >= 1e-4f); };
        // the staged chunk's entries against the running state, as the one-wave kernel takes them (liveness gate, nContrib)
        // (round 6, as in blend_fwd_v2q_kernel: the chunk's entries accumulate from zero and join the state the chunk started
        // from at its end -- one rounding at the size of the total per chunk, not one per entry)
        // (the state the chunk starts from waits in LDS meanwhile -- `park`, four rows of 64 this wave owns at that moment --, not
        // in four registers across the loop: the kernel sits at its register budget for five workgroups per CU)
        auto trips_abs = [&](const f4* sl, uint32_t n, uint32_t c0, float* park) {
            park[lane] = cr; park[64 + lane] = cg; park[128 + lane] = cb; if (DEPTH) park[192 + lane] = dd;
            cr = 0.f; cg = 0.f; cb = 0.f; dd = 0.f;
            bool live = true;
            for (uint32_t j = 0; j < n && live; j += 4) {
                Pre p0, p1, p2, p3;
                pre(sl, j, p0); pre(sl, j + 1, p1); pre(sl, j + 2, p2); pre(sl, j + 3, p3);
                post(p0); post(p1); post(p2); post(p3);
                live = any_live();
            }
            if (live && T >= 1e-4f) nc = min(c0 + 64u, count);     // still live: went through the whole chunk
            cr = park[lane] + cr; cg = park[64 + lane] + cg; cb = park[128 + lane] + cb; if (DEPTH) dd = park[192 + lane] + dd;
        };

        // chunk of this wave in round r: c(r) = 256 r + 64 ((hw + r) & 3); records run one round ahead, indices two
        auto chunk_of = [&](uint32_t r) { return 256u * r + 64u * (((uint32_t)hw + r) & 3u); };
        RecV nx = load_chunk(rec12, idx, idxMask, chunk_of(0), count, lane);
        uint32_t gN = load_chunk_index(idx, idxMask, chunk_of(1), count, lane);
        for (uint32_t r = 0; 256u * r < count; r++) {
            if (!any_live()) break;                                // (S is the same in every wave: a uniform exit)
            const uint32_t s0 = 256u * r;
            const int w = (int)(((uint32_t)hw + r) & 3u);          // this wave's part of the round
            const uint32_t c0 = s0 + 64u * (uint32_t)w;
            const uint32_t nParts = min(4u, (count - s0 + 63u) / 64u);
            const bool mine = (uint32_t)w < nParts;
            f4* sl = sg;
            uint32_t n = 0;
            if (mine) n = stage_compact(sl, nx, c0);
            nx = gather_chunk(rec12, gN);                          // (0xFFFFFFFF beyond the list: nothing is loaded)
            gN = load_chunk_index(idx, idxMask, chunk_of(r + 2), count, lane);
            // (parts 1-3 are swept from T = 1; the running state comes back from the fold, where part 0's end state is absolute)
            if (w == 0) {
                if (statePlanes != 0 && c0 != 0) save_state(c0);
                trips_abs(sl, n, c0, &xPre[hw][0][0]);          // (wave 0's xPre rows are written in the fold below, behind this)
            } else if (mine) {
                T = in ? 1.0f : 0.0f; cr = 0.f; cg = 0.f; cb = 0.f; dd = 0.f;
                for (uint32_t j = 0; j < n; j += 4) {              // (no liveness gate: the sums are used only where every entry was live)
                    Pre p0, p1, p2, p3;
                    pre(sl, j, p0); pre(sl, j + 1, p1); pre(sl, j + 2, p2); pre(sl, j + 3, p3);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const Pre& o = u == 0 ? p0 : u == 1 ? p1 : u == 2 ? p2 : p3;
                        const float wgt = T * o.aclamp;
                        cr = fmaf(wgt, o.r, cr); cg = fmaf(wgt, o.g, cg); cb = fmaf(wgt, o.b, cb); if (DEPTH) dd = fmaf(wgt, o.depth, dd);
                        T = T * (1.0f - o.aclamp);
                    }
                }
            }
            if (mine) { xEnd[w][0][lane] = T; xEnd[w][1][lane] = cr; xEnd[w][2][lane] = cg; xEnd[w][3][lane] = cb; xEnd[w][4][lane] = dd; }
            if (w == 0) xDead[0][5][lane] = __uint_as_float(nc);   // (nContrib after part 0)
            __syncthreads();
            // Fold, for this lane's pixel: P = the absolute state behind the parts [0, qDone) of the round.  A part q the pixel
            // is live through (T only falls: P.T x its end T >= 1e-4 says so) is composed onto P; a pixel dead at the part's
            // start keeps its state; a pixel that CROSSES 1e-4 inside part q (crossedIn = q) stops folding there, wave q takes
            // its chunk again from P with the one-wave kernel's gate and publishes that pixel's state, and the fold goes on
            // behind part q FROM THAT STATE.  The composed product and the sequential one round differently: a pixel within
            // ~1e-6 of the threshold can come back from its second take still live (round 4 read it as finished for the
            // round's later parts while the next round blended on: entries skipped, checkpoint lanes the backward reads
            // never written -- the advisor's finding).  So the fold is a loop: such a pixel (`again`) is folded through the
            // parts behind from its re-swept state, crossing and being re-swept again if need be; every other pixel's fold
            // is done in the first pass, and the loop's last vote is the barrier the round needed anyway.
            float PT = xEnd[0][0][lane], Pr = xEnd[0][1][lane], Pg = xEnd[0][2][lane], Pb = xEnd[0][3][lane], Pd = xEnd[0][4][lane];
            uint32_t Pnc = __float_as_uint(xDead[0][5][lane]);
            // prefix of this wave's own part (its checkpoint): known once the fold has passed the parts in front of it; a
            // pixel that never gets there live is not stored (T = 0)
            // (parked in LDS, not in five registers across the loop: the kernel sits at its register budget for five
            // workgroups per CU, and the loop's live state put 8 - 11 VGPRs into scratch)
            float myPT = 0.0f;
            uint32_t qDone = 1;
            for (;;) {
                int crossedIn = 0;
#pragma unroll
                for (int q = 1; q < 4; q++) {
                    if ((uint32_t)q >= qDone && (uint32_t)q < nParts && !crossedIn) {
                        if (q == w) { myPT = PT; xPre[hw][0][lane] = Pr; xPre[hw][1][lane] = Pg; xPre[hw][2][lane] = Pb; if (DEPTH) xPre[hw][3][lane] = Pd; }
                        if (PT >= 1e-4f) {
                            const float Tend = PT * xEnd[q][0][lane];
                            if (Tend * foldScale >= 1e-4f) {       // (foldScale: 1 exactly, but for the tests' forced second takes)
                                Pr = fmaf(PT, xEnd[q][1][lane], Pr); Pg = fmaf(PT, xEnd[q][2][lane], Pg); Pb = fmaf(PT, xEnd[q][3][lane], Pb);
                                if (DEPTH) Pd = fmaf(PT, xEnd[q][4][lane], Pd);
                                PT = Tend;
                                Pnc = min(s0 + 64u * (uint32_t)(q + 1), count);
                            } else crossedIn = q;
                        }
                        if (!crossedIn) qDone = (uint32_t)q + 1u;
                    }
                }
                if (!__syncthreads_or(crossedIn != 0)) break;
                if (w != 0 && mine && __any(crossedIn == w)) {
                    // pixels finish inside this part: its entries again, in sequence from the folded prefix (the chunk is
                    // still staged); the other lanes idle with T = 0
                    T = crossedIn == w ? PT : 0.0f; cr = Pr; cg = Pg; cb = Pb; dd = Pd; nc = 0;
                    trips_abs(sl, n, c0, &xDead[w][1][0]);          // (this part's xDead rows are written right below, by this wave)
                    if (crossedIn == w) {
                        xDead[w][0][lane] = T; xDead[w][1][lane] = cr; xDead[w][2][lane] = cg; xDead[w][3][lane] = cb; xDead[w][4][lane] = dd;
                        xDead[w][5][lane] = __uint_as_float(nc);
                    }
                }
                __syncthreads();
                bool again = false;
                if (crossedIn) {
                    PT = xDead[crossedIn][0][lane]; Pr = xDead[crossedIn][1][lane]; Pg = xDead[crossedIn][2][lane];
                    Pb = xDead[crossedIn][3][lane]; Pd = xDead[crossedIn][4][lane]; Pnc = __float_as_uint(xDead[crossedIn][5][lane]);
                    qDone = (uint32_t)crossedIn + 1u;
                    again = PT >= 1e-4f && qDone < nParts;         // came back live with parts still in front of it
                    if (!again) qDone = 4u;                        // finished (or nothing left): later parts leave it alone
                }
                if (!__syncthreads_or(again)) break;               // (also: xDead / xEnd are written again in the next round)
            }
            // the part's checkpoint, now that every pixel's prefix of it is final (the one-wave kernel saves one iff some
            // pixel reaches the chunk live)
            if (statePlanes != 0 && w != 0 && mine && __any(myPT >= 1e-4f))
                save_state_vals(c0, myPT, xPre[hw][0][lane], xPre[hw][1][lane], xPre[hw][2][lane], DEPTH ? xPre[hw][3][lane] : 0.0f);
            T = PT; cr = Pr; cg = Pg; cb = Pb; dd = Pd; nc = Pnc;
        }
        if (hw == 0) {
            if (cutStore && any_live() && lane == 0 && cutStore[tile] != 0u) hostWords[0] = 1u;
            // (the pixel is rebuilt from the lane here instead of being kept across the rounds: the kernel spilled its item-invariant
            // values to scratch at five workgroups per CU)
            const int xe = X0 + k * 8 + (lane & 7), ye = Y0 + h * 8 + (lane >> 3);
            if (xe < XL && ye < YL) {
                const size_t pix = (size_t)ye * W + xe;
                const float bg = whiteBg ? T : 0.0f;
                outColor[3 * pix] = cr + bg; outColor[3 * pix + 1] = cg + bg; outColor[3 * pix + 2] = cb + bg;
                if (DEPTH) outDepth[pix] = dd;
                outAlpha[pix] = 1.0f - T; lastContrib[pix] = nc; finalT[pix] = T;
            }
            uint32_t m = (xe < XL && ye < YL) ? nc : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
            if (lane == 0 && m) atomicMax(&blockWork[b], m);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// (wave_sum10_transposed: gs_wavesum.h)

// A list entry as the backward keeps it in LDS: the conic pre-multiplied by C = -1/(2 ln 2), so that the exponent of
// 2^e comes out of two fused multiply-adds, e = dx^2 A + dxdy B + dy^2 Dq (A = c00 C, B = (c01 + c10) C, Dq = c11 C).
// The FORWARD has to evaluate the quadratic form in the reference's operation order to stay within 1e-4 of it
// (pair_exponent, gauss_alpha_raw); the backward only needs alpha to ~1e-6 (gradients are held to 1e-3), and this
// form is four instructions per 16x8 half and splat shorter.
struct RecB {
    float mx, my, A, B, Dq, r, g, b, op, depth;
};
__device__ __forceinline__ RecB unpack_b(const f4 a, const f4 b, const f4 c)
{
    RecB r;
    r.mx = a.x; r.my = a.y; r.A = a.z; r.B = a.w;
    r.Dq = b.x; r.r = b.z; r.g = b.w;
    r.b = c.x; r.op = c.y; r.depth = c.z;
    return r;
}
constexpr float EXP_INV_C1 = -1.3862943611198906f;     // 1 / C = -2 ln 2: gives the conic back at the flush

struct PairB {
    f2 dx, dxdy, dx2, e2;
    float dy, dy2;
};
__device__ __forceinline__ void pair_exponent_bwd(const RecB& s, f2 px, float py, PairB& o)
{
    o.dx = px - splat2(s.mx);
    o.dy = py - s.my;
    o.dxdy = o.dx * splat2(o.dy);
    o.dx2 = o.dx * o.dx;
    o.dy2 = o.dy * o.dy;
    o.e2 = fma2(o.dxdy, splat2(s.B), fma2(o.dx2, splat2(s.A), splat2(o.dy2 * s.Dq)));
}

// per-pixel-pair state of the backward sweep.  With T the running transmittance, R_i = sum_{j<=i} T_j a_j S_j
// (S_j = cot . sample_j), K = cot . final (colour, depth) + T_n cT_n and sc the reference's T-anchor scale
// (T' = 1 - outAlpha = sc T_n), the sweep carries the three products it actually uses:
//   Ts = sc T,   Q = sc (K - R)   (what the rest of the list and the background still owe, times sc).
struct PairState {
    f2 px, Ts, Q;
    f2 cCx, cCy, cCz, cD;                // cotangents of colour / depth
    float py;
    uint32_t nc0, nc1;
};

// One splat against one pixel pair; adds the pair's contributions to the packed accumulators
//   acc: 0 sum h dx   1 sum h dy   2 sum h dx^2   3 sum h dxdy   4 sum h dy^2   5 dop   6 dr   7 dg   8 db   9 ddepth
// with h = dL/d(alpha) raw gated = -2 x (-1/2 dL/d(exponent)): the factor -1/2, the conic factors of the mean gradient
// (linear in the first two sums) and the sign are applied once per splat at the flush, not once per pixel.
// The cotangent of T_{i+1} is what the rest of the list and the background still owe,
//   c_i = (K - R_i) / T_{i+1},   since  cot . (C_final - C_i) = sum_{j>i} T_j a_j S_j,
// so   dL/d(alpha_i) = sc T_i (S_i - c_i) = Ts_i S_i - Q_i / (1 - alpha_i),   Q_i = Q_{i-1} - sc T_i a_i S_i:
// the sweep needs neither T nor R themselves, and one reciprocal of 1 - alpha (>= 0.01) per pixel.
template <bool DEPTH>
__device__ __forceinline__ void pair_bwd(const RecB& s, uint32_t i, const PairB& e, PairState& p, f2 (&acc)[10])
{
    const f2 G = (f2){__builtin_amdgcn_exp2f(e.e2.x), __builtin_amdgcn_exp2f(e.e2.y)};
    const f2 raw = splat2(s.op) * G;
    const bool a0 = i < p.nc0, a1 = i < p.nc1;
    f2 alpha;
    alpha.x = a0 ? fminf(raw.x, 0.99f) : 0.0f;
    alpha.y = a1 ? fminf(raw.y, 0.99f) : 0.0f;
    const f2 contrib = p.Ts * alpha;
    // without a depth cotangent cD = 0: the product is +-0 and fma(cCz, b, +-0) is the rounded product itself
    const f2 S = fma2(p.cCx, splat2(s.r), fma2(p.cCy, splat2(s.g),
                      DEPTH ? fma2(p.cCz, splat2(s.b), p.cD * splat2(s.depth)) : p.cCz * splat2(s.b)));
    p.Q = fma2(-contrib, S, p.Q);
    const f2 oma = splat2(1.0f) - alpha;
    const f2 owed = p.Q * (f2){__builtin_amdgcn_rcpf(oma.x), __builtin_amdgcn_rcpf(oma.y)};
    const f2 dAlpha = fma2(p.Ts, S, -owed);
    f2 gate;
    gate.x = (a0 && !(raw.x > 0.99f)) ? dAlpha.x : 0.0f;
    gate.y = (a1 && !(raw.y > 0.99f)) ? dAlpha.y : 0.0f;
    const f2 hh = gate * raw;
    acc[0] = fma2(hh, e.dx, acc[0]);
    acc[1] = fma2(hh, splat2(e.dy), acc[1]);
    acc[2] = fma2(e.dx2, hh, acc[2]);
    acc[3] = fma2(e.dxdy, hh, acc[3]);
    acc[4] = fma2(splat2(e.dy2), hh, acc[4]);
    acc[5] = fma2(G, gate, acc[5]);
    acc[6] = fma2(contrib, p.cCx, acc[6]);
    acc[7] = fma2(contrib, p.cCy, acc[7]);
    acc[8] = fma2(contrib, p.cCz, acc[8]);
    if (DEPTH) acc[9] = fma2(contrib, p.cD, acc[9]);
    p.Ts = p.Ts * oma;
}

// 10 sums per splat: dmx dmy dc00 dc01(=dc10) dc11 dop dr dg db ddepth; flushed into the reference's packed
// row order (dmx dmy dc00 dc01 dc10 dc11 dr dg db dop ddepth) of gradAcc16
// DEPTH = false: no depth cotangent (the default training case): nine sums per splat
template <int SEG, bool DEPTH>
__global__ __launch_bounds__(64) void blend_bwd_v2_kernel(
    int W, int H, int tileW, int tileH, int gridW, int blocksX, int whiteBg, const float4* __restrict__ rec12,
    const uint32_t* __restrict__ sortedIdx, uint32_t idxMask, const uint32_t* __restrict__ tileRanges,
    const uint32_t* __restrict__ itemRow, const uint4* __restrict__ segSlot, uint32_t qslotCap, int statePlanes,
    const uint32_t* __restrict__ blockWork,
    const uint32_t* __restrict__ itemBlock, uint32_t* __restrict__ counters, uint32_t* __restrict__ bwdQueue, uint32_t nq,
    const float* __restrict__ cotColor,
    const float* __restrict__ cotDepth, const float* __restrict__ cotAlpha, const float* __restrict__ outColor,
    const float* __restrict__ outDepth, const float* __restrict__ outAlpha, const uint32_t* __restrict__ lastContrib,
    const float* __restrict__ finalT, const float* __restrict__ segState, float* __restrict__ gradAcc16, GsVirtGeom vg)
{
    __shared__ float part[SEG][16];   // per splat: the ten sums at wave_sum10_transposed's lanes, conic terms in the gaps
    __shared__ f4 sg[192];
    const int lane = threadIdx.x;
    const uint32_t nItems = __builtin_amdgcn_readfirstlane(counters[GS_CNT_ITEMS]);
    // (popping the next item ahead of time was measured slower: vector-memory results return in order, so the first
    // record load of the current item then waits behind the contended atomic)
    // Work distribution (round 4), as in the forward: nq queues, one per XCD.  The item list is in block order, a block's
    // segments next to each other; queue x hands out the x-th nq-th of it (by item count), i.e. a horizontal stripe of the
    // image with about the same work as the others.  All the segments of a block -- which re-read the block's per-pixel
    // cotangents and state, 9 KB per item -- and the blocks next to it -- which share most of their records -- then run on
    // one XCD and find those bytes in its L2.  A wave whose own stripe is done takes from the others'.
    const uint32_t xcd = blockIdx.x % nq, slot = blockIdx.x / nq;
    const uint32_t perQ = (nItems + nq - 1u) / nq;
    const uint32_t staticPerQ = min(gridDim.x / nq, perQ);      // items of each stripe covered by the waves' first items
    uint32_t dead = 0;
    for (bool first = true;; first = false) {
        uint32_t item = 0xFFFFFFFFu;
        if (first && slot < staticPerQ) item = xcd * perQ + slot;
        else {
            for (uint32_t t = 0; t < nq && item == 0xFFFFFFFFu; t++) {
                const uint32_t y = (xcd + t) % nq;
                if ((dead >> y) & 1u) continue;
                uint32_t k = 0;
                if (lane == 0) k = atomicAdd(&bwdQueue[y * 32u], 1u);
                k = __builtin_amdgcn_readfirstlane(k);
                const uint32_t idx = staticPerQ + k;
                if (idx < perQ && y * perQ + idx < nItems) item = y * perQ + idx;
                else dead |= 1u << y;
            }
        }
        if (item == 0xFFFFFFFFu || item >= nItems) {
            if (first) continue;       // (a wave beyond the static share: try the queues)
            break;                     // every queue is empty; they only grow: every wave reaches this exit
        }
        const uint32_t packed = __builtin_amdgcn_readfirstlane(itemBlock[item]);
        const uint32_t row = __builtin_amdgcn_readfirstlane(itemRow[item]);
        const int b = (int)(packed >> 10);
        const uint32_t seg = packed & 1023u;
        const int by = b / blocksX, bx = b - by * blocksX;
        const int tile = ((by * BLK) / tileH) * gridW + (bx * BLK) / tileW;
        const uint32_t start = __builtin_amdgcn_readfirstlane(tileRanges[2 * tile]);
        const uint32_t work = __builtin_amdgcn_readfirstlane(blockWork[b]);
        const uint32_t i0 = seg * SEG, i1 = min(i0 + SEG, work);
        // the four quadrants' checkpoint slots in front of this segment (unused for segment 0)
        uint32_t qslot[4] = {0, 0, 0, 0};
        if (seg != 0) {
            const uint4 q4 = segSlot[row];
            qslot[0] = __builtin_amdgcn_readfirstlane(q4.x); qslot[1] = __builtin_amdgcn_readfirstlane(q4.y);
            qslot[2] = __builtin_amdgcn_readfirstlane(q4.z); qslot[3] = __builtin_amdgcn_readfirstlane(q4.w);
        }

        int BX0, BY0, BXL, BYL;     // the block's pixel origin and limits (block lists: blocks are enumerated per tile)
        gs_block_pixels(vg, bx, by, W, H, BX0, BY0, BXL, BYL);
        PairState ps[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            PairState& p = ps[h];
            const int y = BY0 + h * 8 + (lane >> 3);
            p.py = (float)y;
            p.Ts = p.Q = p.cCx = p.cCy = p.cCz = p.cD = splat2(0.f);
            uint32_t ncs[2] = {0, 0};
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int x = BX0 + k * 8 + (lane & 7);
                p.px[k] = (float)x;
                if (x < BXL && y < BYL) {
                    const size_t pix = (size_t)y * W + x;
                    const uint32_t n = lastContrib[pix];
                    if (n > i0) {
                        ncs[k] = n;
                        const float gx = cotColor[3 * pix], gy = cotColor[3 * pix + 1], gz = cotColor[3 * pix + 2];
                        const float gd = DEPTH ? cotDepth[pix] : 0.0f;
                        p.cCx[k] = gx; p.cCy[k] = gy; p.cCz[k] = gz; p.cD[k] = gd;
                        const float cA = cotAlpha ? cotAlpha[pix] : 0.0f;
                        const float Tn = finalT[pix];
                        const float bg = whiteBg ? Tn : 0.0f;
                        const float cTn = -cA + (whiteBg ? (gx + gy + gz) : 0.0f);
                        // What the rest of the list still owes: cot . (final sums - the sums in front of the segment).  Round 6:
                        // the DIFFERENCE per channel first, then the dot product -- deep in a list both are of the size of the
                        // total and their difference small, and cot . final - cot . checkpoint (rounds 2-5) carried two dot
                        // products' roundings at the size of the total; the forward keeps its sums so that the difference
                        // itself is good (blend_fwd_v2q_kernel: base + chunk)
                        const float sc = (1.0f - outAlpha[pix]) / Tn;     // the reference's T = 1 - outAlpha anchor
                        float T0 = 1.0f, s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;      // state in front of the segment
                        const uint32_t phys = qslot[h * 2 + k];
                        if (seg != 0 && phys < qslotCap) {
                            const float* st = segState + (size_t)phys * (statePlanes * 64) + lane;
                            T0 = st[0]; s0 = st[64]; s1 = st[128]; s2 = st[192];
                            if (DEPTH) s3 = st[256];
                        }
                        const float d0 = (outColor[3 * pix] - bg) - s0, d1 = (outColor[3 * pix + 1] - bg) - s1, d2 = (outColor[3 * pix + 2] - bg) - s2;
                        const float owedC = fmaf(gx, d0, fmaf(gy, d1, fmaf(gz, d2, DEPTH ? gd * (outDepth[pix] - s3) : 0.0f)));
                        p.Ts[k] = sc * T0;
                        p.Q[k] = sc * (owedC + Tn * cTn);
                    }
                }
            }
            p.nc0 = ncs[0]; p.nc1 = ncs[1];
        }

        // sweep length of each half (max nContrib over its pixels): past it the half is dead, like one out of reach
        uint32_t hm0 = max(ps[0].nc0, ps[0].nc1), hm1 = max(ps[1].nc0, ps[1].nc1);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            hm0 = max(hm0, (uint32_t)__shfl_xor((int)hm0, d, 64));
            hm1 = max(hm1, (uint32_t)__shfl_xor((int)hm1, d, 64));
        }
        const uint32_t* __restrict__ idx = sortedIdx + start;
        const uint32_t n = i1 - i0;
        // rows of splats that turn out culled are never written: start from zeros
        for (uint32_t r = lane; r < n * 4; r += 64) reinterpret_cast<f4*>(&part[0][0])[r] = (f4){0.f, 0.f, 0.f, 0.f};
        // Each 64-entry chunk is compacted at staging time, as in the forward: lane j tests list entry c0 + j against
        // the two 16x8 halves of the block (rect_min_q) and parks it only if it can reach one of them, tagged with its
        // list position and the halves it reaches.  Entries out of reach cost nothing below.
        const float bx0 = (float)BX0, bx1 = bx0 + 15.0f, by0 = (float)BY0;
        for (uint32_t c0 = i0; c0 < i1; c0 += 64) {
            uint32_t nEff;
            {
                const RecV v = load_chunk(rec12, idx, idxMask, c0, i1, lane);
                const float X0 = bx0 - v.a.x, X1 = bx1 - v.a.x, Yt = by0 - v.a.y;
                const uint32_t li = c0 + (uint32_t)lane;
                const bool far0 = li >= hm0 || rect_min_q(v.a.z, v.a.w, v.b.x, v.b.y, X0, X1, Yt, Yt + 7.0f) > CULL_QMIN;
                const bool far1 = li >= hm1 || rect_min_q(v.a.z, v.a.w, v.b.x, v.b.y, X0, X1, Yt + 8.0f, Yt + 15.0f) > CULL_QMIN;
                const bool keep = (c0 + lane < i1) && !(far0 && far1);
                const unsigned long long m = __ballot(keep);
                const uint32_t pos = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (keep) {
                    const uint32_t tag = (c0 - i0 + (uint32_t)lane) | (far0 ? 256u : 0u) | (far1 ? 512u : 0u);
                    // RecB: the conic goes in pre-multiplied (mx, my, A, B | Dq, -, r, g | b, op, depth, tag)
                    sg[pos * 3] = (f4){v.a.x, v.a.y, v.a.z * EXP_C1, (v.a.w + v.b.x) * EXP_C1};
                    sg[pos * 3 + 1] = (f4){v.b.y * EXP_C1, 0.0f, v.b.z, v.b.w};
                    sg[pos * 3 + 2] = (f4){v.c.x, v.c.y, v.c.z, __uint_as_float(tag)};
                }
                nEff = (uint32_t)__popcll(m);
            }
          for (uint32_t jl = 0; jl < nEff; jl++) {
            const f4 rc = sg[jl * 3 + 2];
            const uint32_t tag = __builtin_amdgcn_readfirstlane(__float_as_uint(rc.w));
            const uint32_t i = i0 + (tag & 255u);
            const RecB s = unpack_b(sg[jl * 3], sg[jl * 3 + 1], rc);
            PairB e0, e1;
            const bool k0 = (tag & 256u) != 0, k1 = (tag & 512u) != 0;     // wave-uniform: half out of reach
            if (!k0) pair_exponent_bwd(s, ps[0].px, ps[0].py, e0);
            if (!k1) pair_exponent_bwd(s, ps[1].px, ps[1].py, e1);
            f2 acc2[10];
#pragma unroll
            for (int q = 0; q < 10; q++) acc2[q] = splat2(0.0f);
            if (!k0) pair_bwd<DEPTH>(s, i, e0, ps[0], acc2);
            if (!k1) pair_bwd<DEPTH>(s, i, e1, ps[1], acc2);
            float acc[10];
#pragma unroll
            for (int q = 0; q < 10; q++) acc[q] = acc2[q].x + acc2[q].y;
            float w = wave_sum10_transposed<!DEPTH>(acc);
            // slot 4 r + q of the splat's row: 0 a1 (sum h dx), 2 a2 (sum h dy), 1 dc00, 3 dc01, 8 dc11, 10 dop, 9 dr,
            // 11 dg, 4 db, 6 ddepth; idle quads park the conic terms the flush needs for the mean gradient:
            // 5 c00, 7 c11, 13 c01 + c10
            if (lane == 20) w = s.A * EXP_INV_C1;        // c00
            if (lane == 28) w = s.Dq * EXP_INV_C1;       // c11
            if (lane == 52) w = s.B * EXP_INV_C1;        // c01 + c10
            if ((lane & 3) == 0) part[i - i0][lane >> 2] = w;
          }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the LDS writes have landed (single wave)
        __builtin_amdgcn_wave_barrier();
        for (uint32_t e = lane; e < n * 11; e += 64) {
            const uint32_t j = e / 11, q = e - j * 11;
            // packed column q (dmx dmy dc00 dc01 dc10 dc11 dr dg db dop ddepth) <- slot
            float v;
            if (q < 2) {
                // d mean = -(2 c00 A1 + (c01 + c10) A2,  2 c11 A2 + (c01 + c10) A1),  A = -1/2 of the stored sums
                const float a1 = -0.5f * part[j][0], a2 = -0.5f * part[j][2], cs = part[j][13];
                v = q == 0 ? -(2.0f * part[j][5] * a1 + cs * a2) : -(2.0f * part[j][7] * a2 + cs * a1);
            } else {
                const uint32_t src = (0x6a4b98331ull >> ((q - 2) * 4)) & 15u;     // 1 3 3 8 9 11 4 10 6
                v = (DEPTH || q != 10) ? part[j][src] : 0.0f;
                if (q < 6) v *= -0.5f;                                            // the four conic columns
            }
            if (v != 0.0f) atomicAdd(&gradAcc16[(size_t)(idx[i0 + j] & idxMask) * 16 + q], v);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
constexpr int SEGLEN = GS_SEG_LEN;
// the grid the fused forward is launched with (its queue starts behind the waves' static first items)
// four waves per quadrant (blend_fwd_v2w_kernel) where the image has fewer quadrants than the chip has wave slots
// (gs_ctx::fwdWide: 1 / 0 force, -1 by that rule; the trace buffer and the 16x8 variant keep the one-wave kernel)
bool blend_forward_v2_wide(const gs_ctx* c)
{
    if (c->fwdTrace || !c->fwdQuadrants) return false;
    return c->fwdWide == 1 || (c->fwdWide < 0 && c->numPixBlocks * 4 <= c->numCUs * 4 * c->fwdWavesPerSimd);
}
// a staging wave beside every sweeping wave (blend_fwd_v2p_kernel; gs_ctx::fwdPair): where the one-wave kernel would run
// fwdPair < 0: by the depth of the lists -- the pair count of the context's PREVIOUS forward (a word the expansion leaves in
// mapped host memory: read without a wait, a forward late) per pixel block.  Measured (MI355X, blend forward, one-wave / pair):
// c3 grown to 1 M Gaussians, 8400 pairs per block: 0.890 -> 0.805 ms; c3 at 200 x 200 tiles, 5070: 0.237 / 0.238; c3, 3160:
// 0.175 / 0.176; c2 0.174 / 0.174; the 2 M garden scene under its depth cuts, 740: 0.311 -> 0.320.  The decision is taken ONCE
// per forward (gs_render_forward, fwdPairNow): the binning's bookkeeping and the launch must see the same grid.
bool blend_forward_v2_pair_decide(const gs_ctx* c)
{
    if (c->fwdTrace || !c->fwdQuadrants || blend_forward_v2_wide(c)) return false;
    if (c->fwdPair >= 0) return c->fwdPair != 0;
    const unsigned long long lastM = c->missHost ? c->missHost[8] : 0u;
    // (4500: round 6's lists are trimmed rects in row groups -- the grown scene's 8400 pairs per block became 5500, its staging-wave
    // forward still 5 % ahead, 0.868 -> 0.825 ms; 200 x 200 tiles on block lists, 5070: level either way)
    return c->numPixBlocks > 0 && lastM / (unsigned long long)c->numPixBlocks >= 4500ull;
}
bool blend_forward_v2_pair(const gs_ctx* c)
{
    return c->fwdPairNow;
}
int blend_forward_v2_grid(const gs_ctx* c)
{
    const int fwdItems = c->numPixBlocks * 4;
    const int fwdGrid = blend_forward_v2_wide(c) ? c->numCUs * GS_V2W_WGS
                        : blend_forward_v2_pair(c) ? c->numCUs * (c->fwdPair > 1 ? c->fwdPair : GS_V2P_WGS)
                                                   : c->numCUs * 4 * c->fwdWavesPerSimd;
    return fwdGrid > fwdItems ? fwdItems : fwdGrid;
}

void fill_seg_base(gs_ctx* c, SegBaseArgs& a)
{
    a.nBlocks = c->numPixBlocks; a.blocksX = c->blocksX; a.tileW = c->tileW; a.tileH = c->tileH; a.gridW = c->gridW;
    a.tileRanges = c->tileRanges; a.tileTotal = nullptr;
    a.segBase = c->segBase; a.blockWork = c->blockWork; a.counters = c->counters; a.workHint = c->workHint;
    a.blockOrder = c->blockOrder; a.queueStart = (uint32_t)blend_forward_v2_grid(c); a.fwdQueue = c->fwdQueue; a.nq = c->fwdQueues; a.spatial = c->fwdSpatial;
}

int launch_blend_forward_v2(gs_ctx* c, float* outColor, float* outDepth, float* outAlpha)
{
    // the backward of THIS forward reads what it wrote; without a depth image there is no depth cotangent to come either
    // (render-only forwards keep no checkpoints at all: the arena's arithmetic below then runs on four planes and is not used)
    const int planes = c->depthGradient && outDepth ? 5 : 4;
    c->fwd.statePlanes = planes;
    const int blocksX = c->blocksX, nBlocks = c->numPixBlocks;
    const int nItems = nBlocks * 4, grid = blend_forward_v2_grid(c);
    if (c->segBaseDone) c->segBaseDone = false;        // the tile sort's launch has done it (binning.hip)
    else {
        SegBaseArgs sa;
        fill_seg_base(c, sa);
        hipLaunchKernelGGL(seg_base_kernel<SEGLEN>, dim3(1), dim3(1024), 0, c->stream, sa);
    }
    const uint32_t* cuts = c->fwd.cutsActive ? c->fwd.cutStore : nullptr;
    // capacity in slots of THIS forward's planes (the arena is sized for five)
    const uint32_t qcap = (uint32_t)(c->qslotCap * 5 / c->fwd.statePlanes);
    c->fwd.qslotCap = qcap;
    // three quarters of the arena at most are handed out statically, up to 32 slots (2048 list entries of one quadrant)
    // per wave; the rest is the shared part, in eight
    uint32_t own = (uint32_t)(((unsigned long long)qcap * 3ull / 4ull) / (unsigned long long)grid);
    if (own > 32u) own = 32u;
    c->fwd.qslotStatic = own * (uint32_t)grid;
    const uint32_t partSlots = (qcap - c->fwd.qslotStatic) / 8u;
    // test knob (GS_TUNE_POISON_CHECKPOINTS): every checkpoint lane this forward does not write reads back as NaN
    if (c->poisonCheckpoints && c->segState)
        GS_HIP_CHECK(c, hipMemsetAsync(c->segState, 0xFF, (size_t)c->qslotCap * 5 * 64 * sizeof(float), c->stream));
    if (blend_forward_v2_wide(c)) {     // four waves per quadrant: the grid is workgroups, the arena's static shares go per wave
        uint32_t ownW = (uint32_t)(((unsigned long long)qcap * 3ull / 4ull) / (unsigned long long)(grid * 4));
        if (ownW > 32u) ownW = 32u;
        c->fwd.qslotStatic = ownW * (uint32_t)grid * 4u;
        const uint32_t partW = (qcap - c->fwd.qslotStatic) / 8u;
        auto kw = outDepth ? blend_fwd_v2w_kernel<SEGLEN, true> : blend_fwd_v2w_kernel<SEGLEN, false>;
        hipLaunchKernelGGL(kw, dim3(grid), dim3(256), 0, c->stream, c->W, c->H, c->tileW,
                           c->tileH, c->gridW, blocksX, nItems, c->whiteBg, reinterpret_cast<const float4*>(c->packed12),
                           c->sortedRaw, c->idxMask, c->tileRanges, c->segBase, (uint32_t)c->segCap, c->renderOnly ? 0 : c->fwd.statePlanes, outColor, outDepth,
                           outAlpha, c->lastContrib, c->finalT, c->segState, c->segSlot, qcap, ownW, partW, ckpt_pool(partW, (uint32_t)grid * 4u), c->blockWork, c->counters, c->fwdQueue,
                           (uint32_t)c->fwdQueues, c->blockOrder, cuts, c->missDev, c->fwdFoldScale, c->virt);
        GS_HIP_CHECK(c, hipGetLastError());
        if (c->renderOnly) c->fwd.statePlanes = 0;      // (backward_preflight refuses such a forward)
        return GS_OK;
    }
    if (blend_forward_v2_pair(c)) {
        auto kp = outDepth ? blend_fwd_v2p_kernel<SEGLEN, true> : blend_fwd_v2p_kernel<SEGLEN, false>;
        hipLaunchKernelGGL(kp, dim3(grid), dim3(128), 0, c->stream, c->W, c->H, c->tileW,
                           c->tileH, c->gridW, blocksX, nItems, c->whiteBg, reinterpret_cast<const float4*>(c->packed12),
                           c->sortedRaw, c->idxMask, c->tileRanges, c->segBase, (uint32_t)c->segCap, c->renderOnly ? 0 : c->fwd.statePlanes, outColor, outDepth,
                           outAlpha, c->lastContrib, c->finalT, c->segState, c->segSlot, qcap, own, partSlots, ckpt_pool(partSlots, (uint32_t)grid), c->blockWork, c->counters, c->fwdQueue, (uint32_t)c->fwdQueues, c->blockOrder,
                           cuts, c->missDev, c->virt);
        GS_HIP_CHECK(c, hipGetLastError());
        if (c->renderOnly) c->fwd.statePlanes = 0;
        return GS_OK;
    }
    auto kern = outDepth ? blend_fwd_v2q_kernel<SEGLEN, true> : blend_fwd_v2q_kernel<SEGLEN, false>;
    hipLaunchKernelGGL(kern, dim3(grid / 4), dim3(256), 0, c->stream, c->W, c->H, c->tileW,      // (grid = waves, a multiple of four)
                       c->tileH, c->gridW, blocksX, nItems, c->whiteBg, reinterpret_cast<const float4*>(c->packed12),
                       c->sortedRaw, c->idxMask, c->tileRanges, c->segBase, (uint32_t)c->segCap, c->renderOnly ? 0 : c->fwd.statePlanes, outColor, outDepth,
                       outAlpha, c->lastContrib, c->finalT, c->segState, c->segSlot, qcap, own, partSlots, ckpt_pool(partSlots, (uint32_t)grid), c->blockWork, c->counters, c->fwdQueue, (uint32_t)c->fwdQueues, c->blockOrder,
                       c->fwdTrace, cuts, c->missDev, (uint32_t)c->fwdSlowSlot, c->virt);
    GS_HIP_CHECK(c, hipGetLastError());
    if (c->renderOnly) c->fwd.statePlanes = 0;          // (backward_preflight refuses such a forward)
    return GS_OK;
}

void fill_bwd_prep(gs_ctx* c, int N, uint32_t queueStart, BwdPrepArgs& p)
{
    p.nBlocks = c->numPixBlocks;
    p.blockWork = c->fwd.blockWork;
    p.itemBlock = c->itemBlock;
    p.itemRow = c->itemRow;
    p.segBase = c->segBase;
    p.itemCap = (uint32_t)c->itemCap;
    p.counters = c->counters;
    p.queueStart = queueStart;
    p.bwdQueue = c->bwdQueue;
    p.clearBuf = reinterpret_cast<float4*>(c->gradAcc16);
    p.clearCount = (size_t)N * 4;
    // the view's cuts are renewed whenever the caller keeps them (gs_set_view_hints), in force this forward or not
    p.cutStore = c->fwd.cutStore;
    p.cutsInForce = c->fwd.cutsActive ? 1 : 0;
    p.tileRanges = c->tileRanges;
    p.sortedIdx = c->sortedRaw;
    p.idxMask = c->idxMask;
    p.rec12 = c->packed12;
}

// the grid the fused backward will be launched with (its queue starts behind the waves' static first items)
int blend_backward_v2_grid(const gs_ctx* c)
{
    int grid = c->numCUs * c->bwdWavesPerCu;
    if ((long long)grid > c->itemCap) grid = (int)c->itemCap;
    return grid < 1 ? 1 : grid;
}

int launch_blend_backward_v2(gs_ctx* c, int N, const float* cotColor, const float* cotDepth, const float* cotAlpha,
                             const float* outColor, const float* outDepth, const float* outAlpha)
{
    const int blocksX = c->blocksX, nBlocks = c->numPixBlocks;
    const int grid = blend_backward_v2_grid(c);
    BwdPrepArgs prep;
    fill_bwd_prep(c, N, (uint32_t)grid, prep);
    // gs_loss_forward_backward has carried all of this along with its own kernel (ssim.hip) when the loss of this
    // forward went through the library
    const bool prepared = c->fwd.bwdPrepared && c->fwd.preparedQueueStart == (uint32_t)grid && c->fwd.preparedN == N;
    c->fwd.bwdPrepared = false;          // consumed: a second backward of the same forward prepares for itself
    if (!prepared) {
        const int cutBlocks = prep.cutStore ? gs_div_up(nBlocks, 1024) : 0;
        const int clearBlocks = (int)((prep.clearCount + 8191) / 8192 < 1024 ? (prep.clearCount + 8191) / 8192 : 1024);
        hipLaunchKernelGGL(bwd_items_kernel<SEGLEN>, dim3(GS_ITEM_PARTS + cutBlocks + clearBlocks), dim3(1024), 0, c->stream, prep, cutBlocks);
    }
    auto kern = cotDepth ? blend_bwd_v2_kernel<SEGLEN, true> : blend_bwd_v2_kernel<SEGLEN, false>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, c->stream, c->W, c->H, c->tileW, c->tileH,
                       c->gridW, blocksX, c->whiteBg, reinterpret_cast<const float4*>(c->packed12), c->sortedRaw,
                       c->idxMask, c->tileRanges, c->itemRow, reinterpret_cast<const uint4*>(c->segSlot), c->fwd.qslotCap, c->fwd.statePlanes, c->fwd.blockWork, c->itemBlock, c->counters, c->bwdQueue, (uint32_t)c->bwdQueues, cotColor,
                       cotDepth, cotAlpha, outColor, outDepth, outAlpha, c->lastContrib, c->finalT, c->segState,
                       c->gradAcc16, c->virt);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

}  // namespace gs
