// gs_ctx.h -- context/workspace shared by the translation units of libgsplat_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/gsplat.h"
#include "gs_math.h"

// device counters held in ctx->counters (u32 each)
enum {
    GS_CNT_M = 0,         // pairs actually binned (0 on overflow)
    GS_CNT_OVERFLOW = 1,  // 1 if M exceeded the reserved capacity
    GS_CNT_MREQ = 2,      // M that was required
    GS_CNT_B = 3,         // max tile list length
    GS_CNT_NVIS = 4,      // Gaussians with radius > 0
    GS_CNT_CONTRIB_LO = 5,
    GS_CNT_CONTRIB_HI = 6,
    GS_CNT_ITEMS = 9,     // backward work items (block, segment)
    GS_CNT_QUEUE = 10,    // backward work-queue head
    GS_CNT_QUEUE_FWD = 11,  // forward work-queue head
    GS_CNT_CUT_DROPPED = 12,  // statistics: candidate pairs the depth cuts left out (low 32 bits)
    GS_CNT_QSLOTS = 16,   // ... + 8: checkpoint slots (one 8x8 quadrant each) the fused forward's waves have drawn from the
                          // shared part of the arena, one counter per eighth of it (blend_v2.hip)
    GS_CNT_COUNT = 24
};

constexpr int GS_SORT_THREADS = 256;
constexpr int GS_SORT_ITEMS = 16;
constexpr int GS_SORT_TILE = GS_SORT_THREADS * GS_SORT_ITEMS;  // elements per sort block
constexpr int GS_SMALL_SORT_ITEMS = 16;     // elements per thread of the small (depth) sort's tiles (4: 1024-element tiles were
                                            // slower, 11 -> 14 us per scatter: four times the histogram rows to sum per block)
constexpr int GS_SMALL_SORT_BLOCKS = 160;   // depth sorts of up to this many tiles (655 k Gaussians) take the two-launch passes
constexpr int GS_SORT_MAX_GRID = 2048;      // blocks of a radix kernel launched for a device-resident count (they walk the rest)
constexpr uint32_t GS_SORT_NO_KEY = 0xFFFFFFFFu;   // depth key of a Gaussian that touches no tile (never a real key: a NaN)
constexpr int GS_WIDE_BINS = 4096;          // one-pass tile sort: a 12-bit digit covers every tile id when T <= 4096
constexpr int GS_WIDE_CHUNK = 16;           // sort tiles per scan chunk (an exclusive prefix inside a chunk stays < 2^16)
constexpr int GS_SCAN_BLOCK = 256;
constexpr uint32_t GS_SLICE_MIN_PAIRS = 32768;   // ... of the blocks that have at least this many positions
constexpr int GS_EXPAND_SLICES = 8;       // slices of a wave's positions in the expansion of large inputs (binning.hip)
constexpr int GS_FUSED_SCAN_MAX = 2048;   // scan blocks up to which every expansion block sums the block counts itself
constexpr int GS_SEG_LEN = 64;  // splats per saved-state segment of the fused blend (multiple of 4)

// Adam's parameter step lr m / (sqrt(v) + eps) with the hardware's 1-ulp v_sqrt_f32 and v_rcp_f32 instead of the
// correctly rounded sqrtf and division (~22 VALU instructions per element less; the fused projection backward + Adam
// kernel spent 40 % of its instructions there).  The step differs from the IEEE one by at most ~3e-7 relative, the
// moments not at all; MLX's Metal kernels are fast-math themselves and no reference test pins Adam.
#if defined(__HIPCC__)
__device__ __forceinline__ float gs_adam_delta(float lr, float m, float v, float eps)
{
    return (lr * m) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + eps);
}
#endif

struct GsDp;      // dp.hip: communicator, side stream and events of the data-parallel step

// the fused forward's SH colours as rider workgroups of the binning kernels (gs_rider.h): what a rider needs, by value
struct GsColourRider {
    const float* xyz;
    const float* fdc;
    const float* frest;
    float* packed12;
    const uint32_t* tilesTouched;
    float cam[3];
    int N, degree;
    int unit0, units;        // this launch's share: the waves [unit0, unit0 + units) of 64 Gaussians each
};
// host kernels of the riders (binning.hip), index into gs_ctx::riderShare
enum { GS_RIDE_SS_HIST = 0, GS_RIDE_SS_SCATTER, GS_RIDE_WIDE_TILE, GS_RIDE_HOSTS };
struct GsRiderState {
    bool on = false;         // a geometry-only projection has run: colour units are outstanding
    GsColourRider args = {};
    int next = 0, total = 0; // units handed out so far / units of the forward
};

// Block geometry of the fused path when the caller's tile size is not a multiple of 16 (the reference app's W/4 x H/4):
// every tile is cut into nbx x nby pixel blocks of 16 x 16 (the last column / row narrower), and the fused path bins, sorts and
// blends per BLOCK -- the context's tileW / tileH / gridW / gridH / T then describe that block grid ("block lists"), while the
// op-level entry points swap the caller's tile grid back in (GsRealGeom, api.hip).  nbx == 0: the regular 16 x 16 grid over the
// image, blocks and tiles related by division as before.
struct GsVirtGeom {
    int nbx = 0, nby = 0;    // blocks per tile
    int tw = 0, th = 0;      // the caller's tile size in pixels
};
// pixel origin and exclusive pixel limits of block (bx, by); all arguments wave-uniform
__device__ __forceinline__ void gs_block_pixels(const GsVirtGeom& g, int bx, int by, int W, int H, int& X0, int& Y0, int& XL, int& YL)
{
    if (g.nbx == 0) { X0 = bx * 16; Y0 = by * 16; XL = W; YL = H; return; }
    const int tx = bx / g.nbx, ix = bx - tx * g.nbx, ty = by / g.nby, iy = by - ty * g.nby;
    X0 = tx * g.tw + ix * 16; Y0 = ty * g.th + iy * 16;
    XL = min(W, min(X0 + 16, (tx + 1) * g.tw)); YL = min(H, min(Y0 + 16, (ty + 1) * g.th));
}
constexpr int GS_CUT_SUPER = 4;      // tiles per side of a super-tile of the coarse cuts
// coarse depth cuts handed to the fused projection (binning.hip, cut_super_kernel): null superCut = none
struct GsCutCoarse {
    const uint32_t* superCut = nullptr;
    int superW = 0;
    uint32_t* dropPerBlock = nullptr;
};
struct GsRealGeom {          // the caller's tile grid (what the op-level entry points see)
    int tileW = 16, tileH = 16, gridW = 0, gridH = 0, T = 0, tileBits = 1;
    bool fast16 = false;
};

struct gs_ctx {
    int device = 0;
    GsDp* dp = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    int W = 0, H = 0, tileW = 16, tileH = 16, gridW = 0, gridH = 0, T = 0, degree = 0, whiteBg = 0;
    int tileBits = 1;
    bool fast16 = false;
    GsVirtGeom virt;             // block lists (above); virt.nbx != 0 <=> tileW .. T describe the block grid, `real` the caller's
    GsRealGeom real;
    bool binIsBlockLists = false;  // the context's last binning was a fused forward's under block lists
    int blocksX = 0, blocksY = 0;  // pixel blocks per row / column of the fused path (numPixBlocks = blocksX * blocksY)
    std::string err;

    // capacities
    int capN = 0;
    long long capM = 0;
    bool pairsReserved = false;  // caller sized capM itself: no host check of M per call
    bool reserving = false;      // inside gs_ctx_reserve with a pair reserve (pairsReserved is set once it has succeeded)
    size_t ws_bytes = 0;

    // per-Gaussian workspace
    float* packed12 = nullptr;       // [capN,12] means2d, conic, colour, opacity, depth, pad 
    float* gradAcc16 = nullptr;      // [capN,16] blend-backward accumulator (64-B rows)
    uint32_t* depthKey[2] = {nullptr, nullptr};
    uint32_t* depthVal[2] = {nullptr, nullptr};
    uint32_t* tilesTouched = nullptr;  // [capN] by Gaussian index
    uint4* tilePieces = nullptr;       // [capN] trimmed rects: the four row groups of tileRect (first column | columns << 16), gs_math.h rect_row_groups4
    bool piecesValid = false;          // the projection of the forward in flight wrote them: its expansion enumerates row groups
    ushort4* tileRect = nullptr;       // [capN] x0,y0,x1,y1
    uint32_t* blockSums = nullptr;     // [capN/256+1]
    uint32_t* visPerBlock = nullptr;   // [capN/128+1] visible (radius > 0) Gaussians per projection block; summed on demand
    int visBlocks = 0;
    // per-pair workspace
    uint32_t* pairKey[2] = {nullptr, nullptr};
    uint32_t* pairVal[2] = {nullptr, nullptr};
    // radix scratch
    uint32_t* hist = nullptr;  // [256, nbCap]
    uint32_t* rowTotal = nullptr;  // [256]
    uint16_t* wideCnt = nullptr;     // [nbCap][4096] one-pass tile sort: pairs per (sort tile, tile id), then prefixes inside a chunk
    uint32_t* wideChunk = nullptr;   // [nbCap / 16 + 1][4096] pairs per (chunk, tile id), then prefixes over the chunks
    uint32_t* wideTotal = nullptr;   // [4096] pairs per tile id
    uint2* sortBits = nullptr;     // [GS_SMALL_SORT_BLOCKS] per sort tile: AND / OR of the depth keys that have pairs
    // splitter depth sort (binning.hip): splitters of the previous sort (double-buffered), bucket of every record, bucket starts
    uint32_t* sortSplit[2] = {nullptr, nullptr};   // [256] each
    int splitCur = 0;
    bool haveSplitters = false;
    int splitterNS = 0;                            // how many splitters sortSplit holds: 127 (256 buckets) or 255 (512 buckets)
    unsigned short* bucketId = nullptr;            // [capN] (read as bytes by the 256-bucket kernels)
    uint32_t* bucketStart = nullptr;               // [520]: first record of every bucket, [buckets] = n
    uint32_t* ssChunk = nullptr;                   // [65][512] bucket totals per chunk of 16 sort tiles (sorts of > 160 tiles)
    int colourRiders = 1;          // 1: K = 25 forwards compute their SH colours as riders of the binning kernels (GS_TUNE_COLOUR_RIDERS)
    GsRiderState rider;
    int riderShare[GS_RIDE_HOSTS] = {200, 500, 300};   // permille of a forward's colour units per host kernel
    int splitterSort = 1;          // 1: depth sorts of 16385 .. 655 k records take the splitter buckets (three launches); 0: LSD passes;
                                   // 2: larger sorts too (512 buckets, four launches: measured slower than their LSD passes)
    int nbCap = 0;
    // per-tile
    bool superCutReady = false;      // cut_super_kernel has run for the forward being built (the fused projection uses it first)
    uint32_t* dropPerBlock = nullptr; // [capN / 128 + 1] candidate pairs of the Gaussians each projection block dropped whole under the super-cuts
    int dropBlocks = 0;              // how many of them the current forward wrote (0: none -- no cuts, or an op-level binning)
    uint32_t* superCut = nullptr;    // [ceil(gridW / 4) * ceil(gridH / 4)] the deepest cut of every 4 x 4 tiles (binning.hip, cut_super_kernel)
    int cutSuper = 1;                // 0 (GSPLAT_CUT_SUPER=0): the cut expansion enumerates every Gaussian's rect (A/B)
    uint32_t* tileRanges = nullptr;  // [T,2]
    uint32_t* tileCounts = nullptr;  // [T]
    // per 16x16 pixel block: work estimate and heaviest-first launch order (fast path)
    uint32_t* blockWork = nullptr;   // [numPixBlocks] active buffer: blockWorkOwn, or the caller's (gs_set_block_work_buffer)
    uint32_t* blockWorkOwn = nullptr;
    uint32_t* blockOrder = nullptr;  // [numPixBlocks]
    uint32_t* fwdQueue = nullptr;    // [8][32] work-queue heads of the fused blend forward, one per XCD
    uint32_t* bwdQueue = nullptr;    // [8][32] ... and of the fused blend backward
    int bwdQueues = 8;               // 8 = one per XCD (a stripe of the item list each), 1 = one for the chip (A/B)
    // depth cuts (binning.hip): per tile, 0xFFFFFFFF - (largest depth key still binned); 0 = no cut.  Lives in the
    // caller's per-view hint buffer behind the block-work words (gs_set_view_hints); written by the backward's item
    // kernel, read by the next forward of that view.
    uint32_t* cutStore = nullptr;
    bool allowCuts = true;
    const uint32_t* opCuts = nullptr;    // gs_tile_bin_cut: the caller's per-tile cuts, for the duration of that call
    unsigned long long* scanTmp = nullptr;      // [.. / 1024 + 4] chunk sums of the large prefix
    unsigned long long* scanPrefix = nullptr;   // [capN/64 + 16] prefix of block / segment counts (large inputs only)
    uint2* waveSeg = nullptr;            // [capN/64 + 8] per expansion wave: start and length of its segment of kept pairs
    uint32_t* missHost = nullptr;        // pinned, mapped: [0] = 1 if a tile with a cut ended with live pixels; [1], [2] cut
                                         // statistics; [4] = 1 once a forward overflowed the reserved pairs, [5] = the M it needed
    uint32_t* missDev = nullptr;         // device address of missHost
    hipEvent_t fwdDone = nullptr;        // recorded after the forward blend when cuts were active
    const uint32_t* workHint = nullptr;  // = the caller's block-work buffer: sweep lengths of an earlier forward of this view
    float* gradNormAccum = nullptr;      // caller-owned [N]: the projection backward adds |grad xyz| (gs_set_grad_norm_accum)
    uint32_t* segBase = nullptr;     // [numPixBlocks] first saved-state slot of each block
    float* segState = nullptr;       // [qslotCap][5][64] running (T, C, D) of an 8x8 quadrant, saved every GS_SEG_LEN splats
    uint32_t* segSlot = nullptr;     // [segCap][4] slot of (block's first segment + segment, quadrant); blend_v2.hip
    long long segCap = 0;            // table rows: every list swept to its end
    long long qslotCap = 0;          // slots the arena holds (allocated as they are written)
    long long qslotWanted = 0;       // ... and what the next (re)allocation must hold at least (after an arena overflow)
    bool arenaRegrowPending = false; // an arena overflow has been reported; gs_ctx_reserve acts on it
    uint32_t* itemBlock = nullptr;   // [itemCap] backward work items
    uint32_t* itemRow = nullptr;     // [itemCap] their rows of the checkpoint-slot table
    long long itemCap = 0;
    float* finalT = nullptr;         // [P] exact final transmittance of the fused forward
    int numCUs = 256;
    int numPixBlocks = 0;
    int opBlocks = 0;                // work items of the op-level blend kernels (blend.hip): numPixBlocks, or -- tile sizes
                                     // that are not multiples of 16 -- the 16x16 blocks enumerated per tile
    // launch tuning (gs_ctx_set_tuning; per context): measured optima of tools/sweep.sh as defaults
    int fwdWavesPerSimd = 4, bwdWavesPerCu = 16;
    int fwdQuadrants = 1;            // (retired knob: the forward's items are always 8x8 quadrants)
    int scatterThreads = 0;          // threads per sort tile of the one-pass tile sort: 256 / 512 / 1024, 0 = by the Gaussian count (binning.hip)
    int lsdThreads = 0;              // threads per tile of the LSD depth passes' scatter: 256 / 1024, 0 = by the tile count (binning.hip)
    int fwdWide = -1;                // blend forward with four waves per quadrant: 1 / 0, -1 = where the image has fewer quadrants than wave slots (blend_v2.hip)
    float fwdFoldScale = 1.0f;       // test knob (GS_TUNE_FWD_FOLD_TEST_SCALE): factor on the composed T in the four-wave fold's
                                     // "did the pixel cross 1e-4 inside this part" test; below 1 forces second takes that come back live
    int fwdSlowSlot = 3;             // GS_TUNE_FWD_SLOW_SLOT: waves of the one-wave forward in a hardware wave slot >= this take their
                                     // static first item and nothing from the queues (16 = off; blend_v2.hip)
    int fwdPair = -1;                // GS_TUNE_FWD_PAIR: 0 = the one-wave forward, 1 = a staging wave beside every sweeping wave
                                     // (blend_fwd_v2p_kernel, 12 workgroups per CU), n > 1 = that with n workgroups per CU,
                                     // -1 (default) = where the previous forward's lists were deep (blend_forward_v2_pair_decide)
    bool fwdPairNow = false;         // this forward's decision (taken once, in gs_render_forward)
    bool trimRects = true;           // GS_TUNE_TRIM_RECTS: at 16 x 16 tiles the fused forward bins a Gaussian on its 3-sigma square cut by the
                                     // box of q <= 40.3 (projection.hip; gs_math.h block_rect_of_splat) instead of on the whole square
    bool rowGroups = true;           // GS_TUNE_TRIM_RECTS = 2 (default) / 1: trimmed rects cut further into four row groups, or the box alone
    int renderOnly = 0;              // GS_TUNE_RENDER_ONLY: fused forwards keep no checkpoints (statePlanes 0) and can have no backward
    int poisonCheckpoints = 0;       // test knob (GS_TUNE_POISON_CHECKPOINTS): the checkpoint arena is NaN-filled in front of every fused forward
    int rankSort = 1;                // depth sorts of <= 16384 records by rank on the whole chip (0: the one-workgroup radix sort; binning.hip)
    int fwdSpatial = 0;              // 0 (default): the blocks are dealt to the forward's queues round-robin in launch order (deepest first
                                     // over the whole image); 1: queue x gets the x-th stripe of the image (a third of the fabric
                                     // traffic, but equal block counts are not equal work: +25 % on the grown scene; GSPLAT_FWD_SPATIAL)
    int fwdQueues = 8;               // work queues of the fused forward: 8 = one per XCD (blend_v2.hip), 1 = one for the chip (A/B)
    int opFwdPpl = 1, opBwdPpl = 1;  // pixels per lane of the op-level blend kernels (blend.hip)
    bool segBaseWanted = false;      // gs_render_forward (16x16-block path): the binning may do the blend forward's
    bool segBaseDone = false;        //   bookkeeping in its tile-sort launch (gs_bwd_prep.h, seg_base_body) / it has
    int wideTileSort = 1;            // 1: one-pass tile sort when T <= 4096 (binning.hip); 0: the two 8-bit passes (A/B, tests)
    int hostOverflowErrors = 1;      // 0: a reserved-capacity overflow is reported by gs_sync only (GS_TUNE_HOST_OVERFLOW_ERRORS)
    int depthGradient = 1;           // 0: the caller promises cot_depth == NULL in every fused backward (default training,
                                     // SURVEY a11): the forward then checkpoints (T, R, G, B) without the depth sum
    unsigned long long* fwdTrace = nullptr;   // diagnostic: per-item (start, end, iterations, hw id) of the fused forward
    const uint32_t* adamGate = nullptr;       // device word: non-zero = every optimizer kernel leaves the parameters alone
                                              // (default: counters + GS_CNT_OVERFLOW; gs_set_update_gate)
    // data-parallel steps, round 5: the step's gate rides in its first collective instead of in a 4-byte all-reduce of its own
    float* overflowRider = nullptr;           // gs_set_overflow_rider: where the first kernel of a backward stores the forward's overflow
                                              // word as 0.0f / 1.0f (behind the colour cotangents / the gradient arena)
    long long ccBlockFloats = 0;              // gs_set_gathered_gate: floats per rank block of the gathered colour cotangents (0: 3 N, no
    int ccBlockCount = 0;                     //   gathered gate), how many blocks,
    uint32_t* gatheredGateOut = nullptr;      //   and where the SH rebuild stores the OR of their rider words
    uint32_t* gateSeen = nullptr;             // gs_set_gate_seen: set to 1 by an optimizer kernel that finds its gate raised
    // per-pixel (saved forward state for the fused path)
    uint32_t* lastContrib = nullptr;  // [P]
    float* lossPartials = nullptr;    // [lossPartialBlocks*4 + 16]
    float* lossTargetCache = nullptr; // caller-owned [2][3][H][W]: the target's windowed mean and mean of squares (ssim.hip)
    bool lossTargetCacheFilled = false;
    int lossPartialBlocks = 0;
    float* windowDev = nullptr;       // [121] default SSIM window
    // densify scan scratch: [densifyTileCap] tile sums + 8 counters, grown on demand
    int* densifyTiles = nullptr;
    int densifyTileCap = 0;
    // the planned densify event (densify.hip): its plan words on the device, their copy in pinned host memory, the event behind them
    uint32_t* densifyPlan = nullptr;
    uint32_t* densifyPlanHost = nullptr;
    uint32_t* densifyPlanHostDev = nullptr;
    hipEvent_t densifyDone = nullptr;
    bool densifyPlanned = false;
    float** densifyTable = nullptr;      // device: the six tensor starts of a packed planned gather (dn_packed_table_kernel)
    // counters
    uint32_t* counters = nullptr;  // device [GS_CNT_COUNT]
    uint32_t* countersHost = nullptr;  // pinned host mirror

    // results of the last binning
    const uint32_t* sortedIdx = nullptr;  // plain Gaussian indices (valid when sortedPlainValid)
    const uint32_t* sortedRaw = nullptr;  // what the sort produced: packed (tile << idxBits | index) or plain
    uint32_t idxMask = 0xFFFFFFFFu;
    int idxBits = 0;
    bool sortedPlainValid = false;
    bool binValid = false;
    int binN = 0;

    // stage profiling (HIP events on the ctx stream)
    struct ProfEvent { hipEvent_t a, b; int stage; bool extra; };
    unsigned profMask = 0;
    std::vector<ProfEvent> profPool;   // created lazily, reused
    size_t profUsed = 0;

    // saved fused-forward state
    struct {
        bool valid = false;
        bool blendBackwardDone = false;   // between gs_render_backward_dp_begin and _finish
        bool consumed = false;            // gs_render_backward_adam has overwritten the parameters: no further backward
        uint32_t* blockWork = nullptr;    // the buffer the forward measured into
        int N = 0, K = 0;
        const float *xyz = nullptr, *fdc = nullptr, *frest = nullptr, *scales = nullptr, *rot = nullptr,
                    *opacity = nullptr;
        const float *outColor = nullptr, *outDepth = nullptr, *outAlpha = nullptr;
        gs::CamParams cam;
        uint32_t* cutStore = nullptr;  // the view's cut words at the time of this forward (nullptr: none kept)
        bool cutsActive = false;     // this forward binned under depth cuts
        bool missChecked = true;     // ... and gs_forward_missed has been asked since
        bool bwdPrepared = false;    // the loss kernel has built the backward's item list and cleared its accumulator
        uint32_t preparedQueueStart = 0;
        int preparedN = 0;
        int statePlanes = 5;         // planes per checkpoint slot this forward wrote (5 with the depth sum, 4 without)
        uint32_t qslotCap = 0;       // ... and how many such slots the arena held for it
        uint32_t qslotStatic = 0;    // ... of which handed to the waves up front (the shared counter starts behind them)
        bool arenaOverflow = false;  // the host has been told that this forward ran out of checkpoint slots: no backward
        bool missed = false;         // ... and the answer was yes: its outputs are not final, no backward from it
    } fwd;
};

#define GS_HIP_CHECK(ctx, expr)                                                              \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                  \
            return GS_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

static inline int gs_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }
// does the depth sort of n records take the two-launch passes (binning.hip, radix_sort)?
static inline bool gs_small_depth_sort(long long n) { return gs_div_up(n, GS_SORT_THREADS * GS_SMALL_SORT_ITEMS) <= GS_SMALL_SORT_BLOCKS; }

// RAII stage timer: records a start/stop event pair around a launch sequence when profiling is on
struct GsStageTimer {
    gs_ctx* c;
    int slot = -1;
    // extra: time that belongs to a stage another timer of the same call already counts (its `calls` stay one per call)
    GsStageTimer(gs_ctx* ctx, int stage, bool extra = false) : c(ctx)
    {
        if (!((c->profMask >> stage) & 1u)) return;
        if (c->profUsed == c->profPool.size()) {
            if (c->profPool.size() >= 16384) return;
            gs_ctx::ProfEvent e;
            if (hipEventCreate(&e.a) != hipSuccess) return;
            if (hipEventCreate(&e.b) != hipSuccess) { (void)hipEventDestroy(e.a); return; }
            c->profPool.push_back(e);
        }
        slot = (int)c->profUsed++;
        c->profPool[slot].stage = stage;
        c->profPool[slot].extra = extra;
        (void)hipEventRecord(c->profPool[slot].a, c->stream);
    }
    ~GsStageTimer()
    {
        if (slot >= 0) (void)hipEventRecord(c->profPool[slot].b, c->stream);
    }
};

// ---- internal launchers (defined in the .hip files) -------------------------
namespace gs {

CamParams make_cam(const gs_camera* cam, int W, int H);

// projection.hip
int launch_projection_forward(gs_ctx* c, int N, int K, const float* scales, const float* rot, const float* means3d,
                              const float* shs, const CamParams& cam, float* means2d, float* depths, float* color,
                              float* cov2d, float* conic, float* radii, float* rectMin, float* rectMax);
int launch_projection_backward(gs_ctx* c, int N, int K, const float* scales, const float* rot, const float* means3d,
                               const float* shs, const CamParams& cam, const float* cotDepths,
                               const float* cotMeans2d, const float* cotCov2d, const float* cotColor,
                               const float* cotConic, float* gScales, float* gRot, float* gMeans, float* gShs,
                               float* gCam);
int launch_projection_fused_forward(gs_ctx* c, int N, int K, const float* xyz, const float* fdc, const float* frest,
                                    const float* scales, const float* rot, const float* opacity,
                                    const CamParams& cam, float* radii);
int launch_projection_fused_backward(gs_ctx* c, int N, int K, const float* xyz, const float* fdc,
                                     const float* frest, const float* scales, const float* rot,
                                     const float* opacity, const CamParams& cam, float* gXyz, float* gFdc,
                                     float* gFrest, float* gScales, float* gRot, float* gOpacity, bool emitColorCot = false);
int launch_projection_fused_backward_adam(gs_ctx* c, int N, int K, const float* xyz, const float* fdc,
                                          const float* frest, const float* scales, const float* rot,
                                          const float* opacity, const CamParams& cam, const float* pBase, float* mBase,
                                          float* vBase, const float lr[6], float b1, float b2, float eps, float gscale);
bool depth_sort_takes_splitters(const gs_ctx* c, int N);      // binning.hip
int launch_colour_rest(gs_ctx* c);      // gs_rider.h: the colour units the binning kernels have not taken along
int launch_color_cot(gs_ctx* c, int N, float* out);
int launch_sh_grad_from_views(gs_ctx* c, int N, int K, int R, const float* xyz, const float* mgAll,
                              const float* camCentersHost, float* gFdc, float* gFrest);
int launch_sh_grad_from_views_adam(gs_ctx* c, int N, int K, int R, const float* xyz, const float* mgAll,
                                   const float* camCentersHost, const float* fdcParam, const float* frestParam,
                                   const float* pBase, float* mBase, float* vBase, float lrDc, float lrRest, float b1,
                                   float b2, float eps, float gscale);
int launch_projection_geom_backward(gs_ctx* c, int N, const float* xyz, const float* scales, const float* rot,
                                    const float* opacity, const CamParams& cam, float* gXyz, float* gScales, float* gRot,
                                    float* gOpacity, float* xyzOwn, float* colorCot = nullptr);
int launch_sh_views_dir_adam(gs_ctx* c, int N, int K, int R, const float* xyz, const float* mgAll, const float* camCentersHost,
                             const float* const* ownXyzHost, const float* fdcParam, const float* frestParam, const float* pBase,
                             float* mBase, float* vBase, float lrDc, float lrRest, float b1, float b2, float eps, float gscale,
                             float* xyzAdd);
int launch_pack11_to_12(gs_ctx* c, int N, const float* packed11);
int launch_pack_gaussians(gs_ctx* c, int N, const float* means2d, const float* conic, const float* color,
                          const float* opacity, const float* depths, float* packed11);

// binning.hip
int launch_bin_prep(gs_ctx* c, int N, const float* rectMin, const float* rectMax, const float* radii,
                    const float* depths);
int launch_binning(gs_ctx* c, int N, bool wantPlain);  // depth sort, scan, expand, tile sort, ranges
int ensure_plain_sorted(gs_ctx* c);
int launch_tile_counts(gs_ctx* c);
int launch_cut_super(gs_ctx* c, const uint32_t* cuts);   // the view's cuts reduced to the deepest cut of every 4 x 4 tiles (c->superCut)
int cut_super_width(const gs_ctx* c);
int launch_build_packed_tile_indices(gs_ctx* c, uint32_t B, int32_t* out);

// blend.hip
int launch_blend_forward(gs_ctx* c, float* outColor, float* outDepth, float* outAlpha, uint32_t* lastContrib);
int launch_blend_backward(gs_ctx* c, int N, const float* cotColor, const float* cotDepth, const float* cotAlpha,
                          const float* outAlpha, const uint32_t* lastContrib);
int launch_gradacc_to_packed11(gs_ctx* c, int N, float* gradPacked11);

// blend_v2.hip (fused fast path)
int launch_blend_forward_v2(gs_ctx* c, float* outColor, float* outDepth, float* outAlpha);
int launch_blend_backward_v2(gs_ctx* c, int N, const float* cotColor, const float* cotDepth, const float* cotAlpha,
                             const float* outColor, const float* outDepth, const float* outAlpha);
int blend_backward_v2_grid(const gs_ctx* c);
int blend_forward_v2_grid(const gs_ctx* c);
bool blend_forward_v2_pair_decide(const gs_ctx* c);

// ssim.hip
int launch_ssim_forward(gs_ctx* c, int H, int W, int C, int K, const float* img1, const float* img2,
                        const float* window, float* ssim, float* mu1, float* mu2, float* s1, float* s2, float* s12);
int launch_ssim_backward(gs_ctx* c, int H, int W, int C, int K, const float* gradOut, float gradOutConst,
                         const float* img1, const float* img2, const float* window, const float* mu1,
                         const float* mu2, const float* s1, const float* s2, const float* s12, float* g1, float* g2,
                         float l1Weight);
int launch_loss(gs_ctx* c, const float* render, const float* target, const float* renderDepth,
                const float* targetDepth, const unsigned char* depthMask, float lambdaDssim, float lambdaDepth,
                float* lossOut, float* cotColor, float* cotDepth);

// optim.hip
// densify.hip
int launch_accum_grad_norm(gs_ctx* c, int N, const float* xyzGrad, const float* accumIn, float* accumOut);
int launch_classify(gs_ctx* c, int N, const float* gradAccum, float denom, const float* scales, int scaleStride,
                    const float* opacity, float gradThreshold, float maxScale, float minOpacity, int allowDensify,
                    int* actions, int* outputCounts);
int launch_densify_offsets(gs_ctx* c, int N, const int* actions, const int* outputCounts, int* offsets,
                           long long stats[5]);
int launch_build_densify_map(gs_ctx* c, int N, const int* actions, const int* offsets, int total, int* gather,
                             int* noiseMode);
int launch_densify_gather(gs_ctx* c, int total, int K, const float* xyz, const float* fdc, const float* frest,
                          const float* scales, const float* rot, const float* opacity, const int* gather,
                          const int* noiseMode, const float* baseNoise, float* oXyz, float* oFdc, float* oFrest,
                          float* oScales, float* oRot, float* oOpacity);
int launch_densify_plan(gs_ctx* c, int N, const int* actions, const int* outputCounts, int* offsets);
int densify_plan_read(gs_ctx* c, int wait, long long stats[8], int* ready);
int launch_build_densify_map_planned(gs_ctx* c, int N, const int* actions, const int* offsets, int cap, int* gather, int* noiseMode);
int launch_densify_gather_planned(gs_ctx* c, int cap, int K, const float* xyz, const float* fdc, const float* frest,
                                  const float* scales, const float* rot, const float* opacity, const int* gather,
                                  const int* noiseMode, unsigned long long noiseSeed, float* oXyz, float* oFdc, float* oFrest,
                                  float* oScales, float* oRot, float* oOpacity);
int launch_densify_gather_planned_packed(gs_ctx* c, int cap, int K, const float* xyz, const float* fdc, const float* frest,
                                         const float* scales, const float* rot, const float* opacity, const int* gather,
                                         const int* noiseMode, unsigned long long noiseSeed, float* outBase, const int order[6]);
int launch_densify_noise(gs_ctx* c, unsigned long long seed, int rows, float* out);
// knn.hip
int launch_dist_topk(gs_ctx* c, int N, int k, int qBegin, int qCount, const float* xyz, float* out);
// ply.hip
int launch_ply_pack(gs_ctx* c, int N, int K, const float* xyz, const float* fdc, const float* frest,
                    const float* opacity, const float* scales, const float* rot, float* rows);
int ply_write_file(gs_ctx* c, const char* path, int N, int K, const float* xyz, const float* fdc, const float* frest,
                   const float* opacity, const float* scales, const float* rot);
int ply_probe_file(gs_ctx* c, const char* path, long long* N, int* M, int* D);
int ply_load_file(gs_ctx* c, const char* path, int N, int K, float* xyz, float* fdc, float* frest, float* opacity,
                  float* scales, float* rot);
int launch_adam(gs_ctx* c, long long n, float* params, const float* grads, float* m, float* v, int nseg,
                const long long* segEnd, const float* segLr, float b1, float b2, float eps, float gradScale,
                const float* add = nullptr, long long addN = 0);

}  // namespace gs
