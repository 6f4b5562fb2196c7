// api.hip -- C ABI (include/gsplat.h): context, workspace, argument checks, launch sequencing.
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "gs_ctx.h"
#include <vector>

using namespace gs;

namespace {

int fail(gs_ctx* c, int code, const char* msg)
{
    if (c) c->err = msg;
    return code;
}

template <class T>
int dev_alloc(gs_ctx* c, T** p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    GS_HIP_CHECK(c, hipMalloc((void**)p, count * sizeof(T)));
    c->ws_bytes += count * sizeof(T);
    return GS_OK;
}

template <class T>
void dev_free(T*& p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

void free_gaussian_ws(gs_ctx* c)
{
    dev_free(c->packed12); dev_free(c->gradAcc16);
    dev_free(c->depthKey[0]); dev_free(c->depthKey[1]); dev_free(c->depthVal[0]); dev_free(c->depthVal[1]);
    dev_free(c->tilesTouched); dev_free(c->tileRect); dev_free(c->tilePieces); dev_free(c->waveSeg); dev_free(c->scanPrefix); dev_free(c->scanTmp); dev_free(c->blockSums);
    dev_free(c->visPerBlock); dev_free(c->dropPerBlock);
    dev_free(c->bucketId);
    dev_free(c->densifyTiles);
    c->densifyTileCap = 0;
}

void free_pair_ws(gs_ctx* c)
{
    // (the checkpoint arena, segState, is not part of it: ensure_arena / gs_ctx_destroy)
    dev_free(c->pairKey[0]); dev_free(c->pairKey[1]); dev_free(c->pairVal[0]); dev_free(c->pairVal[1]);
    dev_free(c->segSlot); dev_free(c->itemBlock); dev_free(c->itemRow);
}

// The checkpoint arena of the fused blend (blend_v2.hip): slots are taken as they are written, so the arena is sized
// from what forwards use, not from every list being swept to its end.  Reserved capacity: one 8x8-quadrant slot per 80
// reserved pairs (the bench scene writes one per ~100 pairs binned, the 100 k / 800x800 config one per ~30, against
// reserves of 3x and 5x their pairs; gs_ctx_reserve regrows to 1.5x the need after an overflow).  Without a reserve nothing may overflow silently between two host checks, so the arena holds the full bound.
int ensure_arena(gs_ctx* c)
{
    if (!c->fast16 || c->capM <= 0) return GS_OK;
    long long want = (c->pairsReserved || c->reserving) ? c->capM / 80 : c->segCap * 4;
    if (want < 65536) want = 65536;
    if (want > c->segCap * 4) want = c->segCap * 4;
    if (want < c->qslotWanted) want = c->qslotWanted;
    if (want <= c->qslotCap) return GS_OK;
    GS_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (c->segState) { c->ws_bytes -= (size_t)c->qslotCap * 5 * 64 * sizeof(float); dev_free(c->segState); }
    const int rc = dev_alloc(c, &c->segState, (size_t)want * 5 * 64);
    if (rc) { c->qslotCap = 0; return rc; }
    c->qslotCap = want;
    c->fwd.valid = false;
    return GS_OK;
}

// grows the workspace to hold N Gaussians and M pairs; synchronises only when it has to reallocate
int ensure_capacity(gs_ctx* c, int N, long long M)
{
    bool grewN = false, grewM = false;
    if (N > c->capN) {
        GS_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        free_gaussian_ws(c);
        const size_t n = (size_t)N;
        int rc;
        if ((rc = dev_alloc(c, &c->packed12, n * 12))) return rc;
        if ((rc = dev_alloc(c, &c->gradAcc16, n * 16))) return rc;
        for (int i = 0; i < 2; i++) {
            if ((rc = dev_alloc(c, &c->depthKey[i], n))) return rc;
            if ((rc = dev_alloc(c, &c->depthVal[i], n))) return rc;
        }
        if ((rc = dev_alloc(c, &c->tilesTouched, n))) return rc;
        if ((rc = dev_alloc(c, &c->tileRect, n))) return rc;
        if ((rc = dev_alloc(c, &c->tilePieces, n))) return rc;
        if ((rc = dev_alloc(c, &c->waveSeg, (n / 64 + 8) * GS_EXPAND_SLICES))) return rc;
        if ((rc = dev_alloc(c, &c->scanPrefix, (n / 64 + 16) * GS_EXPAND_SLICES))) return rc;
        if ((rc = dev_alloc(c, &c->scanTmp, (n / 64 + 16) * GS_EXPAND_SLICES / 1024 + 8))) return rc;
        const size_t nb = n / GS_SCAN_BLOCK + 2;
        if ((rc = dev_alloc(c, &c->blockSums, nb))) return rc;
        if ((rc = dev_alloc(c, &c->visPerBlock, n / 128 + 2))) return rc;
        dev_free(c->dropPerBlock);
        if ((rc = dev_alloc(c, &c->dropPerBlock, n / 128 + 2))) return rc;
        if ((rc = dev_alloc(c, &c->bucketId, n + 16))) return rc;
        c->capN = N;
        grewN = true;
    }
    if (M > c->capM) {
        GS_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        free_pair_ws(c);
        int rc;
        for (int i = 0; i < 2; i++) {
            if ((rc = dev_alloc(c, &c->pairKey[i], (size_t)M))) return rc;
            if ((rc = dev_alloc(c, &c->pairVal[i], (size_t)M))) return rc;
        }
        if (c->fast16) {
            // saved-state slots: sum over pixel blocks of ceil(list/SEG)-1 <= (pairs seen by pixel blocks)/SEG
            const long long blocksPerTile = (long long)(c->tileW / 16) * (c->tileH / 16);
            c->segCap = M / GS_SEG_LEN * blocksPerTile + 16;
            c->itemCap = c->segCap + c->numPixBlocks;
            if ((rc = dev_alloc(c, &c->segSlot, (size_t)c->segCap * 4))) return rc;
            // (entries are only ever read where this forward wrote them; zeroed once so that a backward run on a
            // forward that overflowed -- host errors off -- reads slot 0, never an address out of bounds)
            GS_HIP_CHECK(c, hipMemset(c->segSlot, 0, (size_t)c->segCap * 4 * sizeof(uint32_t)));
            if ((rc = dev_alloc(c, &c->itemBlock, (size_t)c->itemCap))) return rc;
            if ((rc = dev_alloc(c, &c->itemRow, (size_t)c->itemCap))) return rc;
        }
        c->capM = M;
        grewM = true;
    }
    if (grewN || grewM) {
        const long long big = c->capM > c->capN ? c->capM : c->capN;
        const int nb = gs_div_up(big, GS_SORT_TILE) + 1;
        if (nb > c->nbCap) {
            dev_free(c->hist); dev_free(c->wideCnt); dev_free(c->wideChunk);
            int rc = dev_alloc(c, &c->hist, (size_t)256 * nb);
            if (rc) return rc;
            if (c->T <= GS_WIDE_BINS) {      // the one-pass tile sort's count tables (binning.hip)
                if ((rc = dev_alloc(c, &c->wideCnt, (size_t)GS_WIDE_BINS * nb))) return rc;
                if ((rc = dev_alloc(c, &c->wideChunk, (size_t)GS_WIDE_BINS * (nb / GS_WIDE_CHUNK + 2)))) return rc;
            }
            c->nbCap = nb;
        }
        c->binValid = false;
        c->fwd.valid = false;
    }
    return ensure_arena(c);
}

long long default_pair_capacity(const gs_ctx* c, int N)
{
    long long m = (long long)N * 12 + 4LL * c->T + 65536;
    return m;
}

int zero_counters(gs_ctx* c)
{
    GS_HIP_CHECK(c, hipMemsetAsync(c->counters, 0, sizeof(uint32_t) * GS_CNT_COUNT, c->stream));
    return GS_OK;
}

// [sync] pulls the counters to the host
int read_counters(gs_ctx* c)
{
    GS_HIP_CHECK(c, hipMemcpyAsync(c->countersHost, c->counters, sizeof(uint32_t) * GS_CNT_COUNT,
                                   hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return GS_OK;
}

int overflow_error(gs_ctx* c, uint32_t need)
{
    char buf[260];
    if (c->missHost && c->missHost[4] == 2u) {
        snprintf(buf, sizeof buf, "the fused forward ran out of checkpoint slots (%lld held): its image is complete, but no "
                 "backward can be taken from it and no optimizer step was; call gs_ctx_reserve (it regrows the arena)", c->qslotCap);
        c->err = buf;
        return GS_ERR_WORKSPACE_OVERFLOW;
    }
    snprintf(buf, sizeof buf, "tile-splat pairs M=%u exceed the reserved capacity %lld: that forward rendered nothing and "
             "no optimizer step was taken from it; call gs_ctx_reserve", need, c->capM);
    c->err = buf;
    return GS_ERR_WORKSPACE_OVERFLOW;
}
int overflow_error(gs_ctx* c) { return overflow_error(c, c->countersHost[GS_CNT_MREQ]); }

// Deferred overflow of a forward under reserved capacity (no host check of M per call): the expansion kernel raises
// two words in mapped host memory.  Looked at -- without waiting -- by every entry point that continues a training
// step, so the error surfaces at the first call after the device got there; the device-side gate (adamGate) has kept
// the parameters untouched meanwhile.  Stays raised until gs_sync has reported it or gs_ctx_reserve has fixed it.
int deferred_overflow(gs_ctx* c)
{
    if (c->hostOverflowErrors && c->missHost && c->missHost[4]) return overflow_error(c, c->missHost[5]);
    return GS_OK;
}

// runs the binning pipeline; in auto-capacity mode it checks M on the host and regrows once
template <class Prep>
int bin_with_capacity(gs_ctx* c, int N, bool reserved, bool wantPlain, Prep&& prep)
{
    int rc;
    if ((rc = ensure_capacity(c, N, c->capM > 0 ? c->capM : default_pair_capacity(c, N)))) return rc;
    for (int attempt = 0; attempt < 2; attempt++) {
        if (N == 0 && (rc = zero_counters(c))) return rc;      // otherwise the prep kernel's first block clears them
        {
            GsStageTimer t(c, GS_STAGE_PROJ_FWD);
            if ((rc = prep())) return rc;
        }
        {
            GsStageTimer t(c, GS_STAGE_BIN);
            if ((rc = launch_binning(c, N, wantPlain))) return rc;
        }
        if (reserved) break;
        if ((rc = read_counters(c))) return rc;
        if (!c->countersHost[GS_CNT_OVERFLOW]) break;
        if (attempt == 1) return overflow_error(c);
        c->missHost[4] = 0;      // handled here (the wait above is behind the kernel that raised it): regrow and repeat
        const long long need = (long long)c->countersHost[GS_CNT_MREQ];
        if ((rc = ensure_capacity(c, N, need + need / 2 + 1024))) return rc;
    }
    c->binValid = true;
    c->binN = N;
    return GS_OK;
}

// waits for a forward that ran under depth cuts and records whether it missed (the wait is behind the forward only)
int settle_cut_forward(gs_ctx* c)
{
    if (!c->fwd.cutsActive || c->fwd.missChecked) return GS_OK;
    // busy-wait: the answer unblocks the launches of the rest of the step, and a blocking wait wakes up too late
    // (~0.1 ms) to keep the queue behind the loss kernel filled
    for (;;) {
        const hipError_t q = hipEventQuery(c->fwdDone);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) GS_HIP_CHECK(c, q);
    }
    c->fwd.missChecked = true;
    c->fwd.missed = c->missHost[0] != 0u;
    return GS_OK;
}

// what every backward entry point asks first: a usable forward (not consumed, not overflowed, and -- under depth
// cuts -- one that did not miss; a caller that never asked gs_forward_missed is answered here)
int backward_preflight(gs_ctx* c, const char* who, bool wantsDepth)
{
    if (!c->fwd.valid || c->fwd.consumed) {
        c->err = std::string(who) + ": no gs_render_forward on this context";
        return GS_ERR_NO_FORWARD;
    }
    int rc = deferred_overflow(c);
    if (rc) return rc;
    if (c->fwd.arenaOverflow) {
        c->err = std::string(who) + ": the forward ran out of checkpoint slots (reported by gs_sync): call gs_ctx_reserve and "
                 "repeat it";
        return GS_ERR_WORKSPACE_OVERFLOW;
    }
    if ((rc = settle_cut_forward(c))) return rc;
    if (c->fast16 && c->fwd.statePlanes == 0) {
        c->err = std::string(who) + ": the forward ran render-only (GS_TUNE_RENDER_ONLY): it kept nothing a backward could start from";
        return GS_ERR_NO_FORWARD;
    }
    if (wantsDepth && (!c->fwd.outDepth || (c->fast16 && c->fwd.statePlanes != 5))) {
        c->err = std::string(who) + ": cot_depth given, but the forward took no depth image (out_depth NULL) or ran with "
                 "GS_TUNE_DEPTH_GRADIENT off (no depth checkpoints)";
        return GS_ERR_INVALID_ARG;
    }
    if (c->fwd.missed) {
        c->err = std::string(who) + ": the forward ran under depth cuts and missed (gs_forward_missed): repeat it "
                 "with gs_set_depth_cuts(ctx, 0) first";
        return GS_ERR_NO_FORWARD;
    }
    return GS_OK;
}

// gs_copy_overflow_flag: a one-thread kernel, not hipMemcpyAsync -- the runtime's device-to-device copy is a blit kernel
// between two barrier packets (5.6 us + ~7 us of idle in front of the blend backward of every data-parallel step of the
// torch exchange, tools/trace_gaps.py)
__global__ void copy_word_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst) { *dst = *src; }

// The op-level entry points see the caller's tile grid; under block lists (gs_ctx.h) the context's own fields describe the
// fused path's block grid, so they are swapped for the duration of the call.
struct RealGeomScope {
    gs_ctx* c;
    GsRealGeom saved;
    GsVirtGeom virt;
    explicit RealGeomScope(gs_ctx* ctx) : c(ctx)
    {
        if (!c || !c->virt.nbx) { c = nullptr; return; }
        saved.tileW = c->tileW; saved.tileH = c->tileH; saved.gridW = c->gridW; saved.gridH = c->gridH; saved.T = c->T;
        saved.tileBits = c->tileBits; saved.fast16 = c->fast16;
        virt = c->virt;
        c->tileW = c->real.tileW; c->tileH = c->real.tileH; c->gridW = c->real.gridW; c->gridH = c->real.gridH; c->T = c->real.T;
        c->tileBits = c->real.tileBits; c->fast16 = c->real.fast16;
        c->virt = GsVirtGeom();
    }
    ~RealGeomScope()
    {
        if (!c) return;
        c->tileW = saved.tileW; c->tileH = saved.tileH; c->gridW = saved.gridW; c->gridH = saved.gridH; c->T = saved.T;
        c->tileBits = saved.tileBits; c->fast16 = saved.fast16;
        c->virt = virt;
    }
};

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

int gs_abi_version(void) { return GSPLAT_ABI_VERSION; }

int gs_ctx_create(int device, int W, int H, int tile_w, int tile_h, int sh_degree, int white_bg, gs_ctx** out)
{
    if (!out) return GS_ERR_INVALID_ARG;
    *out = nullptr;
    if (W <= 0 || H <= 0 || tile_w <= 0 || tile_h <= 0 || sh_degree < 0 || sh_degree > 4) return GS_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return GS_ERR_NO_DEVICE;
    gs_ctx* c = new gs_ctx();
    c->device = device;
    c->W = W; c->H = H; c->tileW = tile_w; c->tileH = tile_h;
    c->gridW = (W + tile_w - 1) / tile_w; c->gridH = (H + tile_h - 1) / tile_h;
    c->T = c->gridW * c->gridH;
    c->degree = sh_degree; c->whiteBg = white_bg ? 1 : 0;
    c->fast16 = (tile_w % 16 == 0) && (tile_h % 16 == 0);
    c->blocksX = gs_div_up(W, 16); c->blocksY = gs_div_up(H, 16);
    c->real.tileW = tile_w; c->real.tileH = tile_h; c->real.gridW = c->gridW; c->real.gridH = c->gridH; c->real.T = c->T;
    c->real.fast16 = c->fast16;
    {
        // Block lists (gs_ctx.h, GsVirtGeom): for a tile size that is not a multiple of 16 the fused path works on the grid
        // of 16 x 16 blocks enumerated per tile.  GSPLAT_BLOCK_LISTS=0 keeps round 3's form (tile lists + the generic blend
        // kernels that scan a tile's list per block) for A/B runs.
        const char* e = getenv("GSPLAT_BLOCK_LISTS");
        const int nbx = gs_div_up(tile_w, 16), nby = gs_div_up(tile_h, 16);
        if (!c->fast16 && !(e && atoi(e) == 0) && (long long)c->gridW * nbx <= 65535 && (long long)c->gridH * nby <= 65535) {
            c->virt.nbx = nbx; c->virt.nby = nby; c->virt.tw = tile_w; c->virt.th = tile_h;
            c->tileW = 16; c->tileH = 16;
            c->gridW *= nbx; c->gridH *= nby;
            c->T = c->gridW * c->gridH;
            c->fast16 = true;
            c->blocksX = c->gridW; c->blocksY = c->gridH;
        }
    }
    if (const char* e = getenv("GSPLAT_COLOUR_RIDERS")) c->colourRiders = atoi(e);      // tuning experiments (tools/rider_ab.py)
    if (const char* e = getenv("GSPLAT_FWD_WIDE")) c->fwdWide = atoi(e) < 0 ? -1 : atoi(e) != 0;
    if (const char* e = getenv("GSPLAT_RANK_SORT")) c->rankSort = atoi(e) != 0;
    if (const char* e = getenv("GSPLAT_FWD_SPATIAL")) c->fwdSpatial = atoi(e) != 0;
    if (const char* e = getenv("GSPLAT_LSD_THREADS")) { const int v = atoi(e); c->lsdThreads = (v == 256 || v == 1024) ? v : 0; }
    if (const char* e = getenv("GSPLAT_SCATTER_THREADS")) { const int v = atoi(e); c->scatterThreads = (v == 256 || v == 512 || v == 1024) ? v : 0; }
    if (const char* e = getenv("GSPLAT_BWD_QUEUES")) { const int q = atoi(e); if (q == 1 || q == 2 || q == 4 || q == 8) c->bwdQueues = q; }
    if (const char* e = getenv("GSPLAT_FWD_QUEUES")) { const int q = atoi(e); if (q == 1 || q == 2 || q == 4 || q == 8) c->fwdQueues = q; }
    if (const char* e = getenv("GSPLAT_RIDER_SHARES")) {      // tuning experiments: permille of the colour units per host kernel
        // exactly GS_RIDE_HOSTS comma-separated values, in the enum's order (ss_hist, ss_scatter, wide_tile); anything else
        // is refused loudly -- a sweep that passes four values would measure other splits than the ones it prints
        int vals[GS_RIDE_HOSTS + 1], n = 0;
        for (const char* q = e; *q && n <= GS_RIDE_HOSTS; n++) {
            vals[n] = atoi(q);
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
        }
        if (n == GS_RIDE_HOSTS) for (int i = 0; i < GS_RIDE_HOSTS; i++) c->riderShare[i] = vals[i];
        else fprintf(stderr, "gsplat: GSPLAT_RIDER_SHARES needs %d comma-separated values (got %s): ignored\n", GS_RIDE_HOSTS, e);
    }
    if (c->gridW > 65535 || c->gridH > 65535) { delete c; return GS_ERR_INVALID_ARG; }
    int bits = 1;
    while ((1LL << bits) < c->T) bits++;
    c->tileBits = bits;
    for (bits = 1; (1LL << bits) < c->real.T; bits++) {}
    c->real.tileBits = bits;
    auto bail = [&](int code) { gs_ctx_destroy(c); return code; };
    if (hipSetDevice(device) != hipSuccess) return bail(GS_ERR_HIP);
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) return bail(GS_ERR_HIP);
    c->stream = c->own_stream;
    const size_t P = (size_t)W * H;
    c->numPixBlocks = c->blocksX * c->blocksY;
    c->opBlocks = c->real.fast16 ? c->numPixBlocks : c->real.T * gs_div_up(tile_w, 16) * gs_div_up(tile_h, 16);
    const size_t maxBlocks = (size_t)(c->opBlocks > c->numPixBlocks ? c->opBlocks : c->numPixBlocks);
    if (dev_alloc(c, &c->blockWorkOwn, maxBlocks) || dev_alloc(c, &c->blockOrder, maxBlocks + 8) || dev_alloc(c, &c->fwdQueue, 8 * 32) || dev_alloc(c, &c->bwdQueue, 8 * 32) ||
        dev_alloc(c, &c->segBase, (size_t)c->numPixBlocks) || dev_alloc(c, &c->finalT, P))
        return bail(GS_ERR_HIP);
    c->blockWork = c->blockWorkOwn;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            c->numCUs = prop.multiProcessorCount;
    }
    if (const char* e = getenv("GSPLAT_CUT_SUPER")) c->cutSuper = atoi(e) != 0;
    if (dev_alloc(c, &c->superCut, (size_t)c->T + 16) || dev_alloc(c, &c->tileRanges, (size_t)c->T * 2) || dev_alloc(c, &c->tileCounts, (size_t)c->T) ||
        dev_alloc(c, &c->lastContrib, P) ||
        dev_alloc(c, &c->lossPartials, (size_t)(c->lossPartialBlocks = gs_div_up(W, 16) * gs_div_up(H, 16) * 3) * 4 + 16) || dev_alloc(c, &c->windowDev, 121) ||
        dev_alloc(c, &c->counters, GS_CNT_COUNT) || dev_alloc(c, &c->rowTotal, 256) ||
        dev_alloc(c, &c->sortBits, GS_SMALL_SORT_BLOCKS) || dev_alloc(c, &c->wideTotal, GS_WIDE_BINS) ||
        dev_alloc(c, &c->bucketStart, 520) || dev_alloc(c, &c->sortSplit[0], 256) || dev_alloc(c, &c->sortSplit[1], 256) ||
        dev_alloc(c, &c->ssChunk, 65 * 512))
        return bail(GS_ERR_HIP);
    if (hipHostMalloc((void**)&c->countersHost, sizeof(uint32_t) * GS_CNT_COUNT) != hipSuccess) return bail(GS_ERR_HIP);
    // the depth cuts' miss word: host memory the forward kernel writes directly, read after the fwdDone event
    if (hipHostMalloc((void**)&c->missHost, 64, hipHostMallocMapped) != hipSuccess) return bail(GS_ERR_HIP);
    for (int i = 0; i < 16; i++) c->missHost[i] = 0;
    if (hipHostGetDevicePointer((void**)&c->missDev, c->missHost, 0) != hipSuccess) return bail(GS_ERR_HIP);
    if (hipEventCreateWithFlags(&c->fwdDone, hipEventDisableTiming) != hipSuccess) return bail(GS_ERR_HIP);
    float win[121];
    gs_ssim_window(11, 1.5f, win);
    if (hipMemcpy(c->windowDev, win, sizeof win, hipMemcpyHostToDevice) != hipSuccess) return bail(GS_ERR_HIP);
    if (hipMemset(c->counters, 0, sizeof(uint32_t) * GS_CNT_COUNT) != hipSuccess) return bail(GS_ERR_HIP);
    if (hipMemset(c->tileRanges, 0, sizeof(uint32_t) * 2 * c->T) != hipSuccess) return bail(GS_ERR_HIP);
    c->adamGate = c->counters + GS_CNT_OVERFLOW;
    *out = c;
    return GS_OK;
}

int gs_ctx_destroy(gs_ctx* c)
{
    if (!c) return GS_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)gs_dp_shutdown(c);
    free_gaussian_ws(c);
    free_pair_ws(c);
    dev_free(c->segState);
    dev_free(c->hist); dev_free(c->wideCnt); dev_free(c->wideChunk); dev_free(c->wideTotal); dev_free(c->rowTotal); dev_free(c->sortBits); dev_free(c->bucketStart); dev_free(c->ssChunk); dev_free(c->sortSplit[0]); dev_free(c->sortSplit[1]); dev_free(c->tileRanges); dev_free(c->tileCounts); dev_free(c->superCut);
    dev_free(c->lastContrib); dev_free(c->lossPartials); dev_free(c->windowDev);
    dev_free(c->counters); dev_free(c->blockWorkOwn); dev_free(c->blockOrder); dev_free(c->fwdQueue); dev_free(c->bwdQueue); dev_free(c->segBase); dev_free(c->finalT);
    for (auto& e : c->profPool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (c->countersHost) (void)hipHostFree(c->countersHost);
    if (c->missHost) (void)hipHostFree(c->missHost);
    if (c->fwdDone) (void)hipEventDestroy(c->fwdDone);
    if (c->densifyDone) (void)hipEventDestroy(c->densifyDone);
    if (c->densifyPlanHost) (void)hipHostFree(c->densifyPlanHost);
    dev_free(c->densifyPlan);
    dev_free(c->densifyTable);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return GS_OK;
}

int gs_ctx_set_stream(gs_ctx* c, void* hip_stream)
{
    if (!c) return GS_ERR_INVALID_ARG;
    GS_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    c->stream = (hipStream_t)hip_stream;
    return GS_OK;
}

int gs_ctx_reserve(gs_ctx* c, int max_gaussians, long long max_pairs)
{
    if (!c || max_gaussians < 0 || max_pairs < 0) return fail(c, GS_ERR_INVALID_ARG, "gs_ctx_reserve: negative size");
    (void)hipSetDevice(c->device);
    GS_HIP_CHECK(c, hipStreamSynchronize(c->stream));      // a pending overflow report lands before it is cleared
    if (c->missHost[4] == 2u || c->arenaRegrowPending) {
        // the last forward ran out of checkpoint slots: its waves kept counting, so static part + counter is the need
        uint32_t parts[8], used = 0;
        GS_HIP_CHECK(c, hipMemcpy(parts, c->counters + GS_CNT_QSLOTS, sizeof parts, hipMemcpyDeviceToHost));
        // (what the waves drew from the eight parts together.  A draw that does not fit its part is given back (atomicSub),
        // and once every part is empty a wave keeps counting on its own part what it would have drawn, so the sum is what
        // was drawn + what was wanted and not had.  Between a failed draw's add and its sub another wave can see the
        // counter inflated and take a part that still has room for empty -- for good: partsEmpty is sticky per wave.  The
        // cost is a regrow a little early on an arena that is nearly full, never a wrong result; accepted.)
        for (uint32_t x : parts) used += x;
        // (counted in slots of that forward's planes; the arena is sized in five-plane slots)
        const long long need5 = (((long long)used + c->fwd.qslotStatic) * c->fwd.statePlanes + 4) / 5;
        c->qslotWanted = need5 + need5 / 2 + 4096;
        // the counters are the LAST forward's, which need not be the one that ran out (a report may be several forwards
        // old by the time the host acts on it): a pending regrow always grows the arena, by half at least
        const long long half = c->qslotCap + c->qslotCap / 2;
        if (c->qslotWanted < half) c->qslotWanted = half;
    }
    c->reserving = max_pairs > 0;      // (sizes the checkpoint arena for a reserve, ensure_arena)
    int rc = ensure_capacity(c, max_gaussians, max_pairs);
    if (rc == GS_OK) rc = ensure_arena(c);
    c->reserving = false;
    if (rc == GS_OK) {
        // (only now: a ctx whose allocation failed must not believe in a reserve it does not hold)
        if (max_pairs > 0) c->pairsReserved = true;
        c->missHost[4] = 0; c->arenaRegrowPending = false;
    }
    return rc;
}

size_t gs_workspace_bytes(const gs_ctx* c) { return c ? c->ws_bytes : 0; }

int gs_sync(gs_ctx* c)
{
    if (!c) return GS_ERR_INVALID_ARG;
    const int rc = read_counters(c);
    if (rc) return rc;
    if (c->missHost[4]) {            // an earlier forward's overflow nobody has been told about yet
        const uint32_t need = c->missHost[5];
        const int orc = overflow_error(c, need);
        if (c->missHost[4] == 2u) { c->arenaRegrowPending = true; c->fwd.arenaOverflow = true; }
        c->missHost[4] = 0;
        return orc;
    }
    if (c->countersHost[GS_CNT_OVERFLOW]) return overflow_error(c);
    return GS_OK;
}

int gs_overflow_pending(gs_ctx* c, uint32_t out[2])
{
    if (!c || !out) return GS_ERR_INVALID_ARG;
    out[0] = c->missHost ? c->missHost[4] : 0u;
    out[1] = c->missHost ? c->missHost[5] : 0u;
    if (out[0] == 0u && c->arenaRegrowPending) out[0] = 2u;
    return GS_OK;
}

int gs_wait(gs_ctx* c)
{
    if (!c) return GS_ERR_INVALID_ARG;
    GS_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return GS_OK;
}

const char* gs_last_error(const gs_ctx* c) { return c ? c->err.c_str() : "null context"; }

// ---- projection -------------------------------------------------------------------------------
int gs_projection_forward(gs_ctx* c, int N, int K, const float* scales, const float* rotations, const float* means3d,
                          const float* shs, const gs_camera* cam, float* means2d, float* depths, float* color,
                          float* cov2d, float* conic, float* radii, float* rect_min, float* rect_max)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (N < 0 || K < 1 || !cam) return fail(c, GS_ERR_INVALID_ARG, "gs_projection_forward: bad N/K/cam");
    if ((c->degree + 1) * (c->degree + 1) > K) return fail(c, GS_ERR_SIZE_MISMATCH, "K smaller than (degree+1)^2");
    if (N > 0 && (!scales || !rotations || !means3d || !shs || !means2d || !depths || !color || !cov2d || !conic ||
                  !radii || !rect_min || !rect_max))
        return fail(c, GS_ERR_INVALID_ARG, "gs_projection_forward: null buffer");
    return launch_projection_forward(c, N, K, scales, rotations, means3d, shs, make_cam(cam, c->W, c->H), means2d,
                                     depths, color, cov2d, conic, radii, rect_min, rect_max);
}

int gs_projection_backward(gs_ctx* c, int N, int K, const float* scales, const float* rotations,
                           const float* means3d, const float* shs, const gs_camera* cam, const float* cot_depths,
                           const float* cot_means2d, const float* cot_cov2d, const float* cot_color,
                           const float* cot_conic, float* grad_scales, float* grad_rotations, float* grad_means3d,
                           float* grad_shs, float* grad_cam_center_point)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (N < 0 || K < 1 || !cam) return fail(c, GS_ERR_INVALID_ARG, "gs_projection_backward: bad N/K/cam");
    if ((c->degree + 1) * (c->degree + 1) > K) return fail(c, GS_ERR_SIZE_MISMATCH, "K smaller than (degree+1)^2");
    if (N > 0 && (!scales || !rotations || !means3d || !shs || !cot_depths || !cot_means2d || !cot_cov2d ||
                  !cot_color || !cot_conic || !grad_scales || !grad_rotations || !grad_means3d || !grad_shs ||
                  !grad_cam_center_point))
        return fail(c, GS_ERR_INVALID_ARG, "gs_projection_backward: null buffer");
    return launch_projection_backward(c, N, K, scales, rotations, means3d, shs, make_cam(cam, c->W, c->H), cot_depths,
                                      cot_means2d, cot_cov2d, cot_color, cot_conic, grad_scales, grad_rotations,
                                      grad_means3d, grad_shs, grad_cam_center_point);
}

// ---- binning ------------------------------------------------------------------------------------
int gs_tile_bin(gs_ctx* c, int N, const float* rect_min, const float* rect_max, const float* radii,
                const float* depths)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (N < 0 || (N > 0 && (!rect_min || !rect_max || !radii || !depths)))
        return fail(c, GS_ERR_INVALID_ARG, "gs_tile_bin: bad arguments");
    return gs_tile_bin_cut(c, N, rect_min, rect_max, radii, depths, nullptr);
}

int gs_tile_bin_cut(gs_ctx* c, int N, const float* rect_min, const float* rect_max, const float* radii,
                    const float* depths, const uint32_t* tile_cuts)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (N < 0 || (N > 0 && (!rect_min || !rect_max || !radii || !depths)))
        return fail(c, GS_ERR_INVALID_ARG, "gs_tile_bin: bad arguments");
    RealGeomScope real(c);
    c->binIsBlockLists = false;
    c->dropBlocks = 0; c->superCutReady = false;      // (no fused projection in front of this binning)
    c->fwd.valid = false;
    c->fwd.cutsActive = false;          // a fused forward's cuts never leak into an op-level binning
    c->fwd.bwdPrepared = false;
    c->opCuts = N > 0 ? tile_cuts : nullptr;
    const bool reserved = c->pairsReserved && c->capN >= N;
    const int rc = bin_with_capacity(c, N, reserved, true, [&]() { return launch_bin_prep(c, N, rect_min, rect_max, radii, depths); });
    c->opCuts = nullptr;
    return rc;
}

int gs_tile_bin_info(gs_ctx* c, uint32_t* M, uint32_t* B)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!c->binValid) return fail(c, GS_ERR_NO_FORWARD, "gs_tile_bin_info: no binning on this context");
    if (c->binIsBlockLists) return fail(c, GS_ERR_NO_FORWARD, "gs_tile_bin_info: the last binning on this context was a fused forward's block lists (tile size not a multiple of 16); call gs_tile_bin");
    RealGeomScope real(c);
    int rc = launch_tile_counts(c);
    if (rc) return rc;
    if ((rc = read_counters(c))) return rc;
    if (c->countersHost[GS_CNT_OVERFLOW]) return overflow_error(c);
    if (M) *M = c->countersHost[GS_CNT_M];
    if (B) *B = c->countersHost[GS_CNT_B];
    return GS_OK;
}

int gs_tile_bin_views(gs_ctx* c, const uint32_t** sorted_gauss_idx, const uint32_t** tile_ranges,
                      const uint32_t** tile_counts)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!c->binValid) return fail(c, GS_ERR_NO_FORWARD, "gs_tile_bin_views: no binning on this context");
    if (c->binIsBlockLists) return fail(c, GS_ERR_NO_FORWARD, "gs_tile_bin_views: the last binning on this context was a fused forward's block lists (tile size not a multiple of 16); call gs_tile_bin");
    RealGeomScope real(c);
    if (tile_counts) {
        const int rc = launch_tile_counts(c);
        if (rc) return rc;
        *tile_counts = c->tileCounts;
    }
    if (sorted_gauss_idx) {
        const int rc = ensure_plain_sorted(c);
        if (rc) return rc;
        *sorted_gauss_idx = c->sortedIdx;
    }
    if (tile_ranges) *tile_ranges = c->tileRanges;
    return GS_OK;
}

int gs_tile_bin_export(gs_ctx* c, uint32_t* sorted_gauss_idx, uint32_t* tile_ranges, uint32_t* tile_counts)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!c->binValid) return fail(c, GS_ERR_NO_FORWARD, "gs_tile_bin_export: no binning on this context");
    if (c->binIsBlockLists) return fail(c, GS_ERR_NO_FORWARD, "gs_tile_bin_export: the last binning on this context was a fused forward's block lists (tile size not a multiple of 16); call gs_tile_bin");
    RealGeomScope real(c);
    int rc;
    if ((rc = launch_tile_counts(c))) return rc;
    if (sorted_gauss_idx && (rc = ensure_plain_sorted(c))) return rc;
    if ((rc = read_counters(c))) return rc;
    if (c->countersHost[GS_CNT_OVERFLOW]) return overflow_error(c);
    const size_t M = c->countersHost[GS_CNT_M];
    if (sorted_gauss_idx && M)
        GS_HIP_CHECK(c, hipMemcpyAsync(sorted_gauss_idx, c->sortedIdx, M * 4, hipMemcpyDeviceToDevice, c->stream));
    if (tile_ranges)
        GS_HIP_CHECK(c, hipMemcpyAsync(tile_ranges, c->tileRanges, (size_t)c->T * 8, hipMemcpyDeviceToDevice, c->stream));
    if (tile_counts)
        GS_HIP_CHECK(c, hipMemcpyAsync(tile_counts, c->tileCounts, (size_t)c->T * 4, hipMemcpyDeviceToDevice, c->stream));
    return GS_OK;
}

int gs_build_packed_tile_indices(gs_ctx* c, uint32_t B, int32_t* out)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!c->binValid) return fail(c, GS_ERR_NO_FORWARD, "gs_build_packed_tile_indices: no binning on this context");
    if (c->binIsBlockLists) return fail(c, GS_ERR_NO_FORWARD, "gs_build_packed_tile_indices: the last binning on this context was a fused forward's block lists (tile size not a multiple of 16); call gs_tile_bin");
    RealGeomScope real(c);
    if (B > 0 && !out) return fail(c, GS_ERR_INVALID_ARG, "gs_build_packed_tile_indices: null output");
    const int rc = ensure_plain_sorted(c);
    if (rc) return rc;
    return launch_build_packed_tile_indices(c, B, out);
}

// ---- packing / blending ---------------------------------------------------------------------------
int gs_pack_gaussians(gs_ctx* c, int N, const float* means2d, const float* conic, const float* color,
                      const float* opacity, const float* depths, float* packed)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (N < 0 || (N > 0 && (!means2d || !conic || !color || !opacity || !depths || !packed)))
        return fail(c, GS_ERR_INVALID_ARG, "gs_pack_gaussians: bad arguments");
    return launch_pack_gaussians(c, N, means2d, conic, color, opacity, depths, packed);
}

int gs_blend_forward(gs_ctx* c, int N, const float* packed, float* out_color, float* out_depth, float* out_alpha,
                     uint32_t* last_contrib)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!c->binValid) return fail(c, GS_ERR_NO_FORWARD, "gs_blend_forward: call gs_tile_bin first");
    if (c->binIsBlockLists) return fail(c, GS_ERR_NO_FORWARD, "gs_blend_forward: the last binning on this context was a fused forward's block lists (tile size not a multiple of 16); call gs_tile_bin");
    RealGeomScope real(c);
    if (N != c->binN) return fail(c, GS_ERR_SIZE_MISMATCH, "gs_blend_forward: N differs from the binned N");
    if (!out_color || !out_depth || !out_alpha || !last_contrib || (N > 0 && !packed))
        return fail(c, GS_ERR_INVALID_ARG, "gs_blend_forward: null buffer");
    // the op-level blend overwrites the ctx's packed records (and, backward, its accumulator): whatever a fused forward
    // saved on this ctx -- incl. a backward preparation the loss kernel carried along -- is gone
    c->fwd.valid = false; c->fwd.bwdPrepared = false;
    int rc = launch_pack11_to_12(c, N, packed);
    if (rc) return rc;
    return launch_blend_forward(c, out_color, out_depth, out_alpha, last_contrib);
}

int gs_blend_backward(gs_ctx* c, int N, const float* packed, const float* cot_color, const float* cot_depth,
                      const float* cot_alpha, const float* out_color, const float* out_depth, const float* out_alpha,
                      const uint32_t* last_contrib, float* grad_packed)
{
    (void)out_color; (void)out_depth;   // undone by the reference but never read back
    if (!c) return GS_ERR_INVALID_ARG;
    if (!c->binValid) return fail(c, GS_ERR_NO_FORWARD, "gs_blend_backward: call gs_tile_bin first");
    if (c->binIsBlockLists) return fail(c, GS_ERR_NO_FORWARD, "gs_blend_backward: the last binning on this context was a fused forward's block lists (tile size not a multiple of 16); call gs_tile_bin");
    RealGeomScope real(c);
    if (N != c->binN) return fail(c, GS_ERR_SIZE_MISMATCH, "gs_blend_backward: N differs from the binned N");
    if (!cot_color || !out_alpha || !last_contrib || (N > 0 && (!packed || !grad_packed)))
        return fail(c, GS_ERR_INVALID_ARG, "gs_blend_backward: null buffer");
    c->fwd.valid = false; c->fwd.bwdPrepared = false;      // as gs_blend_forward
    int rc = launch_pack11_to_12(c, N, packed);
    if (rc) return rc;
    if ((rc = launch_blend_backward(c, N, cot_color, cot_depth, cot_alpha, out_alpha, last_contrib))) return rc;
    return launch_gradacc_to_packed11(c, N, grad_packed);
}

// ---- SSIM -----------------------------------------------------------------------------------------
int gs_ssim_window(int K, float sigma, float* window)
{
    if (K <= 0 || K > 32 || !window) return GS_ERR_INVALID_ARG;
    float g[32];
    const float center = (float)K / 2.0f;    // 5.5 for K = 11: the reference's off-centre window
    float sum = 0.0f;
    for (int x = 0; x < K; x++) {
        const float d = (float)x - center;
        g[x] = expf(-(d * d) / (2.0f * (sigma * sigma)));
        sum += g[x];
    }
    for (int x = 0; x < K; x++) g[x] = g[x] / sum;
    for (int i = 0; i < K; i++)
        for (int j = 0; j < K; j++) window[i * K + j] = g[i] * g[j];
    return GS_OK;
}

int gs_ssim_forward(gs_ctx* c, int H, int W, int C, int K, const float* img1, const float* img2, const float* window,
                    float* out_ssim, float* out_mu1, float* out_mu2, float* out_sigma1, float* out_sigma2,
                    float* out_sigma12)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (H < 0 || W < 0 || C < 0 || K <= 0 || K > 32) return fail(c, GS_ERR_INVALID_ARG, "gs_ssim_forward: bad shape");
    if (C > 65535) return fail(c, GS_ERR_INVALID_ARG, "gs_ssim_forward: too many channels");
    if ((size_t)H * W * C > 0 && (!img1 || !img2 || !window || !out_ssim || !out_mu1 || !out_mu2 || !out_sigma1 ||
                                  !out_sigma2 || !out_sigma12))
        return fail(c, GS_ERR_INVALID_ARG, "gs_ssim_forward: null buffer");
    return launch_ssim_forward(c, H, W, C, K, img1, img2, window, out_ssim, out_mu1, out_mu2, out_sigma1, out_sigma2,
                               out_sigma12);
}

int gs_ssim_backward(gs_ctx* c, int H, int W, int C, int K, const float* grad_out, const float* img1,
                     const float* img2, const float* window, const float* mu1, const float* mu2, const float* sigma1,
                     const float* sigma2, const float* sigma12, float* grad_img1, float* grad_img2)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (H < 0 || W < 0 || C < 0 || K <= 0 || K > 32) return fail(c, GS_ERR_INVALID_ARG, "gs_ssim_backward: bad shape");
    if ((size_t)H * W * C > 0 && (!grad_out || !img1 || !img2 || !window || !mu1 || !mu2 || !sigma1 || !sigma2 ||
                                  !sigma12 || !grad_img1 || !grad_img2))
        return fail(c, GS_ERR_INVALID_ARG, "gs_ssim_backward: null buffer");
    return launch_ssim_backward(c, H, W, C, K, grad_out, 0.0f, img1, img2, window, mu1, mu2, sigma1, sigma2, sigma12,
                                grad_img1, grad_img2, 0.0f);
}

// ---- fused path -------------------------------------------------------------------------------------
int gs_render_forward(gs_ctx* c, int N, int K, const float* xyz, const float* features_dc,
                      const float* features_rest, const float* scales, const float* rotation, const float* opacity,
                      const gs_camera* cam, float* out_color, float* out_depth, float* out_alpha, float* radii)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (N < 0 || K < 1 || !cam) return fail(c, GS_ERR_INVALID_ARG, "gs_render_forward: bad N/K/cam");
    if ((c->degree + 1) * (c->degree + 1) > K) return fail(c, GS_ERR_SIZE_MISMATCH, "K smaller than (degree+1)^2");
    // (out_depth may be NULL: no depth image is computed, and the backward of this forward takes no depth cotangent)
    if (!out_color || !out_alpha) return fail(c, GS_ERR_INVALID_ARG, "gs_render_forward: null output");
    if (N > 0 && (!xyz || !features_dc || (K > 1 && !features_rest) || !scales || !rotation || !opacity))
        return fail(c, GS_ERR_INVALID_ARG, "gs_render_forward: null parameter tensor");
    c->fwd.valid = false;
    c->binIsBlockLists = c->virt.nbx != 0;
    const bool reserved = c->pairsReserved && c->capN >= N;
    if (reserved) { const int orc = deferred_overflow(c); if (orc) return orc; }
    // a forward under depth cuts that nobody asked about: let its miss word settle before it is reused
    if (c->fwd.cutsActive && !c->fwd.missChecked) GS_HIP_CHECK(c, hipEventSynchronize(c->fwdDone));
    c->fwd.missed = false;
    c->fwd.bwdPrepared = false;
    c->fwd.arenaOverflow = false;
    c->fwd.cutStore = c->cutStore;
    c->fwd.cutsActive = c->cutStore != nullptr && c->allowCuts && N > 0;
    c->fwd.missChecked = !c->fwd.cutsActive;
    if (c->fwd.cutsActive) c->missHost[0] = 0;
    const CamParams cp = make_cam(cam, c->W, c->H);
    c->segBaseWanted = c->fast16 && N > 0;
    c->segBaseDone = false;
    c->fwdPairNow = c->fast16 && blend_forward_v2_pair_decide(c);
    int rc = bin_with_capacity(c, N, reserved, !c->fast16, [&]() {
        return launch_projection_fused_forward(c, N, K, xyz, features_dc, features_rest, scales, rotation, opacity, cp,
                                               radii);
    });
    c->segBaseWanted = false;
    if (rc) { c->rider.on = false; return rc; }
    if (c->rider.on && c->rider.next < c->rider.total) {
        GsStageTimer t(c, GS_STAGE_PROJ_FWD, true);      // the colour units still outstanding belong to the projection
        if ((rc = launch_colour_rest(c))) return rc;
    }
    c->rider.on = false;
    {
        GsStageTimer t(c, GS_STAGE_BLEND_FWD);
        rc = c->fast16 ? launch_blend_forward_v2(c, out_color, out_depth, out_alpha)
                       : launch_blend_forward(c, out_color, out_depth, out_alpha, c->lastContrib);
        if (rc) return rc;
    }
    if (c->fwd.cutsActive) GS_HIP_CHECK(c, hipEventRecord(c->fwdDone, c->stream));
    c->fwd.valid = true;
    c->fwd.blendBackwardDone = false;
    c->fwd.consumed = false;
    c->fwd.blockWork = c->blockWork;
    c->fwd.N = N; c->fwd.K = K;
    c->fwd.xyz = xyz; c->fwd.fdc = features_dc; c->fwd.frest = features_rest; c->fwd.scales = scales;
    c->fwd.rot = rotation; c->fwd.opacity = opacity;
    c->fwd.outColor = out_color; c->fwd.outDepth = out_depth; c->fwd.outAlpha = out_alpha;
    c->fwd.cam = cp;
    return GS_OK;
}

int gs_render_backward(gs_ctx* c, const float* cot_color, const float* cot_depth, const float* cot_alpha,
                       float* grad_xyz, float* grad_features_dc, float* grad_features_rest, float* grad_scales,
                       float* grad_rotation, float* grad_opacity)
{
    if (!c) return GS_ERR_INVALID_ARG;
    { const int prc = backward_preflight(c, "gs_render_backward", cot_depth != nullptr); if (prc) return prc; }
    const int N = c->fwd.N, K = c->fwd.K;
    if (!cot_color) return fail(c, GS_ERR_INVALID_ARG, "gs_render_backward: null cot_color");
    if (N > 0 && (!grad_xyz || !grad_features_dc || (K > 1 && !grad_features_rest) || !grad_scales || !grad_rotation ||
                  !grad_opacity))
        return fail(c, GS_ERR_INVALID_ARG, "gs_render_backward: null gradient buffer");
    int rc;
    {
        GsStageTimer t(c, GS_STAGE_BLEND_BWD);
        rc = c->fast16 ? launch_blend_backward_v2(c, N, cot_color, cot_depth, cot_alpha, c->fwd.outColor,
                                                  c->fwd.outDepth, c->fwd.outAlpha)
                       : launch_blend_backward(c, N, cot_color, cot_depth, cot_alpha, c->fwd.outAlpha, c->lastContrib);
    }
    if (rc) return rc;
    GsStageTimer t(c, GS_STAGE_PROJ_BWD);
    return launch_projection_fused_backward(c, N, K, c->fwd.xyz, c->fwd.fdc, c->fwd.frest, c->fwd.scales, c->fwd.rot,
                                            c->fwd.opacity, c->fwd.cam, grad_xyz, grad_features_dc, grad_features_rest,
                                            grad_scales, grad_rotation, grad_opacity);
}

int gs_render_backward_adam(gs_ctx* c, const float* cot_color, const float* cot_depth, const float* cot_alpha,
                            float* params_base, float* m_base, float* v_base, long long n_arena, const float lr[6],
                            float beta1, float beta2, float eps, float grad_scale)
{
    if (!c) return GS_ERR_INVALID_ARG;
    { const int prc = backward_preflight(c, "gs_render_backward_adam", cot_depth != nullptr); if (prc) return prc; }
    const int N = c->fwd.N, K = c->fwd.K;
    if (!cot_color || !lr || n_arena < 0 || (N > 0 && (!params_base || !m_base || !v_base)))
        return fail(c, GS_ERR_INVALID_ARG, "gs_render_backward_adam: null buffer");
    const float* lo = params_base;
    const float* hi = params_base + n_arena;
    auto inside = [&](const float* p, long long n) { return n == 0 || (p >= lo && p + n <= hi); };
    if (!inside(c->fwd.xyz, 3LL * N) || !inside(c->fwd.fdc, 3LL * N) || !inside(c->fwd.frest, 3LL * (K - 1) * N) ||
        !inside(c->fwd.scales, 3LL * N) || !inside(c->fwd.rot, 4LL * N) || !inside(c->fwd.opacity, N))
        return fail(c, GS_ERR_SIZE_MISMATCH, "gs_render_backward_adam: the forward's tensors do not lie in the arena");
    int rc;
    {
        GsStageTimer t(c, GS_STAGE_BLEND_BWD);
        rc = c->fast16 ? launch_blend_backward_v2(c, N, cot_color, cot_depth, cot_alpha, c->fwd.outColor,
                                                  c->fwd.outDepth, c->fwd.outAlpha)
                       : launch_blend_backward(c, N, cot_color, cot_depth, cot_alpha, c->fwd.outAlpha, c->lastContrib);
    }
    if (rc) return rc;
    c->fwd.consumed = true;     // the parameters the forward saw are gone after this call
    GsStageTimer t(c, GS_STAGE_PROJ_BWD);
    return launch_projection_fused_backward_adam(c, N, K, c->fwd.xyz, c->fwd.fdc, c->fwd.frest, c->fwd.scales, c->fwd.rot,
                                                 c->fwd.opacity, c->fwd.cam, params_base, m_base, v_base, lr, beta1, beta2,
                                                 eps, grad_scale);
}

int gs_render_backward_dp_begin(gs_ctx* c, const float* cot_color, const float* cot_depth, const float* cot_alpha,
                                float* color_cot)
{
    if (!c) return GS_ERR_INVALID_ARG;
    { const int prc = backward_preflight(c, "gs_render_backward_dp_begin", cot_depth != nullptr); if (prc) return prc; }
    const int N = c->fwd.N;
    if (!cot_color || (N > 0 && !color_cot)) return fail(c, GS_ERR_INVALID_ARG, "gs_render_backward_dp_begin: null buffer");
    int rc;
    {
        GsStageTimer t(c, GS_STAGE_BLEND_BWD);
        rc = c->fast16 ? launch_blend_backward_v2(c, N, cot_color, cot_depth, cot_alpha, c->fwd.outColor,
                                                  c->fwd.outDepth, c->fwd.outAlpha)
                       : launch_blend_backward(c, N, cot_color, cot_depth, cot_alpha, c->fwd.outAlpha, c->lastContrib);
    }
    if (rc) return rc;
    c->fwd.blendBackwardDone = true;
    GsStageTimer t(c, GS_STAGE_PROJ_BWD);
    return launch_color_cot(c, N, color_cot);
}

int gs_render_backward_dp_finish(gs_ctx* c, float* grad_xyz, float* grad_scales, float* grad_rotation, float* grad_opacity)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!c->fwd.valid || !c->fwd.blendBackwardDone)
        return fail(c, GS_ERR_NO_FORWARD, "gs_render_backward_dp_finish: no gs_render_backward_dp_begin on this context");
    const int N = c->fwd.N, K = c->fwd.K;
    if (N > 0 && (!grad_xyz || !grad_scales || !grad_rotation || !grad_opacity))
        return fail(c, GS_ERR_INVALID_ARG, "gs_render_backward_dp_finish: null gradient buffer");
    c->fwd.blendBackwardDone = false;
    GsStageTimer t(c, GS_STAGE_PROJ_BWD);
    return launch_projection_fused_backward(c, N, K, c->fwd.xyz, c->fwd.fdc, c->fwd.frest, c->fwd.scales, c->fwd.rot,
                                            c->fwd.opacity, c->fwd.cam, grad_xyz, nullptr, nullptr, grad_scales,
                                            grad_rotation, grad_opacity, true);
}

int gs_render_backward_dp_finish_geom(gs_ctx* c, float* grad_xyz, float* grad_scales, float* grad_rotation, float* grad_opacity,
                                      float* xyz_own)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!c->fwd.valid || !c->fwd.blendBackwardDone)
        return fail(c, GS_ERR_NO_FORWARD, "gs_render_backward_dp_finish_geom: no gs_render_backward_dp_begin on this context");
    const int N = c->fwd.N;
    if (N > 0 && (!grad_xyz || !grad_scales || !grad_rotation || !grad_opacity || !xyz_own))
        return fail(c, GS_ERR_INVALID_ARG, "gs_render_backward_dp_finish_geom: null buffer");
    c->fwd.blendBackwardDone = false;
    GsStageTimer t(c, GS_STAGE_PROJ_BWD);
    return launch_projection_geom_backward(c, N, c->fwd.xyz, c->fwd.scales, c->fwd.rot, c->fwd.opacity, c->fwd.cam, grad_xyz,
                                           grad_scales, grad_rotation, grad_opacity, xyz_own);
}

int gs_render_backward_dp_geom(gs_ctx* c, const float* cot_color, const float* cot_depth, const float* cot_alpha, float* color_cot,
                               float* grad_xyz, float* grad_scales, float* grad_rotation, float* grad_opacity, float* xyz_own)
{
    if (!c) return GS_ERR_INVALID_ARG;
    { const int prc = backward_preflight(c, "gs_render_backward_dp_geom", cot_depth != nullptr); if (prc) return prc; }
    const int N = c->fwd.N;
    if (!cot_color || (N > 0 && (!color_cot || !grad_xyz || !grad_scales || !grad_rotation || !grad_opacity || !xyz_own)))
        return fail(c, GS_ERR_INVALID_ARG, "gs_render_backward_dp_geom: null buffer");
    int rc;
    {
        GsStageTimer t(c, GS_STAGE_BLEND_BWD);
        rc = c->fast16 ? launch_blend_backward_v2(c, N, cot_color, cot_depth, cot_alpha, c->fwd.outColor,
                                                  c->fwd.outDepth, c->fwd.outAlpha)
                       : launch_blend_backward(c, N, cot_color, cot_depth, cot_alpha, c->fwd.outAlpha, c->lastContrib);
    }
    if (rc) return rc;
    GsStageTimer t(c, GS_STAGE_PROJ_BWD);
    return launch_projection_geom_backward(c, N, c->fwd.xyz, c->fwd.scales, c->fwd.rot, c->fwd.opacity, c->fwd.cam, grad_xyz,
                                           grad_scales, grad_rotation, grad_opacity, xyz_own, color_cot);
}

int gs_sh_grad_from_views_adam_dir(gs_ctx* c, int N, int K, int R, const float* xyz, const float* color_cot_all,
                                   const float* cam_centers, const float* const* own_xyz, float* features_dc, float* features_rest,
                                   float* params_base, float* m_base, float* v_base, long long n_arena, float lr_dc, float lr_rest,
                                   float beta1, float beta2, float eps, float grad_scale, float* xyz_add)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (N < 0 || K < 1 || R < 1 || R > 16 || !cam_centers || n_arena < 0)
        return fail(c, GS_ERR_INVALID_ARG, "gs_sh_grad_from_views_adam_dir: bad N/K/R");
    if ((c->degree + 1) * (c->degree + 1) > K) return fail(c, GS_ERR_SIZE_MISMATCH, "K smaller than (degree+1)^2");
    if (N > 0 && (!xyz || !color_cot_all || !features_dc || (K > 1 && !features_rest) || !params_base || !m_base || !v_base || !xyz_add))
        return fail(c, GS_ERR_INVALID_ARG, "gs_sh_grad_from_views_adam_dir: null buffer");
    if ((uintptr_t)xyz_add & 15) return fail(c, GS_ERR_INVALID_ARG, "gs_sh_grad_from_views_adam_dir: xyz_add must be 16-byte aligned");
    const float* lo = params_base;
    const float* hi = params_base + n_arena;
    if (N > 0 && (features_dc < lo || features_dc + 3LL * N > hi ||
                  (K > 1 && (features_rest < lo || features_rest + 3LL * (K - 1) * N > hi))))
        return fail(c, GS_ERR_SIZE_MISMATCH, "gs_sh_grad_from_views_adam_dir: the SH tensors do not lie in the arena");
    if (c->ccBlockFloats > 0 && (R != c->ccBlockCount || c->ccBlockFloats < 3LL * N + 1))
        return fail(c, GS_ERR_SIZE_MISMATCH, "gs_sh_grad_from_views_adam_dir: R / N do not match the gs_set_gathered_gate layout");
    GsStageTimer t(c, GS_STAGE_ADAM);
    return launch_sh_views_dir_adam(c, N, K, R, xyz, color_cot_all, cam_centers, own_xyz, features_dc, features_rest, params_base,
                                    m_base, v_base, lr_dc, lr_rest, beta1, beta2, eps, grad_scale, xyz_add);
}

int gs_render_backward_dp(gs_ctx* c, const float* cot_color, const float* cot_depth, const float* cot_alpha,
                          float* grad_xyz, float* grad_scales, float* grad_rotation, float* grad_opacity, float* color_cot)
{
    const int rc = gs_render_backward_dp_begin(c, cot_color, cot_depth, cot_alpha, color_cot);
    if (rc) return rc;
    return gs_render_backward_dp_finish(c, grad_xyz, grad_scales, grad_rotation, grad_opacity);
}

int gs_sh_grad_from_views(gs_ctx* c, int N, int K, int R, const float* xyz, const float* color_cot_all,
                          const float* cam_centers, float* grad_features_dc, float* grad_features_rest)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (N < 0 || K < 1 || R < 1 || R > 16 || !cam_centers) return fail(c, GS_ERR_INVALID_ARG, "gs_sh_grad_from_views: bad N/K/R");
    if ((c->degree + 1) * (c->degree + 1) > K) return fail(c, GS_ERR_SIZE_MISMATCH, "K smaller than (degree+1)^2");
    if (N > 0 && (!xyz || !color_cot_all || !grad_features_dc || (K > 1 && !grad_features_rest)))
        return fail(c, GS_ERR_INVALID_ARG, "gs_sh_grad_from_views: null buffer");
    if (c->ccBlockFloats > 0 && (R != c->ccBlockCount || c->ccBlockFloats < 3LL * N + 1))
        return fail(c, GS_ERR_SIZE_MISMATCH, "gs_sh_grad_from_views: R / N do not match the gs_set_gathered_gate layout");
    GsStageTimer t(c, GS_STAGE_PROJ_BWD);
    return launch_sh_grad_from_views(c, N, K, R, xyz, color_cot_all, cam_centers, grad_features_dc, grad_features_rest);
}

int gs_sh_grad_from_views_adam(gs_ctx* c, int N, int K, int R, const float* xyz, const float* color_cot_all,
                               const float* cam_centers, float* features_dc, float* features_rest, float* params_base,
                               float* m_base, float* v_base, long long n_arena, float lr_dc, float lr_rest, float beta1,
                               float beta2, float eps, float grad_scale)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (N < 0 || K < 1 || R < 1 || R > 16 || !cam_centers || n_arena < 0)
        return fail(c, GS_ERR_INVALID_ARG, "gs_sh_grad_from_views_adam: bad N/K/R");
    if ((c->degree + 1) * (c->degree + 1) > K) return fail(c, GS_ERR_SIZE_MISMATCH, "K smaller than (degree+1)^2");
    if (N > 0 && (!xyz || !color_cot_all || !features_dc || (K > 1 && !features_rest) || !params_base || !m_base || !v_base))
        return fail(c, GS_ERR_INVALID_ARG, "gs_sh_grad_from_views_adam: null buffer");
    const float* lo = params_base;
    const float* hi = params_base + n_arena;
    if (N > 0 && (features_dc < lo || features_dc + 3LL * N > hi ||
                  (K > 1 && (features_rest < lo || features_rest + 3LL * (K - 1) * N > hi))))
        return fail(c, GS_ERR_SIZE_MISMATCH, "gs_sh_grad_from_views_adam: the SH tensors do not lie in the arena");
    if (c->ccBlockFloats > 0 && (R != c->ccBlockCount || c->ccBlockFloats < 3LL * N + 1))
        return fail(c, GS_ERR_SIZE_MISMATCH, "gs_sh_grad_from_views_adam: R / N do not match the gs_set_gathered_gate layout");
    GsStageTimer t(c, GS_STAGE_ADAM);
    return launch_sh_grad_from_views_adam(c, N, K, R, xyz, color_cot_all, cam_centers, features_dc, features_rest,
                                          params_base, m_base, v_base, lr_dc, lr_rest, beta1, beta2, eps, grad_scale);
}

int gs_loss_forward_backward(gs_ctx* c, const float* render, const float* target, const float* render_depth,
                             const float* target_depth, const unsigned char* depth_mask, float lambda_dssim,
                             float lambda_depth, float* loss_out, float* cot_color, float* cot_depth)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!render || !target || !loss_out || !cot_color)
        return fail(c, GS_ERR_INVALID_ARG, "gs_loss_forward_backward: null buffer");
    if (lambda_depth != 0.0f && (!render_depth || !target_depth || !depth_mask || !cot_depth))
        return fail(c, GS_ERR_INVALID_ARG, "gs_loss_forward_backward: depth loss needs depth buffers");
    { const int orc = deferred_overflow(c); if (orc) return orc; }
    GsStageTimer t(c, GS_STAGE_LOSS);
    return launch_loss(c, render, target, render_depth, target_depth, depth_mask, lambda_dssim, lambda_depth, loss_out,
                       cot_color, cot_depth);
}

int gs_loss_target_cache_floats(gs_ctx* c, long long* n)
{
    if (!c || !n) return GS_ERR_INVALID_ARG;
    *n = 2LL * 3LL * c->H * c->W;
    return GS_OK;
}

int gs_set_loss_target_cache(gs_ctx* c, float* cache, int filled)
{
    if (!c) return GS_ERR_INVALID_ARG;
    c->lossTargetCache = cache;
    c->lossTargetCacheFilled = cache != nullptr && filled != 0;
    return GS_OK;
}

int gs_set_block_work_buffer(gs_ctx* c, uint32_t* buf)
{
    if (!c) return GS_ERR_INVALID_ARG;
    c->blockWork = buf ? buf : c->blockWorkOwn;
    c->workHint = buf;
    c->cutStore = nullptr;
    return GS_OK;
}

int gs_view_hint_words(gs_ctx* c, int* n)
{
    if (!c || !n) return GS_ERR_INVALID_ARG;
    *n = c->numPixBlocks + c->T;
    return GS_OK;
}

int gs_set_view_hints(gs_ctx* c, uint32_t* buf, int words)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (buf && words < c->numPixBlocks + c->T) return fail(c, GS_ERR_SIZE_MISMATCH, "gs_set_view_hints: buffer too small (gs_view_hint_words)");
    c->blockWork = buf ? buf : c->blockWorkOwn;
    c->workHint = buf;
    // cuts are per tile and renewed per 16x16 block: kept only where the two coincide
    c->cutStore = (buf && c->fast16 && c->tileW == 16 && c->tileH == 16) ? buf + c->numPixBlocks : nullptr;
    return GS_OK;
}

int gs_clear_depth_cuts(gs_ctx* c, uint32_t* buf, int words)
{
    if (!c || !buf) return GS_ERR_INVALID_ARG;
    if (words < c->numPixBlocks + c->T) return fail(c, GS_ERR_SIZE_MISMATCH, "gs_clear_depth_cuts: buffer too small (gs_view_hint_words)");
    GS_HIP_CHECK(c, hipMemsetAsync(buf + c->numPixBlocks, 0, sizeof(uint32_t) * (size_t)c->T, c->stream));
    return GS_OK;
}

int gs_set_depth_cuts(gs_ctx* c, int enable)
{
    if (!c) return GS_ERR_INVALID_ARG;
    c->allowCuts = enable != 0;
    return GS_OK;
}

int gs_forward_missed(gs_ctx* c, int* missed)
{
    if (!c || !missed) return GS_ERR_INVALID_ARG;
    *missed = 0;
    if (!c->fwd.valid) return fail(c, GS_ERR_NO_FORWARD, "gs_forward_missed: no gs_render_forward on this context");
    { const int orc = deferred_overflow(c); if (orc) return orc; }
    if (!c->fwd.cutsActive) return GS_OK;
    const int rc = settle_cut_forward(c);
    if (rc) return rc;
    *missed = c->fwd.missed ? 1 : 0;
    return deferred_overflow(c);      // the wait was behind the forward: its overflow word, if any, has landed
}

int gs_cut_stats(gs_ctx* c, uint32_t out[2])
{
    if (!c || !out) return GS_ERR_INVALID_ARG;
    out[0] = out[1] = 0;
    if (!c->fwd.valid || !c->fwd.cutsActive) return GS_OK;
    if (!c->fwd.missChecked) return fail(c, GS_ERR_INVALID_ARG, "gs_cut_stats: ask gs_forward_missed first");
    out[0] = c->missHost[1]; out[1] = c->missHost[2];
    return GS_OK;
}

int gs_ctx_set_tuning(gs_ctx* c, int knob, long long value)
{
    if (!c) return GS_ERR_INVALID_ARG;
    switch (knob) {
    case GS_TUNE_FWD_WAVES_PER_SIMD:
        if (value < 1 || value > 16) return fail(c, GS_ERR_INVALID_ARG, "gs_ctx_set_tuning: forward waves per SIMD must be 1..16");
        c->fwdWavesPerSimd = (int)value; return GS_OK;
    case GS_TUNE_BWD_WAVES_PER_CU:
        if (value < 1 || value > 64) return fail(c, GS_ERR_INVALID_ARG, "gs_ctx_set_tuning: backward waves per CU must be 1..64");
        c->bwdWavesPerCu = (int)value; return GS_OK;
    case GS_TUNE_FWD_QUADRANTS:
        c->fwdQuadrants = value != 0; return GS_OK;
    case GS_TUNE_FWD_QUEUES:
        if (value != 1 && value != 2 && value != 4 && value != 8) return fail(c, GS_ERR_INVALID_ARG, "gs_ctx_set_tuning: forward queues must be 1, 2, 4 or 8");
        c->fwdQueues = (int)value; return GS_OK;
    case GS_TUNE_FWD_FOUR_WAVES:
        c->fwdWide = value < 0 ? -1 : (value != 0); return GS_OK;
    case GS_TUNE_OP_FWD_PPL:
    case GS_TUNE_OP_BWD_PPL:
        if (value != 1 && value != 2 && value != 4) return fail(c, GS_ERR_INVALID_ARG, "gs_ctx_set_tuning: pixels per lane must be 1, 2 or 4");
        (knob == GS_TUNE_OP_FWD_PPL ? c->opFwdPpl : c->opBwdPpl) = (int)value; return GS_OK;
    case GS_TUNE_WIDE_TILE_SORT:
        c->wideTileSort = value != 0; return GS_OK;
    case GS_TUNE_DEPTH_GRADIENT:
        c->depthGradient = value != 0; return GS_OK;
    case GS_TUNE_HOST_OVERFLOW_ERRORS:
        c->hostOverflowErrors = value != 0; return GS_OK;
    case GS_TUNE_SPLITTER_DEPTH_SORT:
        c->splitterSort = value < 0 ? 0 : (value > 2 ? 2 : (int)value); c->haveSplitters = false; return GS_OK;
    case GS_TUNE_COLOUR_RIDERS:
        c->colourRiders = (int)value; return GS_OK;
    case GS_TUNE_FWD_FOLD_TEST_SCALE:
        if (value < 1 || value > 1000) return fail(c, GS_ERR_INVALID_ARG, "gs_ctx_set_tuning: fold test scale is in permille, 1..1000");
        c->fwdFoldScale = (float)value / 1000.0f; return GS_OK;
    case GS_TUNE_POISON_CHECKPOINTS:
        c->poisonCheckpoints = value != 0; return GS_OK;
    case GS_TUNE_RENDER_ONLY:
        c->renderOnly = value != 0; return GS_OK;
    case GS_TUNE_FWD_SLOW_SLOT:
        if (value < 1 || value > 16) return fail(c, GS_ERR_INVALID_ARG, "gs_ctx_set_tuning: the first slow wave slot must be 1..16 (16 = none)");
        c->fwdSlowSlot = (int)value; return GS_OK;
    case GS_TUNE_FWD_PAIR:
        if (value < -1 || value > 16) return fail(c, GS_ERR_INVALID_ARG, "gs_ctx_set_tuning: forward pair workgroups per CU must be -1 (by list depth), 0..16");
        c->fwdPair = (int)value; return GS_OK;
    case GS_TUNE_TRIM_RECTS:
        if (value < 0 || value > 2) return fail(c, GS_ERR_INVALID_ARG, "gs_ctx_set_tuning: trim rects is 0 (the reference's squares), 1 (cut by the ellipse's box) or 2 (and into four row groups)");
        c->trimRects = value != 0; c->rowGroups = value == 2; return GS_OK;
    case GS_TUNE_FWD_TRACE_BUFFER:
        c->fwdTrace = reinterpret_cast<unsigned long long*>((uintptr_t)value); return GS_OK;
    default:
        return fail(c, GS_ERR_INVALID_ARG, "gs_ctx_set_tuning: unknown knob");
    }
}

int gs_copy_overflow_flag(gs_ctx* c, uint32_t* out)
{
    if (!c || !out) return GS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(copy_word_kernel, dim3(1), dim3(1), 0, c->stream, c->counters + GS_CNT_OVERFLOW, out);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

int gs_set_update_gate(gs_ctx* c, const uint32_t* gate)
{
    if (!c) return GS_ERR_INVALID_ARG;
    c->adamGate = gate ? gate : c->counters + GS_CNT_OVERFLOW;
    return GS_OK;
}

int gs_set_overflow_rider(gs_ctx* c, float* dst)
{
    if (!c) return GS_ERR_INVALID_ARG;
    c->overflowRider = dst;
    return GS_OK;
}

int gs_set_gathered_gate(gs_ctx* c, long long block_floats, int count, uint32_t* reduced_out)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (block_floats < 0 || (block_floats > 0 && (count < 1 || count > 16)))
        return fail(c, GS_ERR_INVALID_ARG, "gs_set_gathered_gate: bad block size / count (1..16 blocks)");
    c->ccBlockFloats = block_floats;
    c->ccBlockCount = block_floats > 0 ? count : 0;
    c->gatheredGateOut = block_floats > 0 ? reduced_out : nullptr;
    return GS_OK;
}

int gs_set_gate_seen(gs_ctx* c, uint32_t* seen)
{
    if (!c) return GS_ERR_INVALID_ARG;
    c->gateSeen = seen;
    return GS_OK;
}

int gs_set_grad_norm_accum(gs_ctx* c, float* accum)
{
    if (!c) return GS_ERR_INVALID_ARG;
    c->gradNormAccum = accum;
    return GS_OK;
}

int gs_block_count(gs_ctx* c, int* n)
{
    if (!c || !n) return GS_ERR_INVALID_ARG;
    *n = c->numPixBlocks;
    return GS_OK;
}

int gs_copy_block_work(gs_ctx* c, uint32_t* out)
{
    if (!c || !out) return GS_ERR_INVALID_ARG;
    if (!c->fwd.valid) return fail(c, GS_ERR_NO_FORWARD, "gs_copy_block_work: no gs_render_forward on this context");
    if (!c->fast16) return fail(c, GS_ERR_INVALID_ARG, "gs_copy_block_work: only for tile sizes served by the fused kernels");
    GS_HIP_CHECK(c, hipMemcpyAsync(out, c->fwd.blockWork, sizeof(uint32_t) * (size_t)c->numPixBlocks, hipMemcpyDeviceToDevice,
                                   c->stream));
    return GS_OK;
}

int gs_copy_last_contrib(gs_ctx* c, uint32_t* out)
{
    if (!c || !out) return GS_ERR_INVALID_ARG;
    if (!c->fwd.valid) return fail(c, GS_ERR_NO_FORWARD, "gs_copy_last_contrib: no gs_render_forward on this context");
    GS_HIP_CHECK(c, hipMemcpyAsync(out, c->lastContrib, sizeof(uint32_t) * (size_t)c->W * c->H, hipMemcpyDeviceToDevice,
                                   c->stream));
    return GS_OK;
}

int gs_adam_step(gs_ctx* c, long long n, float* params, const float* grads, float* m, float* v, int nseg,
                 const long long* seg_end, const float* seg_lr, float beta1, float beta2, float eps, float grad_scale)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (n < 0 || nseg < 1 || nseg > 8 || !seg_end || !seg_lr) return fail(c, GS_ERR_INVALID_ARG, "gs_adam_step: bad segments");
    if (n > 0 && (!params || !grads || !m || !v)) return fail(c, GS_ERR_INVALID_ARG, "gs_adam_step: null buffer");
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15)
        return fail(c, GS_ERR_INVALID_ARG, "gs_adam_step: arenas must be 16-byte aligned");
    long long prev = 0;
    for (int i = 0; i < nseg; i++) {
        if (seg_end[i] < prev || seg_end[i] > n) return fail(c, GS_ERR_INVALID_ARG, "gs_adam_step: segments not ascending");
        prev = seg_end[i];
    }
    if (prev != n) return fail(c, GS_ERR_SIZE_MISMATCH, "gs_adam_step: segments do not cover the arena");
    { const int orc = deferred_overflow(c); if (orc) return orc; }
    return launch_adam(c, n, params, grads, m, v, nseg, seg_end, seg_lr, beta1, beta2, eps, grad_scale);
}

int gs_adam_step_add(gs_ctx* c, long long n, float* params, const float* grads, float* m, float* v, int nseg,
                     const long long* seg_end, const float* seg_lr, float beta1, float beta2, float eps, float grad_scale,
                     const float* add, long long add_n)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (add_n < 0 || add_n > n || (add_n > 0 && !add) || ((uintptr_t)add & 15))
        return fail(c, GS_ERR_INVALID_ARG, "gs_adam_step_add: add must be 16-byte aligned and cover at most the arena");
    if (n < 0 || nseg < 1 || nseg > 8 || !seg_end || !seg_lr) return fail(c, GS_ERR_INVALID_ARG, "gs_adam_step_add: bad segments");
    if (n > 0 && (!params || !grads || !m || !v)) return fail(c, GS_ERR_INVALID_ARG, "gs_adam_step_add: null buffer");
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15)
        return fail(c, GS_ERR_INVALID_ARG, "gs_adam_step_add: arenas must be 16-byte aligned");
    long long prev = 0;
    for (int i = 0; i < nseg; i++) {
        if (seg_end[i] < prev || seg_end[i] > n) return fail(c, GS_ERR_INVALID_ARG, "gs_adam_step_add: segments not ascending");
        prev = seg_end[i];
    }
    if (prev != n) return fail(c, GS_ERR_SIZE_MISMATCH, "gs_adam_step_add: segments do not cover the arena");
    { const int orc = deferred_overflow(c); if (orc) return orc; }
    return launch_adam(c, n, params, grads, m, v, nseg, seg_end, seg_lr, beta1, beta2, eps, grad_scale, add_n > 0 ? add : nullptr, add_n);
}

int gs_accum_grad_norm(gs_ctx* c, int N, const float* xyz_grad, const float* accum_in, float* accum_out)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (N < 0 || (N > 0 && (!xyz_grad || !accum_out))) return fail(c, GS_ERR_INVALID_ARG, "gs_accum_grad_norm: bad arguments");
    return launch_accum_grad_norm(c, N, xyz_grad, accum_in, accum_out);
}

int gs_classify_gaussians(gs_ctx* c, int N, const float* grad_accum, float denom, const float* scales,
                          int scale_stride, const float* opacity, float grad_threshold, float max_scale_thresh,
                          float min_opacity_thresh, int allow_densify, int* actions, int* output_counts)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (N < 0 || scale_stride < 3) return fail(c, GS_ERR_INVALID_ARG, "gs_classify_gaussians: bad N / scale_stride");
    if (N > 0 && (!grad_accum || !scales || !opacity || !actions || !output_counts))
        return fail(c, GS_ERR_INVALID_ARG, "gs_classify_gaussians: null buffer");
    return launch_classify(c, N, grad_accum, denom, scales, scale_stride, opacity, grad_threshold, max_scale_thresh,
                           min_opacity_thresh, allow_densify, actions, output_counts);
}

int gs_densify_offsets(gs_ctx* c, int N, const int* actions, const int* output_counts, int* offsets, long long stats[5])
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (N < 0 || !stats || (N > 0 && (!actions || !output_counts || !offsets)))
        return fail(c, GS_ERR_INVALID_ARG, "gs_densify_offsets: bad arguments");
    return launch_densify_offsets(c, N, actions, output_counts, offsets, stats);
}

int gs_build_densify_output_map(gs_ctx* c, int N, const int* actions, const int* offsets, int total,
                                int* gather_indices, int* noise_mode)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (N < 0 || total < 0 || (N > 0 && (!actions || !offsets)) || (total > 0 && (!gather_indices || !noise_mode)))
        return fail(c, GS_ERR_INVALID_ARG, "gs_build_densify_output_map: bad arguments");
    return launch_build_densify_map(c, N, actions, offsets, total, gather_indices, noise_mode);
}

int gs_densify_gather(gs_ctx* c, int total, int K, const float* xyz, const float* features_dc,
                      const float* features_rest, const float* scales, const float* rotation, const float* opacity,
                      const int* gather_indices, const int* noise_mode, const float* base_noise, float* out_xyz,
                      float* out_features_dc, float* out_features_rest, float* out_scales, float* out_rotation,
                      float* out_opacity)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (total < 0 || K < 1) return fail(c, GS_ERR_INVALID_ARG, "gs_densify_gather: bad total / K");
    if (total > 0 && (!xyz || !features_dc || !scales || !rotation || !opacity || !gather_indices || !noise_mode ||
                      !out_xyz || !out_features_dc || !out_scales || !out_rotation || !out_opacity ||
                      (K > 1 && (!features_rest || !out_features_rest))))
        return fail(c, GS_ERR_INVALID_ARG, "gs_densify_gather: null buffer");
    return launch_densify_gather(c, total, K, xyz, features_dc, features_rest, scales, rotation, opacity,
                                 gather_indices, noise_mode, base_noise, out_xyz, out_features_dc, out_features_rest,
                                 out_scales, out_rotation, out_opacity);
}

int gs_densify_plan(gs_ctx* c, int N, const int* actions, const int* output_counts, int* offsets)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (N < 0 || (N > 0 && (!actions || !output_counts || !offsets)))
        return fail(c, GS_ERR_INVALID_ARG, "gs_densify_plan: bad arguments");
    return launch_densify_plan(c, N, actions, output_counts, offsets);
}

int gs_densify_plan_read(gs_ctx* c, int wait, long long plan[8], int* ready)
{
    if (!c || !plan || !ready) return GS_ERR_INVALID_ARG;
    return densify_plan_read(c, wait, plan, ready);
}

int gs_build_densify_output_map_planned(gs_ctx* c, int N, const int* actions, const int* offsets, int capacity,
                                        int* gather_indices, int* noise_mode)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!c->densifyPlanned) return fail(c, GS_ERR_NO_FORWARD, "gs_build_densify_output_map_planned: no gs_densify_plan on this context");
    if (N < 0 || capacity < 0 || (N > 0 && (!actions || !offsets)) || (capacity > 0 && (!gather_indices || !noise_mode)))
        return fail(c, GS_ERR_INVALID_ARG, "gs_build_densify_output_map_planned: bad arguments");
    return launch_build_densify_map_planned(c, N, actions, offsets, capacity, gather_indices, noise_mode);
}

int gs_densify_gather_planned(gs_ctx* c, int capacity, int K, const float* xyz, const float* features_dc,
                              const float* features_rest, const float* scales, const float* rotation, const float* opacity,
                              const int* gather_indices, const int* noise_mode, unsigned long long noise_seed, float* out_xyz,
                              float* out_features_dc, float* out_features_rest, float* out_scales, float* out_rotation,
                              float* out_opacity)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!c->densifyPlanned) return fail(c, GS_ERR_NO_FORWARD, "gs_densify_gather_planned: no gs_densify_plan on this context");
    if (capacity < 0 || K < 1) return fail(c, GS_ERR_INVALID_ARG, "gs_densify_gather_planned: bad capacity / K");
    if (capacity > 0 && (!xyz || !features_dc || !scales || !rotation || !opacity || !gather_indices || !noise_mode ||
                         !out_xyz || !out_features_dc || !out_scales || !out_rotation || !out_opacity ||
                         (K > 1 && (!features_rest || !out_features_rest))))
        return fail(c, GS_ERR_INVALID_ARG, "gs_densify_gather_planned: null buffer");
    return launch_densify_gather_planned(c, capacity, K, xyz, features_dc, features_rest, scales, rotation, opacity,
                                         gather_indices, noise_mode, noise_seed, out_xyz, out_features_dc, out_features_rest,
                                         out_scales, out_rotation, out_opacity);
}

int gs_densify_gather_planned_packed(gs_ctx* c, int capacity, int K, const float* xyz, const float* features_dc,
                                     const float* features_rest, const float* scales, const float* rotation, const float* opacity,
                                     const int* gather_indices, const int* noise_mode, unsigned long long noise_seed,
                                     float* out_base, const int arena_order[6])
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!c->densifyPlanned) return fail(c, GS_ERR_NO_FORWARD, "gs_densify_gather_planned_packed: no gs_densify_plan on this context");
    if (capacity < 0 || K < 1 || !arena_order) return fail(c, GS_ERR_INVALID_ARG, "gs_densify_gather_planned_packed: bad capacity / K / order");
    int seen = 0;
    for (int i = 0; i < 6; i++) if (arena_order[i] >= 0 && arena_order[i] < 6) seen |= 1 << arena_order[i];
    if (seen != 63) return fail(c, GS_ERR_INVALID_ARG, "gs_densify_gather_planned_packed: arena_order must be a permutation of 0..5");
    if (capacity > 0 && (!xyz || !features_dc || (K > 1 && !features_rest) || !scales || !rotation || !opacity || !gather_indices ||
                         !noise_mode || !out_base))
        return fail(c, GS_ERR_INVALID_ARG, "gs_densify_gather_planned_packed: null buffer");
    if ((uintptr_t)out_base & 15) return fail(c, GS_ERR_INVALID_ARG, "gs_densify_gather_planned_packed: out_base must be 16-byte aligned");
    return launch_densify_gather_planned_packed(c, capacity, K, xyz, features_dc, features_rest, scales, rotation, opacity,
                                                gather_indices, noise_mode, noise_seed, out_base, arena_order);
}

int gs_densify_noise(gs_ctx* c, unsigned long long seed, int rows, float* out)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (rows < 0 || (rows > 0 && !out)) return fail(c, GS_ERR_INVALID_ARG, "gs_densify_noise: bad arguments");
    return launch_densify_noise(c, seed, rows, out);
}

static bool ply_args_ok(int N, int K, const void* a, const void* b, const void* r, const void* o, const void* s,
                        const void* q)
{
    if (N < 0 || K < 1) return false;
    if (N == 0) return true;
    return a && b && o && s && q && (K == 1 || r);
}

int gs_ply_write(gs_ctx* c, const char* path, int N, int K, const float* xyz, const float* features_dc,
                 const float* features_rest, const float* opacity, const float* scales, const float* rotation)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!path || !ply_args_ok(N, K, xyz, features_dc, features_rest, opacity, scales, rotation))
        return fail(c, GS_ERR_INVALID_ARG, "gs_ply_write: bad arguments");
    return ply_write_file(c, path, N, K, xyz, features_dc, features_rest, opacity, scales, rotation);
}

int gs_ply_probe(gs_ctx* c, const char* path, long long* N, int* M, int* D)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!path || !N || !M || !D) return fail(c, GS_ERR_INVALID_ARG, "gs_ply_probe: bad arguments");
    return ply_probe_file(c, path, N, M, D);
}

int gs_ply_load(gs_ctx* c, const char* path, int N, int K, float* xyz, float* features_dc, float* features_rest,
                float* opacity, float* scales, float* rotation)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!path || !ply_args_ok(N, K, xyz, features_dc, features_rest, opacity, scales, rotation))
        return fail(c, GS_ERR_INVALID_ARG, "gs_ply_load: bad arguments");
    return ply_load_file(c, path, N, K, xyz, features_dc, features_rest, opacity, scales, rotation);
}

int gs_ply_pack_rows(gs_ctx* c, int N, int K, const float* xyz, const float* features_dc,
                     const float* features_rest, const float* opacity, const float* scales, const float* rotation,
                     float* rows)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (!ply_args_ok(N, K, xyz, features_dc, features_rest, opacity, scales, rotation) || (N > 0 && !rows))
        return fail(c, GS_ERR_INVALID_ARG, "gs_ply_pack_rows: bad arguments");
    return launch_ply_pack(c, N, K, xyz, features_dc, features_rest, opacity, scales, rotation, rows);
}

int gs_dist_topk(gs_ctx* c, int N, int k, int q_begin, int q_count, const float* xyz, float* out)
{
    if (!c) return GS_ERR_INVALID_ARG;
    if (N < 0 || k < 1 || k > 8 || q_begin < 0 || q_count < 0 || (long long)q_begin + q_count > N)
        return fail(c, GS_ERR_INVALID_ARG, "gs_dist_topk: bad N / k / query range");
    if (q_count > 0 && (!xyz || !out)) return fail(c, GS_ERR_INVALID_ARG, "gs_dist_topk: null buffer");
    return launch_dist_topk(c, N, k, q_begin, q_count, xyz, out);
}

int gs_profile_enable(gs_ctx* c, unsigned stage_mask)
{
    if (!c) return GS_ERR_INVALID_ARG;
    GS_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    c->profMask = stage_mask;
    if (stage_mask) c->profUsed = 0;
    return GS_OK;
}

int gs_profile_read(gs_ctx* c, float ms[GS_STAGE_COUNT], int calls[GS_STAGE_COUNT])
{
    if (!c || !ms || !calls) return GS_ERR_INVALID_ARG;
    GS_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < GS_STAGE_COUNT; i++) { ms[i] = 0.0f; calls[i] = 0; }
    for (size_t i = 0; i < c->profUsed; i++) {
        float t = 0.0f;
        if (hipEventElapsedTime(&t, c->profPool[i].a, c->profPool[i].b) == hipSuccess) {
            ms[c->profPool[i].stage] += t;
            if (!c->profPool[i].extra) calls[c->profPool[i].stage] += 1;
        }
    }
    return GS_OK;
}

int gs_last_stats(gs_ctx* c, uint32_t stats[8])
{
    if (!c || !stats) return GS_ERR_INVALID_ARG;
    int rc;
    if (c->binValid && (rc = launch_tile_counts(c))) return rc;
    if ((rc = read_counters(c))) return rc;
    stats[0] = c->countersHost[GS_CNT_NVIS];
    // pairs binned; under depth cuts that is fewer than the pairs required without them (GS_CNT_MREQ, the capacity check)
    stats[1] = c->countersHost[GS_CNT_OVERFLOW] ? c->countersHost[GS_CNT_MREQ] : c->countersHost[GS_CNT_M];
    stats[2] = c->countersHost[GS_CNT_B];
    stats[3] = c->countersHost[GS_CNT_CONTRIB_LO];
    stats[4] = c->countersHost[GS_CNT_CONTRIB_HI];
    stats[5] = c->countersHost[GS_CNT_OVERFLOW];
    stats[6] = (uint32_t)c->capN;
    stats[7] = (uint32_t)(c->capM > 0xFFFFFFFFLL ? 0xFFFFFFFFLL : c->capM);
    return GS_OK;
}

}  // extern "C"
#pragma GCC visibility pop
