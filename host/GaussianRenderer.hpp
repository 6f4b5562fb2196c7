// GaussianRenderer.hpp -- C++ host-side mirror of the reference's GaussianRenderer
// (GaussianSplattingMlx/Trainer/GaussianRenderer.swift) over the C ABI of include/gsplat.h.
// Header-only; link with -lgsplat_hip.  Same constructor arguments, method names and result tuple as the
// reference class; errors are exceptions instead of fatalError()/precondition (GaussianRenderer.swift:721-733, 789).
#pragma once
#include <stdexcept>
#include <string>

#include "../include/gsplat.h"

namespace gsplat {

struct TILE_SIZE_H_W { int w, h; };

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// device pointers of one render (caller-owned)
struct RenderResult {
    float* render;   // [H,W,3]
    float* depth;    // [H,W,1]
    float* alpha;    // [H,W,1]
    float* radii;    // [N]; visibility_filter = radii > 0 (GaussianRenderer.swift:820)
};

class GaussianRenderer {
public:
    const int active_sh_degree, W, H;
    const TILE_SIZE_H_W TILE_SIZE;
    const bool whiteBackground;

    GaussianRenderer(int active_sh_degree_, int W_, int H_, TILE_SIZE_H_W tile, bool whiteBackground_, int device = 0)
        : active_sh_degree(active_sh_degree_), W(W_), H(H_), TILE_SIZE(tile), whiteBackground(whiteBackground_)
    {
        const int rc = gs_ctx_create(device, W, H, tile.w, tile.h, active_sh_degree, whiteBackground ? 1 : 0, &ctx_);
        if (rc != GS_OK) throw Error(rc, "gs_ctx_create failed (no GPU or bad arguments)");
    }
    ~GaussianRenderer() { gs_ctx_destroy(ctx_); }
    GaussianRenderer(const GaussianRenderer&) = delete;
    GaussianRenderer& operator=(const GaussianRenderer&) = delete;

    gs_ctx* ctx() const { return ctx_; }
    void setStream(void* hipStream) { check(gs_ctx_set_stream(ctx_, hipStream)); }
    void reserve(int maxGaussians, long long maxPairs) { check(gs_ctx_reserve(ctx_, maxGaussians, maxPairs)); }
    // Reports (once) a reserved-capacity overflow of any forward since the last report: include/gsplat.h, "Overflow".
    void sync() { check(gs_sync(ctx_)); }
    // waits for the ctx's stream and nothing else (no report delivered, nothing cleared): in front of overflowPending
    void wait() { check(gs_wait(ctx_)); }
    // the overflow report waiting to be delivered (no wait, not cleared): kind 0 none / 1 pairs / 2 checkpoint arena
    void overflowPending(uint32_t& kind, uint32_t& pairsNeeded)
    {
        uint32_t w[2] = {0, 0};
        check(gs_overflow_pending(ctx_, w));
        kind = w[0]; pairsNeeded = w[1];
    }
    // Launch tuning of this context (gs_tuning); results never depend on it (GS_TUNE_FWD_FOUR_WAVES: not beyond ~1e-5).  GS_TUNE_DEPTH_GRADIENT = 0 is what a
    // trainer without a depth loss sets (the forward then checkpoints four planes instead of five).
    void setTuning(gs_tuning knob, long long value) { check(gs_ctx_set_tuning(ctx_, (int)knob, value)); }
    // Data-parallel hosts: the device word every optimizer kernel tests before touching the parameters (the ranks'
    // overflow words of the step, max-reduced); nullptr = this context's own.
    void copyOverflowFlag(uint32_t* deviceWord) { check(gs_copy_overflow_flag(ctx_, deviceWord)); }
    void setUpdateGate(const uint32_t* deviceWord) { check(gs_set_update_gate(ctx_, deviceWord)); }
    // ABI 5: a data-parallel step's gate riding in the step's first payload, for hosts that issue the collectives themselves
    // (dpStep does this inside): the backward stores the forward's overflow word at dst as 0.0f / 1.0f; the SH rebuild ORs the
    // gathered blocks' words; an optimizer kernel that finds its gate raised sets *seen.
    void setOverflowRider(float* deviceDst) { check(gs_set_overflow_rider(ctx_, deviceDst)); }
    void setGatheredGate(long long blockFloats, int count, uint32_t* reducedOut) { check(gs_set_gathered_gate(ctx_, blockFloats, count, reducedOut)); }
    void setGateSeen(uint32_t* deviceSeen) { check(gs_set_gate_seen(ctx_, deviceSeen)); }

    // Per-view hints (optional; include/gsplat.h): one device u32 buffer of viewHintWords() per training view,
    // zero-filled before its first use.  With it the forward starts its deepest blocks first and bins every tile
    // only as deep as the view's previous forward needed it; after each forward ask forwardMissed() (queue the loss
    // first) and, if it says true, render again with setDepthCuts(false).  clearDepthCuts after densify / prune.
    int viewHintWords() { int n = 0; check(gs_view_hint_words(ctx_, &n)); return n; }
    void setViewHints(uint32_t* buf, int words) { check(gs_set_view_hints(ctx_, buf, words)); }
    void setDepthCuts(bool on) { check(gs_set_depth_cuts(ctx_, on ? 1 : 0)); }
    bool forwardMissed() { int m = 0; check(gs_forward_missed(ctx_, &m)); return m != 0; }
    void clearDepthCuts(uint32_t* buf, int words) { check(gs_clear_depth_cuts(ctx_, buf, words)); }

    // forwardWithCameraParams (GaussianRenderer.swift:823-880), raw parameters in, image out.  out.depth may be null
    // (a training step without a depth term): no depth image is computed, backward() then takes no cotDepth.
    RenderResult forwardWithCameraParams(const gs_camera& cam, int imageWidth, int imageHeight, int N, int K,
                                         const float* xyz, const float* features_dc, const float* features_rest,
                                         const float* opacity, const float* scales, const float* rotations,
                                         RenderResult out)
    {
        if (imageWidth != W || imageHeight != H)
            throw Error(GS_ERR_SIZE_MISMATCH, "Renderer image size mismatch");   // precondition at :789-792
        check(gs_render_forward(ctx_, N, K, xyz, features_dc, features_rest, scales, rotations, opacity, &cam,
                                out.render, out.depth, out.alpha, out.radii));
        return out;
    }

    // VJP of the call above (what MLX.valueAndGrad drives in GaussianTrainer.swift:719-722)
    void backward(const float* cotColor, const float* cotDepth, const float* cotAlpha, float* gXyz, float* gFdc,
                  float* gFrest, float* gScales, float* gRotation, float* gOpacity)
    {
        check(gs_render_backward(ctx_, cotColor, cotDepth, cotAlpha, gXyz, gFdc, gFrest, gScales, gRotation, gOpacity));
    }

    // buildLossAndGrad's loss (GaussianTrainer.swift:689-714)
    void loss(const float* render, const float* target, float lambdaDssim, float* lossOut4, float* cotColor)
    {
        check(gs_loss_forward_backward(ctx_, render, target, nullptr, nullptr, nullptr, lambdaDssim, 0.0f, lossOut4,
                                       cotColor, nullptr));
    }

    // The target's windowed SSIM statistics per training view (include/gsplat.h: gs_set_loss_target_cache): cache = device
    // buffer of lossTargetCacheFloats() floats, one per view; filled = false at a view's first loss, true afterwards.
    long long lossTargetCacheFloats() { long long n = 0; check(gs_loss_target_cache_floats(ctx_, &n)); return n; }
    void setLossTargetCache(float* cache, bool filled) { check(gs_set_loss_target_cache(ctx_, cache, filled ? 1 : 0)); }

    // ---- data-parallel step (include/gsplat.h, "row e"): one process per GPU, one renderer per process --------------
    // The densify event without a drain of the queue (ABI 5): the count stays on the device (plan), the host waits for the plan
    // alone with the map, the gather and the optimizer reset already queued (the sequence: trainer.py, split_and_prune).
    void densifyPlan(int N, const int* actions, const int* outputCounts, int* offsets) { check(gs_densify_plan(ctx_, N, actions, outputCounts, offsets)); }
    // plan[8] = new count, applies, total, keep, split, clone, prune, N; false (wait == false only): not there yet
    bool densifyPlanRead(long long plan[8], bool wait = true)
    {
        int ready = 0;
        check(gs_densify_plan_read(ctx_, wait ? 1 : 0, plan, &ready));
        return ready != 0;
    }
    void buildDensifyOutputMapPlanned(int N, const int* actions, const int* offsets, int capacity, int* gather, int* noiseMode)
    {
        check(gs_build_densify_output_map_planned(ctx_, N, actions, offsets, capacity, gather, noiseMode));
    }
    void densifyGatherPlanned(int capacity, int K, const float* xyz, const float* fdc, const float* frest, const float* scales,
                              const float* rot, const float* opacity, const int* gather, const int* noiseMode,
                              unsigned long long noiseSeed, float* oXyz, float* oFdc, float* oFrest, float* oScales, float* oRot,
                              float* oOpacity)
    {
        check(gs_densify_gather_planned(ctx_, capacity, K, xyz, fdc, frest, scales, rot, opacity, gather, noiseMode, noiseSeed,
                                        oXyz, oFdc, oFrest, oScales, oRot, oOpacity));
    }
    void densifyNoise(unsigned long long seed, int rows, float* out) { check(gs_densify_noise(ctx_, seed, rows, out)); }
    // Rank 0 draws the RCCL id (dpUniqueId) and hands its 128 bytes to every rank by whatever channel the launcher has;
    // dpInit is collective.  After forwardWithCameraParams + loss on this rank's view, dpStep runs backward, gradient
    // exchange (RCCL on the library's own side stream) and Adam with grad_scale = 1 / world; replicas stay identical.
    static void dpUniqueId(unsigned char id[GS_DP_UNIQUE_ID_BYTES])
    {
        const int rc = gs_dp_unique_id(id);
        if (rc != GS_OK) throw Error(rc, "gs_dp_unique_id failed (RCCL not loadable)");
    }
    void dpInit(const unsigned char id[GS_DP_UNIQUE_ID_BYTES], int rank, int world) { check(gs_dp_init(ctx_, id, rank, world)); }
    void dpAttach(void* ncclComm, int rank, int world) { check(gs_dp_attach(ctx_, ncclComm, rank, world)); }
    void dpShutdown() { check(gs_dp_shutdown(ctx_)); }
    void dpStep(gs_dp_mode mode, const gs_dp_step_args& args) { check(gs_dp_step(ctx_, (int)mode, &args)); }
    void dpAllReduceSum(float* deviceBuf, long long n) { check(gs_dp_allreduce_sum(ctx_, deviceBuf, n)); }
    // floats of one rank's block of color_cot_local / color_cot_all (3 N cotangents + the gate word, padded to four)
    static long long dpCcFloats(int N) { return gs_dp_cc_floats(N); }
    // after every committed densify event; throws Error(GS_ERR_REPLICA_MISMATCH) on EVERY rank if the replicas differ
    void dpCheckReplicas(int N, const float* deviceArena, long long nArena) { check(gs_dp_check_replicas(ctx_, N, deviceArena, nArena)); }
    // ABI 6: the same check in two halves (queue it at the event, take the verdict where the host waits anyway), and the ranks'
    // densify plans compared at once on the side stream (gs_densify_plan_read's words; no wait for the ctx stream's queue)
    void dpCheckReplicasBegin(int N, const float* deviceArena, long long nArena) { check(gs_dp_check_replicas_begin(ctx_, N, deviceArena, nArena)); }
    void dpCheckReplicasEnd() { check(gs_dp_check_replicas_end(ctx_)); }
    void dpCheckPlan(const long long* planWords, int n) { check(gs_dp_check_plan(ctx_, planWords, n)); }
    // exchange timing (measurement only): sums in ms over the dpSteps since dpExchangeTiming(true), see gs_dp_exchange_read
    void dpExchangeTiming(bool on) { check(gs_dp_exchange_timing(ctx_, on ? 1 : 0)); }
    int dpExchangeRead(float ms[GS_DP_XT_COUNT], int* rcclVersion = nullptr)
    {
        int steps = 0;
        check(gs_dp_exchange_read(ctx_, ms, &steps, rcclVersion));
        return steps;
    }
    // every rank at the same steps (e.g. every 16th and after a densify event); true = the pair reserve was regrown
    bool dpCheckOverflow(long long* pairsNeeded = nullptr)
    {
        int regrown = 0;
        check(gs_dp_check_overflow(ctx_, &regrown, pairsNeeded));
        return regrown != 0;
    }

private:
    void check(int rc) const
    {
        if (rc != GS_OK) throw Error(rc, gs_last_error(ctx_));
    }
    gs_ctx* ctx_ = nullptr;
};

}  // namespace gsplat
