// GaussianRendererHIP.swift -- Swift binding of include/gsplat.h that keeps the reference's GaussianRenderer API
// (GaussianSplattingMlx/Trainer/GaussianRenderer.swift:703-934) on a ROCm host.
//
// NOT COMPILED in this repository's image (no Swift toolchain): delivered as the stub a maintainer would add.
// It shows, call by call, which MLXFastKernel launch each C entry point replaces.  Device memory is owned by the
// caller as `UnsafeMutablePointer<Float>` (hipMalloc'd); on an MLX-less ROCm host that is the natural currency.
import CGsplat

public struct TILE_SIZE_H_W { public let w: Int; public let h: Int }

public enum GsplatError: Error { case status(Int32, String) }

public final class GaussianRendererHIP {
    public let active_sh_degree: Int
    public let W: Int
    public let H: Int
    public let TILE_SIZE: TILE_SIZE_H_W
    public let whiteBackground: Bool
    private var ctx: OpaquePointer?

    /// GaussianRenderer.init(active_sh_degree:W:H:TILE_SIZE:whiteBackground:) (GaussianRenderer.swift:703-734).
    /// The reference aborts when a kernel is missing; this throws instead.
    public init(active_sh_degree: Int, W: Int, H: Int, TILE_SIZE: TILE_SIZE_H_W, whiteBackground: Bool,
                device: Int32 = 0) throws {
        self.active_sh_degree = active_sh_degree; self.W = W; self.H = H
        self.TILE_SIZE = TILE_SIZE; self.whiteBackground = whiteBackground
        var c: OpaquePointer?
        let rc = gs_ctx_create(device, Int32(W), Int32(H), Int32(TILE_SIZE.w), Int32(TILE_SIZE.h),
                               Int32(active_sh_degree), whiteBackground ? 1 : 0, &c)
        guard rc == 0, let cc = c else { throw GsplatError.status(rc, "gs_ctx_create") }
        ctx = cc
    }
    deinit { if let c = ctx { gs_ctx_destroy(c) } }

    /// Launch tuning of this context (`gs_tuning`); results never depend on it (`GS_TUNE_FWD_FOUR_WAVES`: not beyond ~1e-5).
    public func setTuning(_ knob: gs_tuning, _ value: Int64) throws { try check(gs_ctx_set_tuning(ctx, Int32(knob.rawValue), value)) }
    /// Waits for the stream; throws GS_ERR_WORKSPACE_OVERFLOW once if any forward since the last report needed more
    /// pairs than were reserved (that forward rendered nothing and no optimizer step was taken from it).
    public func sync() throws { try check(gs_sync(ctx)) }

    private func check(_ rc: Int32) throws {
        if rc != 0 { throw GsplatError.status(rc, String(cString: gs_last_error(ctx))) }
    }

    /// forwardWithCameraParams(...) (GaussianRenderer.swift:823-880) on raw parameters: replaces the projection
    /// custom function (:852), buildPackedGaussians (:796), buildGlobalTileSliceInfo (:803) and the tile
    /// composite custom function (:810) with one call.  Outputs render[H,W,3], depth[H,W], alpha[H,W].
    public func forwardWithCameraParams(camera: inout gs_camera, N: Int, K: Int,
                                        xyz: UnsafePointer<Float>, features_dc: UnsafePointer<Float>,
                                        features_rest: UnsafePointer<Float>, scales: UnsafePointer<Float>,
                                        rotation: UnsafePointer<Float>, opacity: UnsafePointer<Float>,
                                        render: UnsafeMutablePointer<Float>, depth: UnsafeMutablePointer<Float>,
                                        alpha: UnsafeMutablePointer<Float>,
                                        radii: UnsafeMutablePointer<Float>?) throws {
        try check(gs_render_forward(ctx, Int32(N), Int32(K), xyz, features_dc, features_rest, scales, rotation,
                                    opacity, &camera, render, depth, alpha, radii))
    }

    /// The VJP MLX.valueAndGrad would have driven (GaussianTrainer.swift:719-722): cotangents of the image in,
    /// gradients of the six raw tensors out.
    public func backward(cotColor: UnsafePointer<Float>, cotDepth: UnsafePointer<Float>?,
                         cotAlpha: UnsafePointer<Float>?, grad_xyz: UnsafeMutablePointer<Float>,
                         grad_features_dc: UnsafeMutablePointer<Float>, grad_features_rest: UnsafeMutablePointer<Float>,
                         grad_scales: UnsafeMutablePointer<Float>, grad_rotation: UnsafeMutablePointer<Float>,
                         grad_opacity: UnsafeMutablePointer<Float>) throws {
        try check(gs_render_backward(ctx, cotColor, cotDepth, cotAlpha, grad_xyz, grad_features_dc,
                                     grad_features_rest, grad_scales, grad_rotation, grad_opacity))
    }

    /// buildLossAndGrad's loss (GaussianTrainer.swift:689-714) with its cotangents; lossOut = device float[4].
    public func loss(render: UnsafePointer<Float>, target: UnsafePointer<Float>, lambda_dssim: Float,
                     lossOut: UnsafeMutablePointer<Float>, cotColor: UnsafeMutablePointer<Float>) throws {
        try check(gs_loss_forward_backward(ctx, render, target, nil, nil, nil, lambda_dssim, 0, lossOut, cotColor, nil))
    }

    // Op-level entry points, one per reference custom function, for a host that swaps kernels one at a time.
    public func projectionScreenFusedForward(N: Int, K: Int, scales: UnsafePointer<Float>, rotations: UnsafePointer<Float>,
                                             means3d: UnsafePointer<Float>, shs: UnsafePointer<Float>,
                                             camera: inout gs_camera, means2d: UnsafeMutablePointer<Float>,
                                             depths: UnsafeMutablePointer<Float>, color: UnsafeMutablePointer<Float>,
                                             cov2d: UnsafeMutablePointer<Float>, conic: UnsafeMutablePointer<Float>,
                                             radii: UnsafeMutablePointer<Float>, rectMin: UnsafeMutablePointer<Float>,
                                             rectMax: UnsafeMutablePointer<Float>) throws {
        // replaces fusedForwardKernel(...) at GaussianRenderer.swift:542-561
        try check(gs_projection_forward(ctx, Int32(N), Int32(K), scales, rotations, means3d, shs, &camera, means2d,
                                        depths, color, cov2d, conic, radii, rectMin, rectMax))
    }

    public func buildGlobalTileSliceInfo(N: Int, rectMin: UnsafePointer<Float>, rectMax: UnsafePointer<Float>,
                                         radii: UnsafePointer<Float>, depths: UnsafePointer<Float>) throws
        -> (totalPairs: UInt32, maxTilePairs: UInt32) {
        // replaces the six kernels + cumsum + two .item() reads of GaussianRenderer.swift:333-490
        try check(gs_tile_bin(ctx, Int32(N), rectMin, rectMax, radii, depths))
        var m: UInt32 = 0, b: UInt32 = 0
        try check(gs_tile_bin_info(ctx, &m, &b))
        return (m, b)
    }

    public func globalTileCompositeForward(N: Int, packed: UnsafePointer<Float>, outColor: UnsafeMutablePointer<Float>,
                                           outDepth: UnsafeMutablePointer<Float>, outAlpha: UnsafeMutablePointer<Float>,
                                           lastContrib: UnsafeMutablePointer<UInt32>) throws {
        // replaces forwardKernel(...) at GaussianRenderer.swift:130-141
        try check(gs_blend_forward(ctx, Int32(N), packed, outColor, outDepth, outAlpha, lastContrib))
    }

    public func globalTileCompositeVJP(N: Int, packed: UnsafePointer<Float>, cotColor: UnsafePointer<Float>,
                                       cotDepth: UnsafePointer<Float>?, cotAlpha: UnsafePointer<Float>?,
                                       outColor: UnsafePointer<Float>, outDepth: UnsafePointer<Float>,
                                       outAlpha: UnsafePointer<Float>, lastContrib: UnsafePointer<UInt32>,
                                       gradPacked: UnsafeMutablePointer<Float>) throws {
        // replaces backwardKernel(...) at GaussianRenderer.swift:208-218
        try check(gs_blend_backward(ctx, Int32(N), packed, cotColor, cotDepth, cotAlpha, outColor, outDepth, outAlpha,
                                    lastContrib, gradPacked))
    }

    // MARK: - rows after the render path (densify / prune, snapshots, init) -- same names as the reference

    /// addGradientAccumulation's kernel (GaussianTrainer.swift:320-338); or fuse it into `backward` with
    /// `gs_set_grad_norm_accum(ctx, accum)`.
    public func accumGradNorm(N: Int, xyzGrad: UnsafePointer<Float>, accumIn: UnsafePointer<Float>?,
                              accumOut: UnsafeMutablePointer<Float>) throws {
        try check(gs_accum_grad_norm(ctx, Int32(N), xyzGrad, accumIn, accumOut))
    }

    /// split_and_prune phases 1-3 (GaussianTrainer.swift:792-856): classify, offsets (one sync: the output count), map.
    public func classifyAndMap(N: Int, gradAccum: UnsafePointer<Float>, denom: Float, scales: UnsafePointer<Float>,
                               opacity: UnsafePointer<Float>, gradThreshold: Float, maxScale: Float, minOpacity: Float,
                               allowDensify: Bool, actions: UnsafeMutablePointer<Int32>, counts: UnsafeMutablePointer<Int32>,
                               offsets: UnsafeMutablePointer<Int32>) throws -> [Int64] {
        try check(gs_classify_gaussians(ctx, Int32(N), gradAccum, denom, scales, 3, opacity, gradThreshold, maxScale,
                                        minOpacity, allowDensify ? 1 : 0, actions, counts))
        var stats = [Int64](repeating: 0, count: 5)      // total, keep, split, clone, prune
        try check(gs_densify_offsets(ctx, Int32(N), actions, counts, offsets, &stats))
        return stats
    }

    /// save_snapshot (GaussianTrainer.swift:909-930) -> PlyWriter.writeGaussianBinary.
    public func writeGaussianBinary(path: String, N: Int, K: Int, xyz: UnsafePointer<Float>, features_dc: UnsafePointer<Float>,
                                    features_rest: UnsafePointer<Float>?, opacity: UnsafePointer<Float>,
                                    scales: UnsafePointer<Float>, rotation: UnsafePointer<Float>) throws {
        try check(gs_ply_write(ctx, path, Int32(N), Int32(K), xyz, features_dc, features_rest, opacity, scales, rotation))
    }

    /// distTopK (GaussianModel.swift:11-31) for the query range [qBegin, qBegin + qCount).
    public func distTopK(N: Int, k: Int, qBegin: Int, qCount: Int, xyz: UnsafePointer<Float>,
                         out: UnsafeMutablePointer<Float>) throws {
        try check(gs_dist_topk(ctx, Int32(N), Int32(k), Int32(qBegin), Int32(qCount), xyz, out))
    }

    /// The target's windowed SSIM statistics per training view (gs_set_loss_target_cache): one device buffer of
    /// lossTargetCacheFloats() floats per view; filled = false at the view's first loss, true afterwards.
    public func lossTargetCacheFloats() throws -> Int { var n: Int64 = 0; try check(gs_loss_target_cache_floats(ctx, &n)); return Int(n) }
    public func setLossTargetCache(_ cache: UnsafeMutablePointer<Float>?, filled: Bool) throws {
        try check(gs_set_loss_target_cache(ctx, cache, filled ? 1 : 0))
    }

    // ---- data-parallel step (include/gsplat.h, "row e") -----------------------------------------------------------
    // The reference trains one view per iteration on one device (GaussianTrainer.swift:486-498); on an 8-GPU node every
    // rank is one process with one renderer, renders its own view, and the library exchanges the gradients over RCCL.

    /// The densify event without a drain of the queue (ABI 5; GaussianTrainer.swift:813-817 reads the count with `.item()`):
    /// the count stays on the device, the host waits for the plan alone with the map, the gather and the optimizer reset queued.
    public func densifyPlan(n: Int, actions: UnsafePointer<Int32>, outputCounts: UnsafePointer<Int32>, offsets: UnsafeMutablePointer<Int32>) throws {
        try check(gs_densify_plan(ctx, Int32(n), actions, outputCounts, offsets))
    }
    /// (new count, applies, total, keep, split, clone, prune, N), or nil while the plan kernel has not run (`wait == false`).
    public func densifyPlanRead(wait: Bool = true) throws -> [Int64]? {
        var plan = [Int64](repeating: 0, count: 8)
        var ready: Int32 = 0
        try check(gs_densify_plan_read(ctx, wait ? 1 : 0, &plan, &ready))
        return ready != 0 ? plan : nil
    }
    /// ABI 6: the planned gather into a PACKED arena (every tensor's segment sized by the new count): what a data-parallel step
    /// wants, its tensor starts computed on the device behind the plan.  `arenaOrder`: tensor ids in arena order (0 xyz,
    /// 1 features_dc, 2 features_rest, 3 scales, 4 rotation, 5 opacity).
    public func densifyGatherPlannedPacked(capacity: Int, k: Int, xyz: UnsafePointer<Float>, featuresDc: UnsafePointer<Float>,
                                           featuresRest: UnsafePointer<Float>?, scales: UnsafePointer<Float>, rotation: UnsafePointer<Float>,
                                           opacity: UnsafePointer<Float>, gather: UnsafePointer<Int32>, noiseMode: UnsafePointer<Int32>,
                                           noiseSeed: UInt64, outBase: UnsafeMutablePointer<Float>, arenaOrder: [Int32]) throws {
        try check(gs_densify_gather_planned_packed(ctx, Int32(capacity), Int32(k), xyz, featuresDc, featuresRest, scales, rotation, opacity,
                                                   gather, noiseMode, noiseSeed, outBase, arenaOrder))
    }
    /// Rank 0 draws the id; hand its 128 bytes to every rank through the launcher's channel.
    public static func dpUniqueId() throws -> [UInt8] {
        var id = [UInt8](repeating: 0, count: Int(GS_DP_UNIQUE_ID_BYTES))
        let rc = gs_dp_unique_id(&id)
        if rc != 0 { throw GsplatError.status(rc, "gs_dp_unique_id failed (RCCL not loadable)") }
        return id
    }
    /// Collective over the ranks that share the id (ncclCommInitRank inside the library).
    public func dpInit(id: [UInt8], rank: Int, world: Int) throws { try check(gs_dp_init(ctx, id, Int32(rank), Int32(world))) }
    public func dpShutdown() throws { try check(gs_dp_shutdown(ctx)) }
    /// Backward + gradient exchange + Adam (grad_scale = 1 / world) of this rank's step; replaces the six
    /// optimizer.applySingle calls (GaussianTrainer.swift:1060-1086) of a single-device iteration.
    public func dpStep(mode: gs_dp_mode, args: inout gs_dp_step_args) throws {
        try check(gs_dp_step(ctx, Int32(mode.rawValue), &args))
    }
    public func dpAllReduceSum(_ buf: UnsafeMutablePointer<Float>, count: Int) throws {
        try check(gs_dp_allreduce_sum(ctx, buf, Int64(count)))
    }
    /// Floats of one rank's block of `color_cot_local` / `color_cot_all` (3 N cotangents + the step's gate word, padded to four).
    public static func dpCcFloats(_ n: Int) -> Int { Int(gs_dp_cc_floats(Int32(n))) }
    /// SURVEY 8(e): after every committed densify event (GaussianTrainer.swift:766-908 replicated per rank); throws
    /// `GS_ERR_REPLICA_MISMATCH` on EVERY rank if the replicas' N or arena checksums differ.
    public func dpCheckReplicas(n: Int, arena: UnsafePointer<Float>, count: Int) throws {
        try check(gs_dp_check_replicas(ctx, Int32(n), arena, Int64(count)))
    }
    /// ABI 6: the check in two halves -- queue it at the event, take the verdict where the host waits for the device anyway.
    public func dpCheckReplicasBegin(n: Int, arena: UnsafePointer<Float>, count: Int) throws {
        try check(gs_dp_check_replicas_begin(ctx, Int32(n), arena, Int64(count)))
    }
    public func dpCheckReplicasEnd() throws { try check(gs_dp_check_replicas_end(ctx)) }
    /// ABI 6: the ranks' densify plans (densifyPlanRead's words) compared at once, on the side stream; throws
    /// `GS_ERR_REPLICA_MISMATCH` on EVERY rank unless all planned the same event.
    public func dpCheckPlan(_ words: [Int64]) throws { try check(gs_dp_check_plan(ctx, words, Int32(words.count))) }
    /// Every rank at the same iterations; true = the pair reserve was regrown (some rank's forward had not fitted).
    public func dpCheckOverflow() throws -> Bool {
        var regrown: Int32 = 0
        var need: Int64 = 0
        try check(gs_dp_check_overflow(ctx, &regrown, &need))
        return regrown != 0
    }
    /// Exchange timing (measurement only): sums in ms over the dpSteps since dpExchangeTiming(true) -- gate / gather / reduce
    /// durations on the library's RCCL stream and the time the render stream stood waiting for them (gs_dp_exchange_read).
    public func dpExchangeTiming(_ on: Bool) throws { try check(gs_dp_exchange_timing(ctx, on ? 1 : 0)) }
    public func dpExchangeRead() throws -> (ms: [Float], steps: Int, rcclVersion: Int) {
        var ms = [Float](repeating: 0, count: 8)      // GS_DP_XT_COUNT
        var steps: Int32 = 0
        var version: Int32 = 0
        try check(gs_dp_exchange_read(ctx, &ms, &steps, &version))
        return (ms, Int(steps), Int(version))
    }
    /// Waits for the ctx's stream and nothing else (no report delivered, nothing cleared): in front of `overflowPending`.
    public func wait() throws { try check(gs_wait(ctx)) }
    /// The overflow report waiting to be delivered (no wait, not cleared): kind 0 none / 1 pairs / 2 checkpoint arena.
    public func overflowPending() throws -> (kind: UInt32, pairsNeeded: UInt32) {
        var w: [UInt32] = [0, 0]
        try check(gs_overflow_pending(ctx, &w))
        return (w[0], w[1])
    }
}
