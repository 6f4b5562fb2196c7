/*
 * gsplat.h -- C ABI of the MI355X-native 3D-Gaussian-Splatting render/backward path.
 *
 * Drop-in boundary for the reference's Trainer render/backward path: every entry
 * point replaces one MLX CustomFunction / MLXFastKernel launch site of the
 * reference (file:line cited per function, relative to the reference repo).
 * A Swift (or any FFI) host keeps the GaussianRenderer / GaussianTrainer API and
 * calls these instead of MLXFast.metalKernel; see INTEGRATION.md.
 *
 * Conventions
 *  - All array arguments are DEVICE pointers (hipMalloc'd, or any framework's
 *    device buffer) to dense row-major f32 / u32 / i32 data, unless marked HOST.
 *  - The caller owns every input/output buffer.  The library owns only the
 *    context workspace (sort scratch, tile lists, saved forward state) and never
 *    frees caller memory.
 *  - Every call is asynchronous on the context's HIP stream; only the functions
 *    marked [sync] wait for the device.
 *  - A context is bound to one device + stream, is not thread-safe, and holds
 *    ONE forward in flight (like the reference's saved-state closures,
 *    GaussianRenderer.swift:119-122): a backward follows its forward on the same ctx.
 *  - Return value: 0 = GS_OK, otherwise a gs_status; gs_last_error() gives text.
 *    Nothing aborts, nothing throws across the ABI (the reference fatalError()s,
 *    GaussianRenderer.swift:264, 721-733, 808, 814).
 */
#ifndef GSPLAT_H
#define GSPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSPLAT_ABI_VERSION 6

typedef enum gs_status {
    GS_OK = 0,
    GS_ERR_INVALID_ARG = 1,        /* null pointer, negative size, bad degree ... */
    GS_ERR_SIZE_MISMATCH = 2,      /* image size / N / K differs from what the ctx or the saved forward holds */
    GS_ERR_WORKSPACE_OVERFLOW = 3, /* tile-splat pairs exceeded the reserved capacity (see gs_ctx_reserve, "Overflow") */
    GS_ERR_HIP = 4,                /* a HIP runtime call failed */
    GS_ERR_NO_FORWARD = 5,         /* backward / tile query without a matching forward on this ctx */
    GS_ERR_NO_DEVICE = 6,          /* no usable GPU */
    GS_ERR_IO = 7,                 /* snapshot file missing, unwritable, truncated or malformed */
    GS_ERR_COMM = 8,               /* RCCL could not be loaded, or a communicator / collective call failed (gs_dp_*) */
    GS_ERR_REPLICA_MISMATCH = 9    /* gs_dp_check_replicas: the ranks do not hold the same model (every rank gets this code) */
} gs_status;

typedef struct gs_ctx gs_ctx;

/* HOST struct: one view.  Same numbers the reference feeds its projection kernel
 * (Trainer/CameraUtil.swift:5-102; GaussianRenderer.swift:852-857):
 * view = (c2w^-1)^T row-major so that p_view = [p,1].view; proj row-major, p_clip = p_view.proj. */
typedef struct gs_camera {
    float view[16];
    float proj[16];
    float cam_center[3];
    float fov_x, fov_y;     /* radians */
    float focal_x, focal_y; /* pixels */
} gs_camera;

/* ---- context ---------------------------------------------------------------------------------- */

/* Replaces GaussianRenderer.init(active_sh_degree:W:H:TILE_SIZE:whiteBackground:) (GaussianRenderer.swift:703-734).
 * Any tile size is served; what differs is how the FUSED entry points (gs_render_forward / _backward*) work inside:
 *   tile sizes that are multiples of 16 (16x16 is the fastest): tile lists as the reference builds them, swept by 16x16
 *     pixel blocks;
 *   any other tile size (the reference app's TILE_SIZE = (W/4, H/4), ColmapDataLoader.swift:495-498): BLOCK LISTS.  Every
 *     tile is cut into 16x16 pixel blocks (the last column / row of a tile narrower) and the fused path bins, sorts and
 *     blends per block: a block's list holds, in the reference's order, the Gaussians of its tile's list that can reach
 *     the block (weight exp(-q/2) >= 2^-29 somewhere on it -- the bound below which the fused kernels drop an entry anyway).
 *     Outputs are those of the reference's tile lists within the documented bars (image 1e-4, gradients 1e-3).  One
 *     consequence for a TRAINING run: a Gaussian that reaches no block of a tile at 2^-29 gets gradients
 *     below 1e-15 of the tensor's scale from the reference and exactly zero here, and the reference's Adam (eps = 1e-15, no bias
 *     correction) turns any non-zero gradient into a step of about the learning rate -- the reference random-walks such
 *     elements, this library leaves them where they are (measured, ten steps at 50x38 tiles: 0.4 % of the opacity / SH-rest
 *     elements more than 1e-3 apart, every loss within 2e-6).  What changes besides is what the fused path REPORTS about its lists: gs_last_stats' M and max list count (Gaussian, block) pairs,
 *     gs_copy_last_contrib / gs_copy_block_work / gs_block_count / the view-hint words refer to blocks and positions in block
 *     lists, and the tile queries (gs_tile_bin_info / _views / _export, gs_build_packed_tile_indices, gs_blend_forward /
 *     _backward) answer GS_ERR_NO_FORWARD after a fused forward until gs_tile_bin has run (they always describe the
 *     caller's tile grid).  View hints and depth cuts work as with 16x16 tiles.  The op-level entry points are unaffected.
 *     (Environment GSPLAT_BLOCK_LISTS=0 at context creation keeps the older form -- tile lists, every block scanning its
 *     tile's list with the generic kernels, no hints or cuts -- for A/B runs.) [sync] */
int gs_ctx_create(int device, int W, int H, int tile_w, int tile_h, int sh_degree, int white_bg, gs_ctx** out);
int gs_ctx_destroy(gs_ctx* ctx);
/* Bind to a caller stream (hipStream_t passed as void*).  NULL is the HIP default (null) stream, as in any
 * HIP call.  A new ctx starts on a private non-blocking stream of its own. [sync] */
int gs_ctx_set_stream(gs_ctx* ctx, void* hip_stream);
/* Pre-size the workspace so that no call allocates (and so no call synchronises) later.
 * max_pairs = capacity for M (sum of tiles touched).  0 keeps the current value. [sync]
 *
 * Overflow.  Without a reserve every forward checks M on the host and regrows the workspace (one wait per forward, as
 * the reference's .item() reads, GaussianRenderer.swift:399, 462).  With a reserve nothing waits, so a forward whose
 * pairs exceed max_pairs cannot fail at its own call: it renders the background only and raises a flag in host
 * memory.  From then on
 *   - on the device, every optimizer kernel of this ctx (gs_render_backward_adam, gs_adam_step,
 *     gs_sh_grad_from_views_adam) whose step belongs to that forward leaves parameters and moments untouched
 *     (see gs_set_update_gate), so no step is ever taken from a blank render;
 *   - on the host, the next gs_render_forward / gs_render_backward* / gs_loss_forward_backward / gs_adam_step /
 *     gs_forward_missed that sees the flag returns GS_ERR_WORKSPACE_OVERFLOW (gs_last_error names the M needed) and
 *     keeps returning it until gs_sync has reported it once or gs_ctx_reserve has been called again. */
int gs_ctx_reserve(gs_ctx* ctx, int max_gaussians, long long max_pairs);
/* Bytes of device workspace currently held. */
size_t gs_workspace_bytes(const gs_ctx* ctx);
/* Wait for the stream and report deferred errors (GS_ERR_WORKSPACE_OVERFLOW of any forward since the last report). [sync] */
int gs_sync(gs_ctx* ctx);
/* The overflow report that is waiting to be delivered, read WITHOUT waiting and WITHOUT clearing it (gs_sync delivers and
 * clears): out[0] = 0 none, 1 a forward needed more pairs than reserved, 2 a fused forward ran out of checkpoint slots;
 * out[1] = the pair count that forward needed (kind 1).  For hosts that must size a regrow from the forward that TRIPPED --
 * which need not be the last one (a data-parallel trainer looks only every 16th step) -- rather than from gs_last_stats.
 * Call it behind a wait for the stream; then gs_sync to take delivery.  (Replaces the reference's per-forward `.item()`
 * check of M, GaussianRenderer.swift:399.) */
int gs_overflow_pending(gs_ctx* ctx, uint32_t out[2] /*HOST*/);
/* Waits for everything queued on the ctx's stream -- the stream captured by gs_ctx_set_stream, which need not be the host
 * framework's current one -- and nothing else: reports nothing, clears nothing (gs_sync does both).  The wait to put in
 * front of gs_overflow_pending. [sync] */
int gs_wait(gs_ctx* ctx);
const char* gs_last_error(const gs_ctx* ctx);
int gs_abi_version(void);

/* ---- a3 / a4: projection ------------------------------------------------------------------------ */

/* gaussian_projection_screen_fused_forward (slang/gaussian_projection_kernels.slang:36-173; launch
 * GaussianRenderer.swift:542-561).  Inputs are ACTIVATED scales/rotations/opacity as in the reference.
 * Outputs: means2d[N,2] depths[N] color[N,3] cov2d[N,2,2] conic[N,2,2] radii[N] rectMin[N,2] rectMax[N,2]. */
int gs_projection_forward(gs_ctx* ctx, int N, int K, const float* scales, const float* rotations,
                          const float* means3d, const float* shs, const gs_camera* cam /*HOST*/,
                          float* means2d, float* depths, float* color, float* cov2d, float* conic,
                          float* radii, float* rect_min, float* rect_max);

/* gaussian_projection_screen_fused_backward (slang/gaussian_projection_kernels.slang:205-398; launch
 * GaussianRenderer.swift:654-675).  grad_shs[N,K,3] is fully written (zeros beyond (deg+1)^2).
 * grad_cam_center_point[N,3] is per point; the reference sums it on the host side (:683-684). */
int gs_projection_backward(gs_ctx* ctx, int N, int K, const float* scales, const float* rotations,
                           const float* means3d, const float* shs, const gs_camera* cam /*HOST*/,
                           const float* cot_depths, const float* cot_means2d, const float* cot_cov2d,
                           const float* cot_color, const float* cot_conic, float* grad_scales,
                           float* grad_rotations, float* grad_means3d, float* grad_shs,
                           float* grad_cam_center_point);

/* ---- a6: tile binning ---------------------------------------------------------------------------- */

/* buildGlobalTileSliceInfo (GaussianRenderer.swift:333-490) = count_tiles_per_gaussian, cumsum,
 * generate_keys, radix_sort_tile_keys_fused_forward, compute_tile_ranges,
 * compute_tile_counts_from_ranges (slang/gaussian_tile_global_kernels.slang:17-367).
 * Result (order: tile, depth bits, Gaussian index) stays in the ctx for gs_blend_*.  No host sync. */
int gs_tile_bin(gs_ctx* ctx, int N, const float* rect_min, const float* rect_max, const float* radii,
                const float* depths);
/* gs_tile_bin under per-tile DEPTH CUTS (the op-level form of what gs_set_view_hints does for the fused path): a pair
 * (Gaussian, tile) is binned only if the Gaussian's depth bits (the u32 pattern of its f32 depth, the reference's low
 * key word, slang/gaussian_tile_global_kernels.slang:73-126) do not exceed 0xFFFFFFFF - tile_cuts[tile];
 * tile_cuts[tile] == 0 means "no cut".  tile_cuts: DEVICE u32 [T], caller-owned, read during the call's kernels.
 * Lists are in key order, so every tile's list is a PREFIX of the list gs_tile_bin builds (ties at the cut are kept);
 * M / B / ranges / counts describe the cut lists.  NULL = gs_tile_bin. */
int gs_tile_bin_cut(gs_ctx* ctx, int N, const float* rect_min, const float* rect_max, const float* radii,
                    const float* depths, const uint32_t* tile_cuts);
/* M = total pairs, B = max pairs in any tile (the reference's two .item() reads, :399, :462). [sync] */
int gs_tile_bin_info(gs_ctx* ctx, uint32_t* M /*HOST*/, uint32_t* B /*HOST*/);
/* Device views owned by the ctx, valid until the next gs_tile_bin / gs_render_forward:
 * sorted_gauss_idx[M], tile_ranges[T,2], tile_counts[T]. */
int gs_tile_bin_views(gs_ctx* ctx, const uint32_t** sorted_gauss_idx, const uint32_t** tile_ranges,
                      const uint32_t** tile_counts);
/* Copies the same three arrays into caller buffers (device; any pointer may be NULL); sorted_gauss_idx
 * must hold M entries (gs_tile_bin_info). */
int gs_tile_bin_export(gs_ctx* ctx, uint32_t* sorted_gauss_idx, uint32_t* tile_ranges, uint32_t* tile_counts);
/* build_packed_tile_indices (slang/gaussian_tile_global_kernels.slang:377-404): the reference's dense
 * zero-padded i32 [T,B] table, for hosts that still want it.  out has T*B entries. */
int gs_build_packed_tile_indices(gs_ctx* ctx, uint32_t B, int32_t* out);

/* ---- a5, a7, a8: packing and alpha blending --------------------------------------------------------- */

/* buildPackedGaussians (GaussianRenderer.swift:85-99): [means2d(2), conic(4), color(3), opacity(1), depth(1)]. */
int gs_pack_gaussians(gs_ctx* ctx, int N, const float* means2d, const float* conic, const float* color,
                      const float* opacity, const float* depths, float* packed /*[N,11]*/);

/* gaussian_tile_global_forward (slang/gaussian_tile_global_kernels.slang:523-614; launch
 * GaussianRenderer.swift:130-141) over the ctx's current tile lists.
 * out_color[P,3] out_depth[P] out_alpha[P] last_contrib[P] (u32). */
int gs_blend_forward(gs_ctx* ctx, int N, const float* packed /*[N,11]*/, float* out_color, float* out_depth,
                     float* out_alpha, uint32_t* last_contrib);

/* gaussian_tile_global_backward (slang/gaussian_tile_global_kernels.slang:648-881; launch
 * GaussianRenderer.swift:208-218).  grad_packed[N,11] is fully written (zero where untouched). */
int gs_blend_backward(gs_ctx* ctx, int N, const float* packed, const float* cot_color, const float* cot_depth,
                      const float* cot_alpha, const float* out_color, const float* out_depth,
                      const float* out_alpha, const uint32_t* last_contrib, float* grad_packed);

/* ---- a10: SSIM -------------------------------------------------------------------------------------- */

/* gaussian(windowSize:sigma:) outer product (LossUtil.swift:47-54, GaussianTrainer.swift:308-314);
 * writes K*K floats to HOST memory.  Off-centre on purpose (centre = K/2.0). */
int gs_ssim_window(int K, float sigma, float* window /*HOST [K*K]*/);
/* ssim_forward (slang/ssim_kernels.slang:94-155; launch GaussianTrainer.swift:584-590); images HWC. */
int gs_ssim_forward(gs_ctx* ctx, int H, int W, int C, int K, const float* img1, const float* img2,
                    const float* window /*device [K*K]*/, float* out_ssim, float* out_mu1, float* out_mu2,
                    float* out_sigma1, float* out_sigma2, float* out_sigma12);
/* ssim_backward (slang/ssim_kernels.slang:181-266; launch GaussianTrainer.swift:611-621). */
int gs_ssim_backward(gs_ctx* ctx, int H, int W, int C, int K, const float* grad_out, const float* img1,
                     const float* img2, const float* window, const float* mu1, const float* mu2,
                     const float* sigma1, const float* sigma2, const float* sigma12, float* grad_img1,
                     float* grad_img2);

/* ---- fused convenience path (what the trainer's lossFn does, GaussianTrainer.swift:652-716) ------- */

/* activations (a2, GaussianRenderer.swift:936-963) + projection + packing + binning + blend, from the six
 * RAW parameter tensors.  xyz[N,3] features_dc[N,1,3] features_rest[N,K-1,3] scales[N,3] rotation[N,4]
 * opacity[N].  Outputs render[H,W,3] depth[H,W] alpha[H,W]; radii[N] may be NULL.  out_depth may be NULL
 * too: a training step without a depth term reads no depth image, and the blend then carries no depth sum
 * (its backward accepts no cot_depth: GS_ERR_INVALID_ARG).  Saves the state gs_render_backward needs inside
 * the ctx.  No host sync when capacity was reserved.
 * Two deliberate deviations from the reference's arithmetic, both where the reference's own is 0 x inf: a
 * Gaussian that no pixel blended gets the exact zero gradient from gs_render_backward* (not J^T 0 evaluated term by term), and a
 * Gaussian whose 2-D covariance has no positive determinant in float32 (cancellation on a needle tens of thousands of pixels
 * long: a conic that is not positive definite) is invisible while it is so -- radius 0, no pairs, zero gradient.  The op-level
 * gs_projection_forward / _backward keep the 1:1 arithmetic. */
int gs_render_forward(gs_ctx* ctx, int N, int K, const float* xyz, const float* features_dc,
                      const float* features_rest, const float* scales, const float* rotation,
                      const float* opacity, const gs_camera* cam /*HOST*/, float* out_color, float* out_depth,
                      float* out_alpha, float* radii);

/* VJP of gs_render_forward w.r.t. the six raw tensors (blend backward, a5 split, projection backward,
 * activation VJPs).  cot_depth / cot_alpha may be NULL (= zeros, the default training case, a11).
 * The forward's inputs and outputs must still be alive and unchanged. */
int gs_render_backward(gs_ctx* ctx, const float* cot_color, const float* cot_depth, const float* cot_alpha,
                       float* grad_xyz, float* grad_features_dc, float* grad_features_rest, float* grad_scales,
                       float* grad_rotation, float* grad_opacity);

/* gs_render_backward + gs_adam_step in one: the projection backward applies every element's Adam update in place
 * instead of writing a gradient arena for gs_adam_step to read back (same arithmetic; ~1/3 fewer bytes over the two
 * kernels).  The six tensors handed to the preceding gs_render_forward must lie inside [params_base, params_base +
 * n_arena); m_base / v_base are the moment arenas with the same layout.  lr HOST [6] in the reference's parameter
 * order xyz, f_dc, f_rest, scales, rotation, opacity (GaussianModel.swift:56-65).  Single-device steps only (with
 * several ranks the gradients have to be exchanged first).  Consumes the forward: a second backward needs a new
 * gs_render_forward. */
int gs_render_backward_adam(gs_ctx* ctx, const float* cot_color, const float* cot_depth, const float* cot_alpha,
                            float* params_base, float* m_base, float* v_base, long long n_arena, const float lr[6],
                            float beta1, float beta2, float eps, float grad_scale);

/* Data-parallel form of gs_render_backward (not in the reference, which is single-device): identical, except that
 * instead of the two SH gradient tensors it returns color_cot[N,3] = the cotangent of the SH colour after the
 * max(., 0) gate.  One view's SH gradient is basis_k(xyz - cam_center) x color_cot, so ranks exchange 12 B per
 * Gaussian (all-gather) instead of 12 K B (all-reduce) and rebuild the sum with gs_sh_grad_from_views. */
int gs_render_backward_dp(gs_ctx* ctx, const float* cot_color, const float* cot_depth, const float* cot_alpha,
                          float* grad_xyz, float* grad_scales, float* grad_rotation, float* grad_opacity,
                          float* color_cot /*[N,3]*/);
/* The same in two halves, so that the exchange of color_cot can overlap with the projection backward: _begin runs
 * the blend backward and produces color_cot (from the blend's accumulator and the gate bits the forward kept);
 * _finish runs the projection backward for the four geometry gradients.  _finish must follow _begin on the ctx. */
int gs_render_backward_dp_begin(gs_ctx* ctx, const float* cot_color, const float* cot_depth, const float* cot_alpha,
                                float* color_cot /*[N,3]*/);
int gs_render_backward_dp_finish(gs_ctx* ctx, float* grad_xyz, float* grad_scales, float* grad_rotation,
                                 float* grad_opacity);
/* grad_features_dc[N,1,3] / grad_features_rest[N,K-1,3] = sum over the R views (R <= 16) of
 * basis_k(xyz - cam_centers[r]) * color_cot_all[r][N][3]. */
int gs_sh_grad_from_views(gs_ctx* ctx, int N, int K, int R, const float* xyz, const float* color_cot_all,
                          const float* cam_centers /*HOST [R,3]*/, float* grad_features_dc, float* grad_features_rest);

/* gs_sh_grad_from_views with the Adam step of the two SH tensors fused in (features_dc / features_rest are the
 * PARAMETER tensors inside [params_base, params_base + n_arena), updated in place; no SH gradient is written).  xyz
 * must still hold the positions the gradients were taken at: call it before the geometry slice's gs_adam_step. */
int gs_sh_grad_from_views_adam(gs_ctx* ctx, int N, int K, int R, const float* xyz, const float* color_cot_all,
                               const float* cam_centers /*HOST [R,3]*/, float* features_dc, float* features_rest,
                               float* params_base, float* m_base, float* v_base, long long n_arena, float lr_dc,
                               float lr_rest, float beta1, float beta2, float eps, float grad_scale);

/* ABI 6 -- the data-parallel step with the SH rows read ONCE.  A view's xyz gradient has a part that comes through the colour:
 * d_r = sum_k grad basis_k(xyz - cam_r) (SH_k . color_cot_r), for which gs_render_backward_dp_finish stages all SH rows a
 * second time.  Like the SH gradient it is a function of replicated values and of the view's gathered colour cotangent,
 * so the kernel that holds the SH rows for their Adam step rebuilds sum_r d_r for all views:
 *   gs_render_backward_dp_finish_geom   = _finish without the SH rows: grad_xyz LACKS d_r (the all-reduce sums the rest),
 *                                         xyz_own[N,3] receives a copy of it (this view's, for the densify statistic);
 *   gs_sh_grad_from_views_adam_dir      = gs_sh_grad_from_views_adam + xyz_add[N,3] = sum_r d_r (16-byte aligned, readable
 *                                         up to 3 N rounded up to four floats, the pad zero) + the densify statistic:
 *                                         own_xyz[r] (HOST array of R DEVICE pointers, NULL for the views of other ranks) is
 *                                         the xyz_own of view r, and |own_xyz[r] + d_r| is added to the accumulator set by
 *                                         gs_set_grad_norm_accum (per view, GaussianTrainer.swift:724-742) -- _finish_geom
 *                                         adds nothing there;
 *   gs_adam_step_add                    = gs_adam_step with gradient g[i] + add[i] on the leading add_n elements (the xyz
 *                                         segment leads the arena): the geometry slice after the all-reduce.
 * sum_r (geometry_r + d_r) becomes (sum_r geometry_r) + (sum_r d_r): the same value up to float association; every rank
 * forms both sums in the same order, so replicas stay bit-identical.  gs_dp_step does this itself. */
int gs_render_backward_dp_finish_geom(gs_ctx* ctx, float* grad_xyz, float* grad_scales, float* grad_rotation,
                                      float* grad_opacity, float* xyz_own /*[N,3]*/);
/* gs_render_backward_dp_begin + _finish_geom with ONE kernel behind the blend backward (the colour cotangents and the rank's
 * gate word ride in the geometry kernel, which is ~10 us without the SH rows -- too short for an all-gather to hide under,
 * so the separate launch and the fork in front of it bought nothing): what gs_dp_step runs. */
int gs_render_backward_dp_geom(gs_ctx* ctx, const float* cot_color, const float* cot_depth, const float* cot_alpha,
                               float* color_cot /*[N,3]*/, float* grad_xyz, float* grad_scales, float* grad_rotation,
                               float* grad_opacity, float* xyz_own /*[N,3]*/);
int gs_sh_grad_from_views_adam_dir(gs_ctx* ctx, int N, int K, int R, const float* xyz, const float* color_cot_all,
                                   const float* cam_centers /*HOST [R,3]*/, const float* const* own_xyz /*HOST [R] or NULL*/,
                                   float* features_dc, float* features_rest, float* params_base, float* m_base, float* v_base,
                                   long long n_arena, float lr_dc, float lr_rest, float beta1, float beta2, float eps,
                                   float grad_scale, float* xyz_add /*[N,3]*/);
int gs_adam_step_add(gs_ctx* ctx, long long n, float* params, const float* grads, float* m, float* v, int nseg,
                     const long long* seg_end /*HOST*/, const float* seg_lr /*HOST*/, float beta1, float beta2, float eps,
                     float grad_scale, const float* add, long long add_n);

/* ---- row e: the data-parallel step (views shard one per rank; RCCL collectives over xGMI issued by the library) ----
 * The reference trains batch-1 on one device (GaussianTrainer.swift:486-498, 958-1086): there is no call site to
 * replace, the contract is BASELINE.json's north-star.  One process per GPU, one ctx per process.  Every rank runs
 * gs_render_forward (+ gs_loss_forward_backward) on its own view -- no data-path collective -- and then gs_dp_step,
 * which runs the backward, exchanges the gradients on a side stream of the library's own and applies Adam with
 * grad_scale = 1 / world: the same update on every rank, so replicas stay identical without a broadcast.  RCCL is
 * dlopen'ed at the first gs_dp_* call (a process that already holds a copy -- PyTorch's -- shares it); a host needs no
 * RCCL binding of its own. */
#define GS_DP_UNIQUE_ID_BYTES 128
/* ncclGetUniqueId: rank 0 calls it and hands the 128 bytes to every rank by any means (file, socket, launcher). */
int gs_dp_unique_id(void* id /*HOST [GS_DP_UNIQUE_ID_BYTES]*/);
/* ncclCommInitRank on the ctx's device: collective over the `world` ranks (<= 16) that share the id. [sync] */
int gs_dp_init(gs_ctx* ctx, const void* id /*HOST*/, int rank, int world);
/* ... or borrow the host's communicator (an ncclComm_t created on the ctx's device); it is never destroyed here. */
int gs_dp_attach(gs_ctx* ctx, void* nccl_comm, int rank, int world);
/* Drops the communicator (destroys it if gs_dp_init made it); gs_ctx_destroy does the same. [sync] */
int gs_dp_shutdown(gs_ctx* ctx);
int gs_dp_info(gs_ctx* ctx, int* rank /*HOST*/, int* world /*HOST; 0 = no communicator*/);

typedef enum gs_dp_mode {
    GS_DP_ALLREDUCE = 0,     /* one all-reduce (sum) of the whole gradient arena: N (11 + 3K) floats */
    GS_DP_SH_COMPRESSED = 1  /* a view's SH gradient is rank-1 per Gaussian (basis_k(xyz - cam) x colour cotangent): ranks
                              * all-gather the colour cotangents (12 B per Gaussian and rank) under the projection
                              * backward, all-reduce only the geometry slice (44 B per Gaussian) under the SH rebuild, and
                              * every rank rebuilds the summed SH gradient itself, fused with its Adam step */
} gs_dp_mode;
/* HOST struct.  The four arenas (parameters, gradients, Adam moments) share ONE layout of n_arena floats; the six
 * tensors handed to the preceding gs_render_forward must lie inside params_base (their gradients are written at the
 * same offsets of grads_base).  seg_end / seg_lr: the arena's Adam segments as for gs_adam_step (nseg <= 8).
 * GS_DP_SH_COMPRESSED only: geom_numel = length of the LEADING slice of the arena that holds xyz, scales, rotation and
 * opacity (it must end a segment; the SH tensors lie behind it); cam_centers = the camera centres of ALL ranks' views
 * of this step in rank order; color_cot_local [gs_dp_cc_floats(N)] / color_cot_all [world][gs_dp_cc_floats(N)] are
 * caller-owned scratch (a rank's block = its [N,3] cotangents, then its word of the step's gate, padded to four floats).
 * GS_DP_ALLREDUCE: grads_base must have room for n_arena + 1 floats (the gate word rides behind the gradients). */
typedef struct gs_dp_step_args {
    const float *cot_color, *cot_depth, *cot_alpha;   /* DEVICE; depth / alpha may be NULL */
    float *params_base, *grads_base, *m_base, *v_base; /* DEVICE, 16-byte aligned */
    long long n_arena, geom_numel;
    int nseg;
    long long seg_end[8];
    float seg_lr[8];
    float beta1, beta2, eps;
    const float* cam_centers;                          /* HOST [world,3] */
    float *color_cot_local, *color_cot_all;            /* DEVICE */
} gs_dp_step_args;
/* Backward + gradient exchange + Adam of one data-parallel step, after gs_render_forward (and the loss) on this ctx.
 * Collective: every rank of the communicator calls it once per step with the same mode.  Every optimizer kernel of the
 * step is gated on the OR over the ranks of the forwards' overflow words, so a rank whose forward did not fit its pair
 * reserve is never the only one to skip the update.  ABI 5: the word rides in the step's FIRST payload -- behind the
 * colour cotangents of the all-gather (GS_DP_SH_COMPRESSED: the SH rebuild ORs the gathered words) or behind the
 * gradient arena of the all-reduce (GS_DP_ALLREDUCE: summed; any value > 0 gates) -- instead of in a 4-byte all-reduce
 * of its own: two collectives per step (one) instead of three (two).  The deferred host-side overflow error ("Overflow"
 * above) is suppressed inside the call -- no rank leaves a step half-way -- and surfaces through gs_dp_check_overflow.
 * Asynchronous. */
int gs_dp_step(gs_ctx* ctx, int mode, const gs_dp_step_args* args /*HOST*/);
/* Floats of one rank's block of the colour-cotangent all-gather: 3 N cotangents + 1 gate word, padded to a multiple of 4. */
long long gs_dp_cc_floats(int N);
/* In-place all-reduce (sum) of a caller buffer, ordered behind the ctx stream's work and joined back into it (e.g.
 * the densification statistic before gs_classify_gaussians, one per event).  Collective. */
int gs_dp_allreduce_sum(gs_ctx* ctx, float* buf /*DEVICE*/, long long n);
/* Has any step since the last call been gated?  If so the ranks agree on the largest pair count any of them needed
 * (a max all-reduce) and each regrows its reserve to 1.5x that.  Every rank calls it at the same steps (e.g. every
 * 16th, and after a densify event); the decision is built from reduced words, so all ranks take the same branch.
 * *regrown = 1 if the reserve changed, *pairs_needed = the agreed count (0: nothing was gated). [sync] */
int gs_dp_check_overflow(gs_ctx* ctx, int* regrown /*HOST*/, long long* pairs_needed /*HOST*/);

/* "verify with an all-reduce'd checksum every densify step".  After a densify / prune event every rank must
 * hold the same model (same classify inputs, same noise seed: GaussianTrainer.swift:766-908 replicated without
 * communication); a rank that diverged would hang the job in the next size-dependent collective.  Collective, fixed
 * size: (N, sum of the arena in f64, sum of |arena| in f64) of every rank are min- and max-reduced; GS_OK if all three
 * agree, else GS_ERR_REPLICA_MISMATCH on EVERY rank (gs_last_error names what differs and this rank's values).
 * arena: DEVICE, n_arena floats (the parameter arena).  The sums are taken in a fixed order: identical replicas give
 * identical bits. [sync] */
int gs_dp_check_replicas(gs_ctx* ctx, int N, const float* arena, long long n_arena);
/* ABI 6: the same check in two halves, for a host that must not drain its queue at a densify event (the planned event in a
 * data-parallel step).  _begin queues the checksum on the ctx stream and the collective behind it on the side stream
 * (asynchronous; a verdict still outstanding from an earlier _begin is taken first); _end waits for the reduced words
 * alone and returns the verdict (GS_OK when none is outstanding).  Call _end where the host waits for the device anyway
 * (next to gs_dp_check_overflow, at the next event, before shutdown): a parted model is then reported at most that much
 * later -- the hang this check exists to prevent, a diverged N, is caught at once by gs_dp_check_plan. */
int gs_dp_check_replicas_begin(gs_ctx* ctx, int N, const float* arena, long long n_arena);
int gs_dp_check_replicas_end(gs_ctx* ctx);                                             /* [sync on the check's words] */
/* ABI 6: the ranks' densify plans compared.  words: n <= 8 HOST values (gs_densify_plan_read's: new N, applies, total,
 * keep, split, clone, prune, N); one fixed-size max all-reduce of (w, -w) on the side stream, which does not wait for the
 * ctx stream's queue (the event's gather keeps running).  GS_ERR_REPLICA_MISMATCH on EVERY rank unless all agree.
 * Collective: every rank calls it after the same gs_densify_plan_read. [sync on the side stream] */
int gs_dp_check_plan(gs_ctx* ctx, const long long* words /*HOST*/, int n);

/* Exchange timing (measurement only; bench.py's `exchange` block): while enabled, every gs_dp_step records HIP events
 * around its collectives on the library's side stream and around the ctx stream's waits for them (up to 512 steps).
 * gs_dp_exchange_read waits for both streams and returns the SUMS in milliseconds over the `steps` steps timed since the
 * enable: ms[GS_DP_XT_GATE / _GATHER / _REDUCE] = duration of the 4-byte gate all-reduce, the colour-cotangent all-gather
 * and the gradient all-reduce on the side stream (they include the wait for the slowest peer; ABI 5: there is no gate
 * collective any more, ms[GS_DP_XT_GATE] stays 0);
 * ms[GS_DP_XT_EXPOSED_GATHER / _EXPOSED_REDUCE] = time the ctx stream stood in its wait for them, i.e. wire time NOT hidden
 * under compute.  rccl_version: ncclGetVersion's code (0 if the loaded library has none).  No reference call site (the
 * reference has no multi-device step; GaussianTrainer.swift:486-498).  [sync] */
enum { GS_DP_XT_GATE = 0, GS_DP_XT_GATHER = 1, GS_DP_XT_REDUCE = 2, GS_DP_XT_EXPOSED_GATHER = 3, GS_DP_XT_EXPOSED_REDUCE = 4,
       GS_DP_XT_COUNT = 8 };
int gs_dp_exchange_timing(gs_ctx* ctx, int enable);
int gs_dp_exchange_read(gs_ctx* ctx, float ms[GS_DP_XT_COUNT] /*HOST*/, int* steps /*HOST*/, int* rccl_version /*HOST*/);

/* buildLossAndGrad's loss (GaussianTrainer.swift:689-714): L = (1-l)*mean|R-G| + l*(1-mean ssim)
 * + ld*sum(|D-Dgt|*mask)/max(sum mask,1e-6), with its cotangents w.r.t. render colour and depth.
 * loss_out: device float[4] = {total, l1, mean ssim, depth loss}.  target_depth/depth_mask (u8)/
 * render_depth/cot_depth may be NULL when lambda_depth == 0. */
int gs_loss_forward_backward(gs_ctx* ctx, const float* render, const float* target, const float* render_depth,
                             const float* target_depth, const unsigned char* depth_mask, float lambda_dssim,
                             float lambda_depth, float* loss_out, float* cot_color, float* cot_depth);

/* The TARGET's windowed statistics (mean and mean of squares under the 11x11 window, two of SSIM's five) are the same at
 * every visit of a training view; computing them is 40 % of the loss kernel's forward passes.  cache: DEVICE f32
 * [gs_loss_target_cache_floats] (= 2 x 3 x H x W), caller-owned, one per training view (like the view hint buffer).
 * filled = 0: the following gs_loss_forward_backward computes them as always and fills the cache (and then counts it as
 * filled); filled = 1: it reads them -- the target passed must be the image the cache was filled from.  NULL (default):
 * no cache.  The loss and its cotangent are bit-identical in all three cases (the cache holds the very floats the kernel
 * computes). */
int gs_loss_target_cache_floats(gs_ctx* ctx, long long* n /*HOST*/);
int gs_set_loss_target_cache(gs_ctx* ctx, float* cache /*DEVICE or NULL*/, int filled);

/* ---- next row: optimizer step ---------------------------------------------------------------- */

/* Adam over one flat parameter arena, as the trainer applies it per tensor (GaussianTrainer.swift:941-948,
 * 1060-1086; learning rates GaussianModel.swift:56-65): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * p -= lr * m / (sqrt(v) + eps), no bias correction (mlx-swift 0.30.6 Adam default; parity unpinned).
 * The arena is split into nseg contiguous segments ending at seg_end[i] (element index, exclusive), each with
 * its own learning rate.  grad_scale multiplies the gradient first (1/world_size after an all-reduce sum). */
int gs_adam_step(gs_ctx* ctx, long long n, float* params, const float* grads, float* m, float* v, int nseg,
                 const long long* seg_end /*HOST*/, const float* seg_lr /*HOST*/, float beta1, float beta2,
                 float eps, float grad_scale);

/* ---- next row: densify / prune -------------------------------------------------------------
 * The three kernels of GaussianTrainer.swift:317-427 one to one, the scan between them, and the gather + per-slot
 * modification that the reference writes as MLX array ops (:858-893).  The host sequence (when to run, the
 * early-outs, committing the new tensors, resetting the accumulators and the Adam state) is the trainer's:
 * gaussiansplattingmlx_amd/trainer.py mirrors split_and_prune (:766-907).  Parity unpinned (no reference test;
 * MLXRandom noise is an input here). */

/* accum_grad_norm (:320-338): accum_out[i] = accum_in[i] + |xyz_grad[i,:]|.  accum_in NULL = zeros; in place ok. */
int gs_accum_grad_norm(gs_ctx* ctx, int N, const float* xyz_grad, const float* accum_in, float* accum_out);
/* classify_gaussians (:343-393).  denom is the scalar the reference broadcasts to [N] (:796).  scales[N,scale_stride]
 * raw (log) scales, opacity[N] raw.  actions 0 keep / 1 split / 2 clone / 3 prune; output_counts 1 / 2 / 2 / 0. */
int gs_classify_gaussians(gs_ctx* ctx, int N, const float* grad_accum, float denom, const float* scales,
                          int scale_stride, const float* opacity, float grad_threshold, float max_scale_thresh,
                          float min_opacity_thresh, int allow_densify, int* actions, int* output_counts);
/* offsets = cumsum(output_counts) - output_counts (:813-815) plus the action counts (:838-841).  Synchronises (the
 * reference's .item(), :816): stats HOST [5] = total outputs, keep, split, clone, prune. */
int gs_densify_offsets(gs_ctx* ctx, int N, const int* actions, const int* output_counts, int* offsets,
                       long long stats[5]);
/* build_densify_output_map (:398-427): gather_indices[total], noise_mode[total] (0 none, 1 split +, 2 split -,
 * 3 clone copy), zero-initialised first (:852). */
int gs_build_densify_output_map(gs_ctx* ctx, int N, const int* actions, const int* offsets, int total,
                                int* gather_indices, int* noise_mode);
/* Phases 4-5 (:858-893): out_X = X[gather_indices]; split children: scales + Float(-log 1.6), xyz +/- mean(exp(source
 * scales)) * 0.1 * noise; clone copies: xyz + 0.01 * noise.  base_noise[total,3] standard normal, or NULL for the
 * no-split-no-clone branch (:864) that gathers only.  Outputs must not alias inputs. */
int gs_densify_gather(gs_ctx* ctx, int total, int K, const float* xyz, const float* features_dc,
                      const float* features_rest, const float* scales, const float* rotation, const float* opacity,
                      const int* gather_indices, const int* noise_mode, const float* base_noise, float* out_xyz,
                      float* out_features_dc, float* out_features_rest, float* out_scales, float* out_rotation,
                      float* out_opacity);

/* The event WITHOUT a drain of the queue (ABI 5).  gs_densify_offsets hands the count to the host, which then sizes and
 * queues everything behind it on a device that has run dry (the reference's `.item()`, GaussianTrainer.swift:813-817).
 * Planned: the count stays on the device for the kernels that need it, and the host waits for the PLAN alone, with the
 * event's remaining work already queued behind it.
 *   gs_densify_plan            = the scan of gs_densify_offsets, no wait.  Leaves the plan in the ctx: new count (= total
 *                                outputs when the event applies, else N), whether it applies -- the reference's early-outs
 *                                (all pruned :828-832; nothing to split, clone or prune :819-826, :843-847) as a predicate
 *                                --, total, keep, split, clone, prune, N.
 *   ..._output_map_planned     = gs_build_densify_output_map for `capacity` output slots (zero-filled); the identity map
 *                                when the event does not apply.
 *   gs_densify_gather_planned  = gs_densify_gather over a capacity-sized grid: rows [0, new count) are written (the outputs
 *                                must hold `capacity` rows, or at least the new count), the rest is left alone; a row that
 *                                does not apply is a plain copy.  The noise of row j is gs_densify_noise's row j for
 *                                noise_seed: no tensor of `total` rows to size.
 *   gs_densify_plan_read       = plan[8] HOST = new count, applies, total, keep, split, clone, prune, N once the plan
 *                                kernel has run (*ready = 1); wait = 0: asks without waiting.  Waits for the PLAN, not for
 *                                what was queued behind it. [sync on the plan when wait != 0]
 *   gs_densify_noise           = out[rows,3] standard normal, row j a function of (seed, j) alone (Philox4x32-10 +
 *                                Box-Muller): the tensor form of the planned gather's noise, for gs_densify_gather.
 * The host sequence -- laying the new model out at capacity strides so that its pointers do not depend on the count,
 * flipping, resetting the optimizer state, repeating the gather should the new count exceed the capacity -- is the
 * trainer's (gaussiansplattingmlx_amd/trainer.py, split_and_prune). */
int gs_densify_plan(gs_ctx* ctx, int N, const int* actions, const int* output_counts, int* offsets);
int gs_densify_plan_read(gs_ctx* ctx, int wait, long long plan[8] /*HOST*/, int* ready /*HOST*/);
int gs_build_densify_output_map_planned(gs_ctx* ctx, int N, const int* actions, const int* offsets, int capacity,
                                        int* gather_indices, int* noise_mode);
int gs_densify_gather_planned(gs_ctx* ctx, int capacity, int K, const float* xyz, const float* features_dc,
                              const float* features_rest, const float* scales, const float* rotation, const float* opacity,
                              const int* gather_indices, const int* noise_mode, unsigned long long noise_seed, float* out_xyz,
                              float* out_features_dc, float* out_features_rest, float* out_scales, float* out_rotation,
                              float* out_opacity);
/* ABI 6: gs_densify_gather_planned into a PACKED arena -- every tensor's segment sized by the NEW count (padded to four
 * floats), which is what a data-parallel step wants (it all-reduces the arena's leading geometry slice: at capacity strides
 * that slice would carry capacity / N times the bytes).  The tensor starts depend on a count the host does not have yet, so
 * they are computed on the device behind the plan: out_base (16-byte aligned, room for the capacity) + the padded segments
 * of the tensors in front, in arena_order -- tensor ids in the order they lie in the arena (0 xyz, 1 features_dc,
 * 2 features_rest, 3 scales, 4 rotation, 5 opacity; the trainer's arena is 0 3 4 5 1 2: geometry first).  The host lays
 * the same layout out once gs_densify_plan_read has given it the count.  Pads are not written. */
int gs_densify_gather_planned_packed(gs_ctx* ctx, int capacity, int K, const float* xyz, const float* features_dc,
                                     const float* features_rest, const float* scales, const float* rotation,
                                     const float* opacity, const int* gather_indices, const int* noise_mode,
                                     unsigned long long noise_seed, float* out_base, const int arena_order[6] /*HOST*/);
int gs_densify_noise(gs_ctx* ctx, unsigned long long seed, int rows, float* out /*[rows,3]*/);

/* ---- next row: snapshot format ---------------------------------------------------------------
 * Data/PlyWriter.swift: binary little-endian PLY, header comment `features_rest_shape M 3`, vertex = x y z,
 * f_dc_0..2, f_rest_0..3M-1 (coefficient-major, channel-minor: [M][3] flattened, NOT INRIA's channel-major),
 * opacity, scale_0..2, rot_0..3 -- all raw (pre-activation) float32.  Tensors are DEVICE pointers; the file is
 * byte-identical to the reference writer's for the same values. [sync] */

/* PlyWriter.writeGaussianBinary (:22-113, :116-146).  Creates missing parent directories (:106-111). */
int gs_ply_write(gs_ctx* ctx, const char* path, int N, int K, const float* xyz, const float* features_dc,
                 const float* features_rest, const float* opacity, const float* scales, const float* rotation);
/* Header of loadGaussianBinaryPLY (:149-183): vertex count and the features_rest_shape comment. */
int gs_ply_probe(gs_ctx* ctx, const char* path, long long* N, int* M, int* D);
/* loadGaussianBinaryPLY (:149-233) into caller buffers sized from gs_ply_probe (K = M + 1; D must be 3).  Fields
 * are found by name, in whatever order the header lists its `property float` lines (:171-176, :189). */
int gs_ply_load(gs_ctx* ctx, const char* path, int N, int K, float* xyz, float* features_dc, float* features_rest,
                float* opacity, float* scales, float* rotation);
/* The device half of the writer alone: rows[N, 14 + 3(K-1)] in the file's vertex layout (for callers with their own
 * IO, e.g. streaming a snapshot to a viewer). */
int gs_ply_pack_rows(gs_ctx* ctx, int N, int K, const float* xyz, const float* features_dc,
                     const float* features_rest, const float* opacity, const float* scales, const float* rotation,
                     float* rows);

/* ---- next row: point-cloud initialisation -----------------------------------------------------
 * distTopK (Trainer/GaussianModel.swift:11-31): for the query points [q_begin, q_begin + q_count) of xyz[N,3], the
 * mean of the k (1..8) smallest squared distances to all N points, the point itself included; other entries of
 * out[N] are left untouched.  The reference's loop visits only the 256-point chunks starting at multiples of 256
 * below N/256 + 1 (stride bug, :13-18) and leaves the rest 0; the host mirror (model_init.py) reproduces that by
 * choosing the query ranges, or covers every point on request. */
int gs_dist_topk(gs_ctx* ctx, int N, int k, int q_begin, int q_count, const float* xyz, float* out);

/* ---- instrumentation ----------------------------------------------------------------------------------- */

/* Per-stage device time, measured with HIP events recorded on the ctx stream around each stage. */
typedef enum gs_stage {
    GS_STAGE_PROJ_FWD = 0,
    GS_STAGE_BIN = 1,
    GS_STAGE_BLEND_FWD = 2,
    GS_STAGE_LOSS = 3,
    GS_STAGE_BLEND_BWD = 4,
    GS_STAGE_PROJ_BWD = 5,
    GS_STAGE_ADAM = 6,
    GS_STAGE_COUNT = 8
} gs_stage;
/* stage_mask: bit s set = record stage s (clears what was recorded); 0 = stop.  Each recorded stage costs two
 * event packets on the stream (a few microseconds of serialisation each), so time only what is needed. */
int gs_profile_enable(gs_ctx* ctx, unsigned stage_mask);
/* Sum of elapsed ms and number of recorded calls per stage since gs_profile_enable(1). [sync] */
int gs_profile_read(gs_ctx* ctx, float ms[GS_STAGE_COUNT] /*HOST*/, int calls[GS_STAGE_COUNT] /*HOST*/);

/* Last forward's workload statistics, read back from the device. [sync]
 * stats[0]=N_visible stats[1]=M stats[2]=max tile list stats[3]=sum over pixels of nContrib (low 32 bits)
 * stats[4]=high 32 bits of that sum, stats[5]=overflow flag. */
int gs_last_stats(gs_ctx* ctx, uint32_t stats[8] /*HOST*/);
/* Per-view block-work buffer for the following gs_render_forward calls (16x16 tiles and block lists; ignored otherwise).
 * The forward's time is set by its longest serial lists, and where a block's list stops cannot be predicted from
 * its length -- but a previous forward of (nearly) the same view has measured it.  buf: DEVICE u32
 * [gs_block_count], caller-owned, zero-filled before its first use, one per training view, valid until replaced or
 * cleared (NULL = the ctx's own scratch, no hint).  Each forward first reads it as the hint (deepest blocks are
 * started first) and then overwrites it with its own measurement (max nContrib per block), which the matching
 * backward consumes.  Changes the launch order only, never a result. */
int gs_set_block_work_buffer(gs_ctx* ctx, uint32_t* buf);
/* The same buffer with room for per-tile DEPTH CUTS behind the block-work words: buf DEVICE u32
 * [gs_view_hint_words], zero-filled before its first use, one per training view.  With 16x16 tiles the backward
 * then records, per tile, the depth key up to which this view's forward needed the tile's list (sweep length + 25 %
 * + 64 entries), and the next forward of the view bins a (Gaussian, tile) pair only if the Gaussian's key does not
 * exceed it.  Tile lists are in key order, so what is binned is a prefix of the reference's list
 * (GaussianRenderer.swift:333-490 bins everything; most of it is never reached once a tile saturates).
 * Exactness: a tile under a cut whose pixels have not all reached T < 1e-4 at the end of its list marks the forward
 * as MISSED.  The caller must ask gs_forward_missed after each gs_render_forward under cuts and, when it says 1,
 * repeat the forward with gs_set_depth_cuts(ctx, 0) (then re-enable) before using any output: a forward that did not
 * miss is identical to the uncut one, output for output.  Block lists (gs_ctx_create): the same per block.  Tile sizes
 * that are multiples of 16 other than 16x16: work hint only, no cuts. */
int gs_view_hint_words(gs_ctx* ctx, int* n);
int gs_set_view_hints(gs_ctx* ctx, uint32_t* buf, int words);
/* Forgets the cuts kept in a view's hint buffer (its work hint stays): call it for every view after the model was
 * rebuilt (densify / prune) -- a stale cut costs a repeated forward, a missing one only a full binning pass. */
int gs_clear_depth_cuts(gs_ctx* ctx, uint32_t* buf, int words);
/* Depth cuts on (default) / off for the following forwards; without a gs_set_view_hints buffer there are none. */
int gs_set_depth_cuts(gs_ctx* ctx, int enable);
/* *missed = 1 if the last gs_render_forward ran under cuts and has to be repeated without them.  Waits for that
 * forward (not for anything queued behind it) when it ran under cuts; immediate otherwise. [sync on the forward] */
int gs_forward_missed(gs_ctx* ctx, int* missed /*HOST*/);
/* After gs_forward_missed on a forward under cuts: out[0] = pairs binned, out[1] = pairs a full binning would have
 * made (both 0 when the forward ran without cuts).  For the caller's policy: the cuts cost a fixed ~40 us per forward
 * (second expansion pass, the wait above, an occasional repeat) and save ~13 us per million pairs left out. */
int gs_cut_stats(gs_ctx* ctx, uint32_t out[2] /*HOST*/);
/* Densification statistic fused into the backward (GaussianTrainer.swift:724-742, accum_grad_norm): when set,
 * gs_render_backward / _dp_finish add |grad_xyz[i,:]| to accum[i] (DEVICE f32 [N], caller-owned; NULL = off).
 * Same arithmetic as gs_accum_grad_norm, one launch fewer per step. */
int gs_set_grad_norm_accum(gs_ctx* ctx, float* accum);
/* The overflow word of the last forward (1 = its pairs exceeded the reserve), copied to a caller device word on the
 * stream: for hosts that make a decision collective, e.g. data-parallel ranks that all-reduce (max) the words of a
 * step so that every replica skips the same optimizer steps. */
int gs_copy_overflow_flag(gs_ctx* ctx, uint32_t* out /*DEVICE*/);
/* The word the optimizer kernels test before they touch anything (non-zero = skip).  NULL (default) = the ctx's own
 * overflow word of the last forward.  The word must stay valid while set. */
int gs_set_update_gate(gs_ctx* ctx, const uint32_t* gate /*DEVICE*/);
/* ABI 5 -- a data-parallel step's gate WITHOUT a collective of its own (what gs_dp_step does inside; these three are for
 * hosts that issue the collectives themselves, e.g. trainer.py over torch.distributed):
 * gs_set_overflow_rider: the first kernel of the following gs_render_backward / gs_render_backward_dp[_begin] calls also
 *   stores the last forward's overflow word at dst as 0.0f / 1.0f -- e.g. at color_cot + 3 N (the word of this rank's
 *   gather block, gs_dp_cc_floats) or at grads + n_arena (summed by the all-reduce: non-zero bits = gated, so the same
 *   address serves as gs_set_update_gate's word).  NULL = off.
 * gs_set_gathered_gate: the following gs_sh_grad_from_views[_adam] calls read color_cot_all as `count` blocks of
 *   block_floats floats (rank r's cotangents at color_cot_all + r * block_floats), take the OR of the blocks' words at
 *   [3 N] as their gate and store it to reduced_out (DEVICE; typically gs_set_update_gate's word, so that the optimizer
 *   kernels queued behind them test the same).  block_floats = 0: off (blocks of 3 N floats, gs_set_update_gate's word).
 * gs_set_gate_seen: every gs_adam_step / gs_sh_grad_from_views* kernel that finds its gate raised sets *seen = 1 (a host
 *   that looks every 16th step learns that some step of the window was skipped).  NULL = off. */
int gs_set_overflow_rider(gs_ctx* ctx, float* dst /*DEVICE*/);
int gs_set_gathered_gate(gs_ctx* ctx, long long block_floats, int count, uint32_t* reduced_out /*DEVICE*/);
int gs_set_gate_seen(gs_ctx* ctx, uint32_t* seen /*DEVICE*/);

/* Launch tuning, per context (defaults are the measured optima on MI355X; results never depend on these -- GS_TUNE_FWD_FOUR_WAVES alone
 * moves them, by rounding: see there). */
typedef enum gs_tuning {
    GS_TUNE_FWD_WAVES_PER_SIMD = 0, /* persistent waves per SIMD of the fused blend forward (default 4) */
    GS_TUNE_BWD_WAVES_PER_CU = 1,   /* persistent waves per CU of the fused blend backward (default 16) */
    GS_TUNE_FWD_QUADRANTS = 2,      /* retired (accepted, ignored): the fused forward's items are 8x8 quadrants; the 16x8 variant
                                     * of ABI 2 was 25 % slower and is gone */
    GS_TUNE_OP_FWD_PPL = 3,         /* pixels per lane (1, 2, 4) of the op-level gs_blend_forward */
    GS_TUNE_OP_BWD_PPL = 4,         /* ... and gs_blend_backward */
    GS_TUNE_FWD_TRACE_BUFFER = 5,   /* DEVICE u64 [4 * items] (as an integer) receiving per-item start/end clocks, 0 = off */
    GS_TUNE_WIDE_TILE_SORT = 7,     /* 1 (default): the pairs are sorted by tile in one pass when the image has <= 4096 tiles;
                                     * 0: two 8-bit radix passes + range kernel (same lists, bit for bit) */
    GS_TUNE_HOST_OVERFLOW_ERRORS = 8, /* 1 (default): entry points that continue a step return GS_ERR_WORKSPACE_OVERFLOW as soon as
                                     * the host sees the flag of an overflowed forward ("Overflow" above).  0: only gs_sync
                                     * reports it -- for data-parallel hosts, where a rank that bailed out of a step on its own
                                     * would leave the others waiting in a collective: the device gate (gs_set_update_gate on
                                     * the max-reduced flags) skips the step on every rank, and the ranks decide TOGETHER when
                                     * to look (gs_sync), reserve and carry on */
    GS_TUNE_SPLITTER_DEPTH_SORT = 9, /* 1 (default): the depth sort of 16385 .. 655 k Gaussians buckets the records between 127 splitters
                                     * kept from the context's previous depth sort and sorts every bucket locally (three launches);
                                     * 0: four least-significant-digit passes (eight); 2: splitter buckets above 655 k Gaussians too
                                     * (255 splitters, four launches -- measured slower than the LSD passes of those sizes, kept for
                                     * tests and A/B).  Same order, bit for bit, whatever the splitters are -- they only balance
                                     * the buckets */
    GS_TUNE_COLOUR_RIDERS = 10,     /* 1 (default): a K = 25 forward computes its SH colours in workgroups that ride along in the binning
                                     * kernels' launches (they leave most CUs idle) instead of in the projection kernel; 0: one
                                     * projection kernel with the SH loads interleaved (rounds 1-2); 2: split, but no riders (all colours in a
                                     * kernel of their own in front of the blend); 3: one kernel, geometry first, then the wave's own
                                     * colours (what 1 does where no binning kernel can host riders).  Same colours, bit for bit */
    GS_TUNE_FWD_QUEUES = 11,        /* work queues of the fused blend forward: 8 (default) = one per XCD, the four quadrant waves of a pixel
                                     * block on one XCD so that its records are fetched into one L2; 1 = one queue for the chip,
                                     * a block's quadrants on four XCDs (rounds 1-3); 2, 4 in between */
    GS_TUNE_FWD_FOUR_WAVES = 12,    /* fused blend forward with four waves per 8x8 quadrant, each sweeping every fourth 64-entry chunk of the
                                     * list (three of them from T = 1, composed in LDS): -1 (default) = where the image has fewer
                                     * quadrants than the chip has wave slots (<= 512x512 on MI355X; there a quadrant's list is a serial
                                     * chain on a half-empty chip), 1 / 0 = always / never.  Image and gradients within the same bars,
                                     * not the same bits as the one-wave kernel (sums are composed, not accumulated, across chunks) */
    GS_TUNE_FWD_FOLD_TEST_SCALE = 13, /* TEST knob, permille (default 1000 = exactly 1): factor on the composed transmittance in the
                                     * four-wave forward's test "did this pixel cross T < 1e-4 inside the part"; a value below 1000
                                     * sends pixels that are still live through a second, sequential take of their part and the fold's
                                     * continuation behind it (the path a pixel within rounding of the threshold takes once in ~1e7).
                                     * Results stay within the bars of GS_TUNE_FWD_FOUR_WAVES (sequential instead of composed sums) */
    GS_TUNE_RENDER_ONLY = 15,       /* 1: the fused forwards of this context render and keep nothing for a backward -- no checkpoints
                                     * (a third to a half of the blend forward's bytes on deep lists), no backward preparation riding in
                                     * the loss.  gs_render_backward* of such a forward return GS_ERR_NO_FORWARD.  For previews, snapshots
                                     * of a training run, and the forward-only config of BASELINE (the reference's own forward custom
                                     * function saves nothing either: its VJP walks the lists backwards).  Same image, bit for bit */
    GS_TUNE_FWD_PAIR = 16,          /* fused blend forward with a STAGING wave beside every sweeping wave (two-wave workgroups: one loads,
                                     * culls and compacts chunk c + 1 into LDS while the other blends chunk c; one barrier per chunk):
                                     * -1 (default) = where the lists are deep (>= 4500 pairs per 16 x 16 block in the context's previous
                                     * forward: a scene grown to the schedule's cap, 0.89 -> 0.80 ms), 0 = never, 1 = always (12
                                     * workgroups per CU), 2..16 = always, with that many workgroups per CU.  The arithmetic and its order are the one-wave kernel's: same image, nContrib and
                                     * checkpoints, bit for bit.  Images small enough for GS_TUNE_FWD_FOUR_WAVES keep that kernel */
    GS_TUNE_FWD_SLOW_SLOT = 17,     /* the one-wave blend forward's persistent waves that sit in a hardware wave slot >= this value take their
                                     * static first item and nothing from the work queues (default 3; 16 = every wave pops).  gfx950's
                                     * SIMD arbiter favours its lower slots (measured: slot 3 runs at half slot 0's pace), and the launch
                                     * used to end with the slow slots' second items.  Same bits (who blends an item changes nothing) */
    GS_TUNE_TRIM_RECTS = 18,        /* 2 (default) / 1: at 16 x 16 tiles the fused forward bins a Gaussian on the tiles of the reference's 3-sigma
                                     * square (GaussianRenderer.swift get_rect, tile_rect) that the axis-aligned box of its ellipse
                                     * q <= 40.3 reaches -- beyond it the blend's staging drops the entry for every quadrant anyway
                                     * (weight < 2^-29): alpha is the untrimmed forward's bit for bit (one-wave forward), colour and depth to the rounding of
                                     * their per-chunk sums (a chunk is 64 list positions: the kept entries group differently; <= 4e-7
                                     * relative), gradients to the noise of their float atomics.  11.5 % fewer pairs to sort on the bench
                                     * scene (step 0.786 -> 0.776 ms, the scene grown to 1 M 3.02 -> 2.87).  What changes is what the fused path
                                     * reports about its lists: M (gs_last_stats), nContrib (gs_copy_last_contrib), gs_copy_block_work
                                     * and gs_tile_bin_export of a fused forward count positions in the TRIMMED lists, as they already
                                     * do under block lists.  2 cuts the box further: its tile rows in four groups, each with the columns
                                     * the ellipse reaches on the group's pixel rows (an elongated, tilted splat never sees the box's
                                     * corners: 14 % fewer pairs again on the bench scene); 1 keeps the box.  0: the reference's lists,
                                     * position for position (tests that hold M and nContrib to the oracle run with 0).  The op-level
                                     * gs_tile_bin is always the reference's */
    GS_TUNE_POISON_CHECKPOINTS = 14, /* TEST knob: 1 = the checkpoint arena is filled with NaN in front of every fused forward, so a
                                     * backward that reads a checkpoint lane its forward did not write shows up as NaN gradients */
    GS_TUNE_DEPTH_GRADIENT = 6      /* 1 (default): gs_render_backward* may get a cot_depth.  0: the caller promises NULL (the
                                     * default training case, GaussianTrainer.swift:280, 949: lambda_depth = 0); the forward
                                     * then saves 4 instead of 5 floats per pixel and 64 list entries, and a backward that
                                     * does bring a cot_depth is refused with GS_ERR_INVALID_ARG */
} gs_tuning;
int gs_ctx_set_tuning(gs_ctx* ctx, int knob, long long value);

/* Number of 16x16 pixel blocks of the fused path (block lists: the blocks enumerated per tile, gs_ctx_create). */
int gs_block_count(gs_ctx* ctx, int* n);
/* Copies the last fused forward's per-block sweep length (max nContrib over the block's pixels) to a device buffer. */
int gs_copy_block_work(gs_ctx* ctx, uint32_t* out);
/* Copies the last fused forward's per-pixel nContrib (u32 [H*W], the reference's lastContrib) to a device buffer. */
int gs_copy_last_contrib(gs_ctx* ctx, uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_H */
