#!/usr/bin/env python3
"""Benchmark of the render/backward hot path (see DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W [--config NAME] [--mode train|fwdbwd|forward]

N > 1 without a launcher (WORLD_SIZE unset): this process starts `python -m torch.distributed.run --nproc-per-node N`
on itself as a CHILD (before anything touches the GPU), relays rank 0's JSON line and exits with the child's code.
Under a launcher (the driver's `torch.distributed.run ... bench.py --gpus N`) it is one rank.

BASELINE.json configs -> default mode:
    c1_10k_400    forward   single-view forward render (projection + binning + blend), metric fwd Mpix/s
    c2_100k_800   fwdbwd    projection + tile blend forward and backward of one view (no loss, no optimizer), views/s
    c3_300k_800   train     full train step: forward, L1 + DSSIM loss, backward, (gradient exchange,) Adam, densify /
                            prune at the reference cadence, views/s   <- the default, BASELINE.json's metric
    c5_garden_2m  train
A step = one pass of that path over one view per rank; all inputs are resident in HBM before timing starts.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy ceiling 6290
VALU_PEAK_TFLOPS = 157.3
DEFAULT_MODE = {"c1_10k_400": "forward", "c2_100k_800": "fwdbwd", "c3_300k_800": "train", "c5_garden_2m": "train",
                "c3_grown_1m": "train"}       # (c3_grown_1m: not a BASELINE config -- the scene of c3 once densification has grown it)
STAGES_OF_MODE = {"forward": ("proj_fwd", "bin", "blend_fwd"),
                  "fwdbwd": ("proj_fwd", "bin", "blend_fwd", "blend_bwd", "proj_bwd"),
                  "train": ("proj_fwd", "bin", "blend_fwd", "loss", "blend_bwd", "proj_bwd", "adam")}
KERNEL_OF_STAGE = {"blend_bwd": "blend_bwd_v2_kernel", "blend_fwd": "blend_fwd_v2", "proj_fwd": "proj_fwd_fused_kernel",
                   "proj_bwd": "proj_bwd_fused_kernel", "adam": "adam_kernel", "loss": "loss_fused_kernel",
                   "bin": "radix_scatter_kernel<false>"}


def algorithmic_bytes(N, K, M, P, T):
    """SURVEY.md 8(d) / BASELINE.md 4, per launch: what a straightforward implementation of the stage must move."""
    return dict(
        proj_fwd=N * (44 + 12 * K) + N * 64,
        proj_bwd=N * (44 + 12 * K) + N * 44 + N * (40 + 12 * K),
        bin=N * 24 + M * 12 * (2 * -(-(32 + max(1, (T - 1).bit_length())) // 8) + 1) + T * 8,
        blend_fwd=M * 48 + P * 24,
        blend_bwd=M * 48 + P * 44 + M * 88 + N * 44,
        loss=3 * P * 32 + 3 * P * 40,
        adam=N * (11 + 3 * K) * 28,
    )


def designed_bytes(N, K, M, M_eff, P, T, S_fwd, fused_adam, colour_riders=False, dp_form=False):
    """Bytes THIS implementation is built to move per launch, stage by stage:
    proj_fwd   read the raw parameters (44 + 12K per Gaussian), write packed12 48 + rect 8 + touched 4 + key/val 8
    bin        depth sort of N (key, value) records: up to 4 passes x (histogram read 4 + scatter read 8 + write 8) = 80 N;
               scan + expansion: 16 N read, 4 M written (one packed word per pair); one-pass tile sort: histogram read
               4 M, u16 count table written, prefixed and read back 3 x 2 M, scatter read 4 M + write 4 M = 18 M; the tile
               ranges come out of the per-tile totals (8 T)
    blend_fwd  per traversed block-splat a 4-B index + a 48-B record; 28 B per pixel out; one (T, R, G, B[, D]) checkpoint
               per pixel of a block every 64 list entries the block went through (S_fwd of them)
    blend_bwd  index + record again, the checkpoint read back, 44 B per pixel of cotangents / state, one 44-B atomic row
               per traversed block-splat, the 64-B accumulator rows cleared
    proj_bwd   parameters + the 64-B accumulator row in; gradients out (or, fused Adam: parameter and both moments
               read and written in place, no gradient arena)
    loss       render + target in, cotangent out, 12 B each per pixel (the SSIM maps never leave the kernel)
    adam       7 arena passes (p, g, m, v in; p, m, v out)
    colour_riders: the projection stage's own launch reads the 44 geometry bytes only (the SH rows are read by rider workgroups
    inside the binning kernels' launches).  dp_form: the data-parallel step's projection backward and Adam are FOUR kernels
    under two stage names, timed per call: no per-launch byte count describes a "stage" there (None)."""
    E = N * (11 + 3 * K)
    if dp_form:
        return dict(designed_bytes(N, K, M, M_eff, P, T, S_fwd, False, colour_riders), proj_bwd=None, adam=None)
    return dict(
        proj_fwd=(N * 44 + N * 68) if colour_riders else (N * (44 + 12 * K) + N * 68),
        bin=N * 96 + M * 18 + T * 8,
        blend_fwd=M_eff * 52 + P * 28 + S_fwd * 4 * 256 * 4,
        blend_bwd=M_eff * (52 + 44) + S_fwd * 4 * 256 * 4 + P * 44 + N * 64,
        proj_bwd=(N * (44 + 12 * K) + N * 64 + E * 24) if fused_adam else (N * (44 + 12 * K) + N * 64 + E * 4),
        loss=3 * P * 12,
        adam=0 if fused_adam else E * 28,
    )


def survey_bytes(N, K, M, M_eff, P, T, fused_adam=False, colour_riders=False, dp_form=False):
    """SURVEY 8(d)'s algorithmic bytes of every stage AS THE STAGE RUNS HERE: {stage: (bytes or None, note)}.  None = the
    stage no longer moves what the survey's formula counts, so a rate over its time would be a rate nothing moved:
      * blend stages: the formula on the M_eff block-splats actually traversed (a list is left at its last contributing entry);
      * proj_bwd with the Adam update fused in (single-device train step): proj_bwd + adam - 2 E 4 - N (44 + 12K),
        E = N (11 + 3K) -- the gradient arena is neither written by the backward nor read by Adam, and the parameters, which
        the survey's two formulas read once each, are read ONCE by the fused kernel (round 6: rounds 4-5 still counted the
        second read, which put c5's line at 0.78 of the HBM peak where its own counters say 0.68);
      * adam, fused: no stage of its own;
      * proj_fwd under colour riders: the SH rows (12 K of the 408 B per Gaussian) are read by rider workgroups inside the
        binning kernels' launches, the stage's time is the geometry half alone;
      * bin: the survey models a 44-bit LSD radix sort of the M (tile, depth) records, 6 passes x 12 B; what runs sorts the N
        depth records, expands, and moves every pair ONCE through a one-pass tile sort."""
    a = algorithmic_bytes(N, K, M, P, T)
    e = algorithmic_bytes(N, K, M_eff, P, T)
    E = N * (11 + 3 * K)
    traversed = "on the M_eff block-splats actually traversed"
    if dp_form:
        # the data-parallel form of the step: colour cotangents + geometry-only projection backward under "proj_bwd", SH rebuild +
        # Adam and the geometry Adam under "adam" -- two launches per stage name, the stage time is their mean per call
        split = ("data-parallel form: two kernels per stage name (colour cotangents + geometry backward; SH rebuild + Adam + geometry "
                 "Adam), the stage time is the mean per call -- no launch moves the survey's bytes")
        return dict(survey_bytes(N, K, M, M_eff, P, T, False, colour_riders), proj_bwd=(None, split), adam=(None, split))
    return {
        "proj_fwd": (None, "colour riders: the SH rows are read by rider workgroups in the binning kernels' launches; this stage's "
                           "time holds the geometry half only") if colour_riders else (a["proj_fwd"], None),
        "bin": (None, "the survey's model is a 6-pass 44-bit radix sort of M records; what runs is a depth sort of N records + one "
                      "pass of the pairs through a 4096-bin tile sort (see GBps_designed_bytes)"),
        "blend_fwd": (e["blend_fwd"], traversed),
        "blend_bwd": (e["blend_bwd"], traversed),
        "proj_bwd": (a["proj_bwd"] + a["adam"] - 2 * E * 4 - N * (44 + 12 * K),
                     "Adam fused in: proj_bwd + adam - 2 N (11 + 3K) 4 (no gradient arena round trip) - N (44 + 12K) (the parameters are read once, not twice)")
                    if fused_adam else (a["proj_bwd"], None),
        "loss": (a["loss"], None),
        "adam": (None, "fused into the projection backward (no launch of its own)") if fused_adam else (a["adam"], None),
    }


def rate_gbps(nbytes, ms):
    """GB/s of nbytes per launch over ms -- or None where there is no time or no byte count.  A figure above the HBM peak is
    returned as it is: sanitize_fractions nulls it in the line AND lists it in `accounting_violations` with its raw value
    (round 5 returned None here, so a byte model that overcounts vanished from the line without a trace: the advisor's finding)."""
    if nbytes is None or not ms or ms <= 0:
        return None
    return round(nbytes / ms / 1e6, 1)


def sanitize_fractions(obj, path=""):
    """No fraction above 1 and no rate above the HBM peak anywhere in a result line: offending values are set to None and
    their paths returned (bench.py puts the list into the line as `accounting_violations`; the tests want it empty)."""
    bad = []
    if isinstance(obj, dict):
        for k, v in list(obj.items()):
            here = f"{path}.{k}" if path else k
            if isinstance(v, (dict, list)):
                bad += sanitize_fractions(v, here)
            elif isinstance(v, (int, float)) and not isinstance(v, bool):
                if (("frac" in k or k == "exec_lane_occupancy") and v > 1.0) or (k.startswith("GBps") and v > HBM_PEAK_GBS) or \
                        (k == "achieved" and obj.get("unit") == "GB/s" and v > HBM_PEAK_GBS):
                    obj[k] = None
                    bad.append(f"{here} = {v}")
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            bad += sanitize_fractions(v, f"{path}[{i}]")
    return bad


def csrc_sha():
    """Identity of the kernel sources: a PMC summary taken on other sources says nothing about this build."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gaussiansplattingmlx_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launcher_command(n_gpus, argv, port=None):
    """The command the driver itself uses for N > 1 (one rank per GPU over RCCL), applied to this file."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), os.path.abspath(__file__), *argv]


def visible_gpus():
    """GPUs this process could use, counted WITHOUT bringing up a GPU runtime in this process (the parent of an N-rank run
    starts the ranks as children and must itself stay off the GPU: torch.cuda.device_count() falls through to
    hipGetDeviceCount when amdsmi is not importable, which opens /dev/kfd and holds it for the whole run -- round 4's advisor).
    The kernel driver's topology first (a node with SIMDs is a GPU; *_VISIBLE_DEVICES masks applied), a short-lived child
    process that asks torch otherwise."""
    mask = None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = len([x for x in v.split(",") if x.strip() != ""])
            mask = n if mask is None else min(mask, n)
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    try:
        have = 0
        for d in os.listdir(nodes):
            props = dict(ln.split(None, 1) for ln in open(os.path.join(nodes, d, "properties")) if " " in ln.strip())
            if int(props.get("simd_count", "0")) > 0:
                have += 1
        # the topology lists every GPU of the HOST, whatever this container's device cgroup exposes: bound it by the render
        # nodes the process can actually open (round 5's advisor: on a box that exposes 1 of 8 the pre-check passed and the
        # ranks died after a full torchrun start-up instead)
        usable = usable_render_nodes()
        if usable is not None:
            have = min(have, usable)
        return have if mask is None else min(have, mask)
    except (OSError, ValueError):
        pass
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True,
                             timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def usable_render_nodes(dri="/dev/dri"):
    """/dev/dri/renderD* nodes this process may open for reading and writing (None: no such directory to ask)."""
    try:
        names = [n for n in os.listdir(dri) if n.startswith("renderD")]
    except OSError:
        return None
    return sum(1 for n in names if os.access(os.path.join(dri, n), os.R_OK | os.W_OK))


def too_few_gpus_message(n_gpus, have):
    return (f"bench.py: --gpus {n_gpus} needs {n_gpus} visible GPUs (one rank per GPU over RCCL) but this box has {have}: RCCL refuses "
            f"two ranks on one device, so the run would die in rank {have}'s torch.cuda.set_device / ncclCommInitRank with nothing "
            "measured.  Run it on a node with enough GPUs, or rehearse the launcher on one card with "
            "`GSPLAT_BENCH_DEVICE=0 python bench.py --gpus N --backend gloo` (all ranks on card 0, gloo collectives).")


def self_launch(n_gpus, argv):
    """Parent of an N-rank run: starts the ranks as children, relays rank 0's line.  Touches no GPU itself.  The ranks'
    stderr is passed through as it comes and its last 40 lines are repeated behind a failure, so that the record of a
    failed run says why."""
    import collections
    import threading
    if not os.environ.get("GSPLAT_BENCH_DEVICE"):
        have = visible_gpus()
        if have < n_gpus:
            print(too_few_gpus_message(n_gpus, have), file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(launcher_command(n_gpus, argv), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True)
    tail = collections.deque(maxlen=400)

    def pump():
        for ln in proc.stderr:
            tail.append(ln.rstrip("\n"))
            sys.stderr.write(ln)
            sys.stderr.flush()

    t = threading.Thread(target=pump, daemon=True)
    t.start()
    line = None
    for ln in proc.stdout:
        ln = ln.rstrip("\n")
        if ln.startswith('{"metric"'):
            line = ln
        else:
            print(ln, file=sys.stderr)
    rc = proc.wait()
    t.join(timeout=10)
    if rc != 0:
        # what the ranks themselves said comes BEFORE torch.distributed.run's own failure report (a banner of ~40 lines)
        lines = list(tail)
        cut = next((i for i, ln in enumerate(lines) if "elastic/multiprocessing/api.py" in ln and "failed" in ln), len(lines))
        own = lines[max(cut - 40, 0):cut]
        print(f"bench.py: the {n_gpus}-rank run failed with exit code {rc}; the last {len(own)} lines of the ranks' stderr"
              f"{' (the launcher report behind them is above)' if cut < len(lines) else ''}:", file=sys.stderr)
        for ln in own:
            print("    | " + ln, file=sys.stderr)
        return rc
    if line is None:
        print("bench.py: the ranks finished without a result line", file=sys.stderr)
        return 1
    print(line)
    return 0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c3_300k_800", choices=sorted(DEFAULT_MODE))
    ap.add_argument("--mode", default=None, choices=["train", "fwdbwd", "forward"],
                    help="what a step is (default: what BASELINE.json names for the config)")
    ap.add_argument("--views", type=int, default=100, help="training views cycled through (SURVEY 8(d): 100); every view is "
                    "visited once before anything is timed, as after the first epoch of a 30000-iteration run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "several ranks on one card together with GSPLAT_BENCH_DEVICE)")
    ap.add_argument("--no-depth-cuts", action="store_true", help="bin every tile list in full (A/B of the depth cuts)")
    ap.add_argument("--cut-min-dropped", type=int, default=None, help="renderer.cutMinDropped: a view keeps binning under depth cuts "
                    "only if they leave out at least this many pairs (default: the renderer's)")
    ap.add_argument("--no-view-hints", action="store_true",
                    help="do not reuse a view's previous per-block sweep lengths to order the forward's items")
    ap.add_argument("--dp-exchange", default="sh_compressed", choices=["sh_compressed", "allreduce"],
                    help="gradient exchange for --gpus > 1 (trainer.py)")
    ap.add_argument("--dp-impl", default=os.environ.get("GSPLAT_DP_IMPL", "torch"), choices=["torch", "native"],
                    help="who issues the collectives of a data-parallel step: torch.distributed, or the library itself "
                         "(gs_dp_step: RCCL on its own side stream; the process group then only carries the RCCL id)")
    ap.add_argument("--no-dp-balance", action="store_true", help="--gpus > 1: deal the views to the steps in index order instead of in "
                    "the cost-balanced order (trainer.balanced_view_order: the views of one step cost about the same)")
    ap.add_argument("--dp-single", action="store_true", help="with --gpus 1: run the DATA-PARALLEL step, collectives included, on a "
                    "1-rank group (torch: a 1-rank nccl process group; native: a 1-rank RCCL communicator inside the library) -- "
                    "a rehearsal of the exchange code path and of the `exchange` block on one card, not a headline number")
    ap.add_argument("--views-per-step", type=int, default=1, help="train mode: views EVERY RANK brings to a step (trainer views_per_rank): "
                    "one update from the mean loss over ranks x views.  `--views-per-step 8` on one GPU is BASELINE config 4's "
                    "arithmetic -- eight views, one update, the SH rebuild and the split Adam over eight blocks -- on one card "
                    "(with --dp-single the blocks also go through a 1-rank all-gather); a side line, never the headline")
    ap.add_argument("--tile", type=int, default=16, help="square tile size; 16 = the fused wave-per-block path, anything "
                    "that is not a multiple of 16 (the reference app's W/4 = 200) = the same kernels on block lists "
                    "(GSPLAT_BLOCK_LISTS=0: the generic blend kernels)")
    ap.add_argument("--keep-checkpoints", action="store_true", help="--mode forward: time the forward a backward could follow (it writes "
                    "checkpoints of the running state) instead of the render-only one")
    ap.add_argument("--two-pass-tile-sort", action="store_true", help="A/B: the two 8-bit tile-sort passes instead of the one-pass sort")
    ap.add_argument("--ppl", default="", help="fwd,bwd pixels per lane of the op-level kernels (tuning)")
    ap.add_argument("--residency", default="", help="fwd waves/SIMD, bwd waves/CU of the persistent kernels (tuning)")
    args = ap.parse_args(argv)
    args.mode = args.mode or DEFAULT_MODE[args.config]
    return args


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE)")

    # stdout carries ONE line, the result: whatever else writes to file descriptor 1 from here on (RCCL prints a version
    # banner there when its first communicator comes up, libraries print warnings) goes to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if os.environ.get("GSPLAT_BENCH_DEVICE"):          # rehearsal only: several ranks on one card
        local_rank = int(os.environ["GSPLAT_BENCH_DEVICE"])
    elif torch.cuda.device_count() < world:
        raise SystemExit(too_few_gpus_message(world, torch.cuda.device_count()))
    if world > 1 and args.backend == "nccl":
        # ProcessGroupNCCL then keeps start / end events per collective: Work._get_duration() for the `exchange` block
        os.environ.setdefault("TORCH_NCCL_ENABLE_TIMING", "1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    dist = None
    dp_single = args.dp_single and world == 1 and args.mode == "train"
    dp_boot = None
    if dp_single and args.backend == "nccl":
        os.environ.setdefault("TORCH_NCCL_ENABLE_TIMING", "1")
    if world > 1 or (dp_single and args.dp_impl == "torch"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            dist.init_process_group(args.backend, rank=0, world_size=1, **(dict(device_id=dev) if args.backend == "nccl" else {}))
        else:
            dist.init_process_group(args.backend, **(dict(device_id=dev) if args.backend == "nccl" else {}))
        pg = dist.group.WORLD

    from gaussiansplattingmlx_amd.renderer import GaussianRenderer
    from gaussiansplattingmlx_amd.scenes import CONFIGS, GROW_ITERATIONS, make_config, perturb
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel, balanced_view_order, view_for

    mode = args.mode
    vps = args.views_per_step if mode == "train" else 1
    if vps > 1 and args.dp_impl == "native" and (world > 1 or dp_single):
        raise SystemExit("bench.py: --views-per-step > 1 takes the torch issuer (--dp-impl torch) or no group at all")
    idx, N, W, H, kind = CONFIGS[args.config]
    params, cams, _ = make_config(args.config, n_views=args.views)
    K = 25
    ts = args.tile
    r = GaussianRenderer(4, W, H, (ts, ts), False, device=local_rank)
    r.depthCuts = not args.no_depth_cuts
    if args.cut_min_dropped is not None:
        r.cutMinDropped = args.cut_min_dropped
    # the forward-only config renders and keeps nothing for a backward (GS_TUNE_RENDER_ONLY: no checkpoints -- a forward that a
    # backward is to follow writes one per 64 list entries and quadrant); --keep-checkpoints times the training forward instead
    render_only = mode == "forward" and not args.keep_checkpoints
    if render_only:
        r.setTuning(render_only=1)
    if os.environ.get("GSPLAT_FWD_PAIR"):          # A/B of the staging-wave forward (GS_TUNE_FWD_PAIR): workgroups per CU, 0 = off
        r.setTuning(fwd_pair=int(os.environ["GSPLAT_FWD_PAIR"]))
    if os.environ.get("GSPLAT_TRIM_RECTS"):        # A/B (GS_TUNE_TRIM_RECTS): 0 = the reference's whole 3-sigma squares, 1 = cut by the ellipse's box, 2 (default) = and into row groups
        r.setTuning(trim_rects=int(os.environ["GSPLAT_TRIM_RECTS"]))
    if os.environ.get("GSPLAT_FWD_SLOW_SLOT"):     # A/B: first hardware wave slot whose waves take no queue items (GS_TUNE_FWD_SLOW_SLOT; 16 = off)
        r.setTuning(fwd_slow_slot=int(os.environ["GSPLAT_FWD_SLOW_SLOT"]))
    if args.two_pass_tile_sort:
        r.setTuning(wide_tile_sort=0)
    if args.ppl:
        f, b = (int(x) for x in args.ppl.split(","))
        r.setTuning(op_fwd_ppl=f, op_bwd_ppl=b)
    if args.residency:
        f, b = (int(x) for x in args.residency.split(","))
        r.setTuning(fwd_quadrants=int(f >= 100), fwd_waves_per_simd=f % 100, bwd_waves_per_cu=b)
    # workspace and parameter arenas carry 1.5x headroom so the densify event in the timed region does not reallocate
    headroom = 1.5 if mode == "train" else 1.0
    # c3_grown_1m: the scene is c3's, grown by the trainer's own schedule before anything is timed (scenes.py)
    grow = GROW_ITERATIONS if args.config == "c3_grown_1m" and mode == "train" else 0
    if grow:
        headroom = 1_600_000 / N
    pair_cap = int(os.environ.get("GSPLAT_BENCH_PAIR_CAP", 0)) or (48 << 20 if args.config == "c3_grown_1m" else
                                                                    {0: 2 << 20, 1: 12 << 20, 2: 24 << 20}.get(idx, 96 << 20))
    r.reserve(int(N * headroom), pair_cap)

    # targets: renders of a perturbed copy of the scene (non-trivial gradients), produced before timing
    V = len(cams)
    gcams = [r._camera(c.worldViewTransform, c.projectionMatrix, c.cameraCenter, c.FoVx, c.FoVy, c.focalX, c.focalY)
             for c in cams]
    targets = []
    if mode != "forward":
        tgt_params = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
        for cam in cams:
            targets.append(r.renderForward(tgt_params, cam).render.clone())
        del tgt_params
    model = GaussModel(params, dev, capacity=int(N * headroom))
    trainer = None
    cots = []
    if mode == "train":
        if dp_single and args.dp_impl == "native":
            import ctypes
            from gaussiansplattingmlx_amd import _lib as gslib
            uid = ctypes.create_string_buffer(gslib.GS_DP_UNIQUE_ID_BYTES)
            if r.lib.gs_dp_unique_id(uid) != 0:
                raise SystemExit("bench.py: gs_dp_unique_id failed (RCCL not loadable)")
            dp_boot = (uid.raw, 0, 1)
        trainer = GaussianTrainer(model, r, iterationCount=30000, process_group=pg, dp_exchange=args.dp_exchange,
                                  exchange_impl=args.dp_impl, exchange_when_single=dp_single, dp_bootstrap=dp_boot,
                                  views_per_rank=vps)
        # densify / prune runs at the reference cadence (every 100 iterations inside [500, 15000]); the iteration
        # counter starts so that iteration 600 falls in the middle of the timed region
        trainer.iteration = max(600 - args.warmup - args.steps // 2, 0)
        if os.environ.get("GSPLAT_BENCH_IT0"):          # diagnostics only: where the iteration counter starts
            trainer.iteration = int(os.environ["GSPLAT_BENCH_IT0"])
        if grow:
            trainer.iteration = 450
    elif mode == "fwdbwd":
        # the cotangent of each view's render under the training loss, computed once: the timed step is projection +
        # binning + blend forward and their backward only (BASELINE.json configs[1])
        for v in range(V):
            res = r.renderForward(model.getParams(), gcams[v])
            _, cot, _ = r.lossForwardBackward(res.render, targets[v], 0.2)
            cots.append(cot.clone())
        grads = {k: torch.empty_like(v) for k, v in model.getParams().items()}
    # Every view is visited once before anything is timed: a 30000-iteration run passes its 100 views 300 times, so what a
    # step costs is what it costs on a view that has been seen before -- the view's sweep-length hints (launch order of the
    # blend forward), its depth cuts and, in train mode, its target's SSIM statistics exist.  The visit is a forward (+ the
    # loss kernel, which also carries the backward's preparation and renews the cuts): nothing is updated.  It is also
    # where the trainer's capacity check of a view's first forward happens (one wait), outside the warm-up and the timed
    # region whatever their lengths.
    pre_visits = 0
    if not args.no_view_hints:
        # (two passes: a view's second forward is the probe of its depth cuts -- renderer.CutPolicy --, after which the
        # policy has decided for the next 64 visits; in a run of a few dozen steps over 100 views every timed step would
        # otherwise be such a probe, one in 65 of a long run's)
        for _ in range(2 if r.depthCuts else 1):
            for v in range(V):
                res = r.renderForward(model.getParams(), gcams[v], viewKey=v, wantDepth=mode != "train")
                if mode == "train":
                    r.lossForwardBackward(res.render, targets[v], 0.2, out=dict(loss=trainer._loss, cotColor=trainer._cot),
                                          targetKey=v)
                r.forwardMissed()          # (reports the forward to the view's cut policy; nothing consumes a missed one here)
            pre_visits += 1
    r.sync()          # raises if the reserve was too small for any of the renders above
    if trainer is not None and not args.no_view_hints:
        trainer._checked_views.update(range(V))

    # Data-parallel: a step takes as long as its slowest rank's view, so the views are dealt to the steps in an order in
    # which the `world` views of a step cost about the same (trainer.balanced_view_order; cost = the block-entries the view's
    # last forward traversed, read off its hint buffer -- the pre-visits have filled it).  Rank 0's order is everybody's.
    order = list(range(V))
    balanced = world > 1 and mode == "train" and not args.no_dp_balance and not args.no_view_hints and pre_visits > 0
    if balanced:
        nblk = ((W + 15) // 16) * ((H + 15) // 16)
        costs = [int(r._work_hints[v][:nblk].to(torch.int64).sum().item()) for v in range(V)]
        ot = torch.tensor(balanced_view_order(costs), dtype=torch.int64, device=dev)
        dist.broadcast(ot, src=0)
        order = [int(x) for x in ot.cpu()]

    def view_of(i, q):
        return order[view_for(i, q, world, V)]

    def step(i):
        i += grow                      # (the growth phase took the steps [0, grow))
        if vps > 1:
            # step i takes the views [i world V, (i + 1) world V) of the order, rank-major: this rank's are V consecutive ones
            mine = [order[((i * world + rank) * vps + j) % V] for j in range(vps)]
            everyone = [cams[order[((i * world + q) * vps + j) % V]] for q in range(world) for j in range(vps)]
            trainer.trainStep([gcams[v] for v in mine], [targets[v] for v in mine],
                              viewKey=None if args.no_view_hints else mine, stepCameras=everyone)
            return
        v = view_of(i, rank)
        key = None if args.no_view_hints else v
        if mode == "train":
            trainer.trainStep(gcams[v], targets[v], viewKey=key,
                              stepCameras=[cams[view_of(i, q)] for q in range(world)] if (world > 1 or dp_single) else None)
        elif mode == "fwdbwd":
            r.renderChecked(model.getParams(), gcams[v], viewKey=key)
            r.renderBackward(cots[v], out=grads)
        else:
            r.renderChecked(model.getParams(), gcams[v], viewKey=key)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if grow:
        # the trainer's own iterations 450 .. 450 + grow, densify / prune at the reference cadence: every rank takes the same
        # (data-parallel) steps, so the replicas grow the same scene
        for i in range(-grow, 0):
            step(i)
        r.sync()
    if trainer is not None:
        it0 = trainer.iteration
    stage_names = STAGES_OF_MODE[mode]
    # warm-up; it also names the two stages with the largest device time, which carry HIP events through the timed region
    # (each recorded stage costs two event packets on the stream per step; the full breakdown is taken behind the region)
    r.profile(stage_names)
    if trainer is not None:
        trainer.prewarmDensify()
    for i in range(args.warmup):
        step(i)
    r.sync()
    wprof = r.profileRead()
    by_warmup = sorted(stage_names, key=lambda k: -wprof[k][0] / max(wprof[k][1], 1)) if args.warmup > 0 else list(stage_names[::-1])
    live_stages = by_warmup[:2]
    barrier()
    r.profile(live_stages)
    misses0 = trainer.forwardMisses if trainer else 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    densify_info = None
    cut_info = {"enabled": not args.no_view_hints and not args.no_depth_cuts}
    if trainer is not None:
        it_lo, it_hi = it0 + args.warmup, it0 + args.warmup + args.steps          # timed iterations [it_lo, it_hi)
        densify_events = [i for i in range(it_lo, it_hi) if i % trainer.split_and_prune_per_iteration == 0
                          and trainer.densifyFromIter <= i <= trainer.densifyUntilIter]
        cut_info["forwards_repeated_in_timed_region"] = trainer.forwardMisses - misses0
        pol = list(r._cut_policy.values())
        # where the views stand at the end of the timed region: binning under their cuts / sitting out (their cuts left out too
        # little at the last probe) / without cuts since the last event that added Gaussians
        cut_info["views_under_cuts"] = sum(1 for q in pol if q.sit_out == 0 and q.since_empty >= 1)
        cut_info["views_sitting_out"] = sum(1 for q in pol if q.sit_out > 0)
        cut_info["views_without_cuts"] = sum(1 for q in pol if q.sit_out == 0 and q.since_empty == 0)
        cut_info["min_pairs_left_out_for_cuts"] = int(r.cutMinDropped)
        densify_info = {"events_in_timed_region": len(densify_events), "at_iterations": densify_events,
                        "last_stats": trainer.lastDensifyStats, "N_after": model.N}
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    live = r.profileRead()
    # per-stage breakdown, outside the timed region; in a data-parallel run the same steps time the exchange
    r.profile(stage_names)
    timed_exchange = trainer is not None and trainer._exchange
    if timed_exchange:
        trainer.exchangeTimingBegin()
    for i in range(min(args.steps, 10)):
        step(args.warmup + args.steps + i)
    prof = r.profileRead()
    exchange = trainer.exchangeTimingRead() if timed_exchange else None
    r.profile(False)
    r.sync()
    # spread of single steps (rank 0's device time between per-step events), outside the timed region as well
    nq = 30
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(nq + 1)]
    marks[0].record()
    for i in range(nq):
        step(args.warmup + args.steps + 10 + i)
        marks[i + 1].record()
    torch.cuda.synchronize()
    each = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(nq))
    step_spread = {"p10": round(each[nq // 10], 4), "p50": round(each[nq // 2], 4), "p90": round(each[(nq * 9) // 10], 4)}
    loss = [float(x) for x in trainer._loss.cpu()] if trainer else None
    # replicas must hold bit-identical parameters (same summed gradients, same Adam, same densify decisions)
    replicas_identical = None
    if world > 1:
        chk = torch.stack([model.arena.double().sum(), model.arena.double().abs().sum(),
                           torch.tensor(float(model.N), dtype=torch.float64, device=dev)])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi))

    # workload statistics of view 0, binned as the timed steps bin it (under the view's depth cuts where the policy
    # applies them) + forward-only rate (outside the timed region)
    r.renderChecked(model.getParams(), gcams[0], viewKey=None if args.no_view_hints else 0)
    st = r.stats()
    if st["overflow"]:
        raise SystemExit(f"bench.py: the forward needed M={st['M']} pairs but only {st['capM']} were reserved "
                         "(GSPLAT_BENCH_PAIR_CAP): nothing measured above is valid")
    overflow_recoveries = trainer.overflowRecoveries if trainer else 0
    last = r.lastContrib().to(torch.int64)
    P, T = W * H, ((W + ts - 1) // ts) * ((H + ts - 1) // ts)
    block_lists = r.blockLists         # a tile size that is not a multiple of 16: the fused kernels on the 16x16 blocks of every tile
    fast16 = ts % 16 == 0 or block_lists
    bs = 16 if fast16 else ts          # the unit that sweeps a list together: a 16x16 block (fused path) or the whole tile
    if block_lists:                    # (blocks are enumerated per tile, not a regular grid over the image: ask the library)
        tile_max = r.blockWork().to(torch.int64)
        T = int(tile_max.numel())
    else:
        Hp, Wp = -(-H // bs) * bs, -(-W // bs) * bs
        pad = torch.zeros(Hp, Wp, dtype=torch.int64, device=dev)
        pad[:H, :W] = last
        tile_max = pad.view(Hp // bs, bs, Wp // bs, bs).amax(dim=(1, 3))
    M_eff = int(tile_max.sum().item())
    # checkpoints written per forward (fused path only)
    S_fwd = int(torch.clamp((tile_max + 63) // 64 - 1, min=0).sum().item()) if fast16 and not render_only else 0
    mean_contrib = float(last.double().mean().item())
    nf = 20
    torch.cuda.synchronize()
    tf0 = time.perf_counter()
    for i in range(nf):       # the views keep their keys: the forward's deepest-first launch order is part of the path
        r.renderForward(model.getParams(), gcams[i % V], viewKey=None if args.no_view_hints else i % V, depthCuts=False)
    torch.cuda.synchronize()
    fwd_ms = (time.perf_counter() - tf0) / nf * 1e3
    r.sync()

    # every collective is behind us: leave the groups in order (the library's communicator first)
    if trainer is not None:
        trainer.closeExchange()
    n_ranks = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    if dist is not None and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    ms_per_step = elapsed / args.steps * 1e3
    M = st["M"]
    Nn = model.N                                  # after the timed region's densify event, if any
    stage_ms = {k: (prof[k][0] / max(prof[k][1], 1)) for k in stage_names}
    fused_adam = mode == "train" and stage_ms.get("adam", 1.0) == 0.0
    riders = fast16 and r.colourRidersActive(Nn, K)
    dp_form = trainer is not None and trainer._dp
    des = designed_bytes(Nn, K, M, M_eff, P, T, S_fwd, fused_adam, colour_riders=riders, dp_form=dp_form)
    # SURVEY 8(d)'s bytes of every stage as it runs here (blend: the traversed block-splats; projection backward with Adam
    # fused in: both formulas less the gradient arena's round trip) -- None where the stage no longer moves them
    surv = survey_bytes(Nn, K, M, M_eff, P, T, fused_adam=fused_adam, colour_riders=riders, dp_form=dp_form)
    # the roofline block describes the stage with the largest time in the breakdown taken right behind the timed region
    # (the warm-up's ranking can differ: c5's first steps bin without depth cuts); its launch time is the one measured
    # LIVE inside the timed region when the warm-up had it among its two largest stages, else the breakdown's
    # (among the stages that HAVE a byte model: the data-parallel form's projection backward and Adam are priced by neither
    # formula -- on a real step they are a tenth of the blend backward, but a rehearsal of several ranks time-slicing one card
    # can rank anything first)
    priced = [k for k in stage_names if surv[k][0] is not None or des[k] is not None]
    dom = max(priced or stage_names, key=lambda k: stage_ms[k])
    if dom in live_stages and live[dom][1] > 0:
        dom_ms, dom_src = live[dom][0] / live[dom][1], "HIP events on the library's stream inside the timed region"
    else:
        dom_ms, dom_src = stage_ms[dom], "HIP events over the steps right behind the timed region (not a live stage of it)"
    roof = roofline_block(dom, dom_ms, dom_src, surv, des, args.config, mode, ts, M_eff, float(bs * bs), fast16)
    # a stage that did not run on its own (Adam fused into the projection backward) has no rate
    stages = {k: {"ms": round(stage_ms[k], 4),
                  "GBps_survey_bytes": rate_gbps(surv[k][0], stage_ms[k]), "survey_bytes": surv[k][0], "survey_note": surv[k][1],
                  "GBps_designed_bytes": rate_gbps(des[k], stage_ms[k])}
              for k in stage_names}

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(mode, params, cams, W, H, [t.cpu().numpy() for t in targets])

    what = {"train": "train views/sec (full step: fwd + L1/DSSIM loss + bwd + Adam + densify/prune at the reference cadence)",
            "fwdbwd": "views/sec (projection + binning + tile blend, forward and backward of one view; no loss, no optimizer)",
            "forward": "fwd Mpix/s (single-view forward render: projection + binning + tile blend)"}[mode]
    scene = {"c1_10k_400": "Lego 400x400 10k random-init Gaussians", "c2_100k_800": "Lego 800x800 100k Gaussians",
             "c3_300k_800": "Lego 800x800 300k Gaussians", "c5_garden_2m": "Mip-NeRF-360 garden 1237x822 2M Gaussians",
             "c3_grown_1m": "Lego 800x800 300k Gaussians grown to the reference schedule's cap of 1M by the trainer's own iterations 450-1600"}[args.config]
    if mode == "forward":
        value, unit = world * args.steps * P / elapsed / 1e6, "Mpix/s"
    else:
        value, unit = world * vps * args.steps / elapsed, "views/s"
    out = {
        "metric": f"{what}, {scene}",
        "value": round(value, 3), "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: synthetic {'garden' if kind == 'garden' else 'Lego'} cameras {W}x{H}, N={N} "
                               f"{'random-init' if kind == 'random_init' else 'trained-like'} Gaussians{f' grown by {grow} untimed train iterations to N={model.N}' if grow else ''}, SH degree 4 (K=25), "
                               f"{ts}x{ts} tiles{' (block lists: the fused kernels on the 16x16 blocks of every tile)' if block_lists else '' if fast16 else ' (generic blend kernels)'}, {V} views, {vps} view{'s' if vps > 1 else ''} per rank per step{' (ONE update from the mean loss over them: config 4 arithmetic on one card, a side line)' if vps > 1 else ''}, mode {mode}{' (render-only: no checkpoints kept for a backward)' if render_only else ''}",
                   "mode": mode, "parallelism": f"dp{world}" + (" (data-parallel step rehearsed on a 1-rank group)" if dp_single else ""),
                   "dp_exchange": args.dp_exchange if (world > 1 or dp_single) and mode == "train" else None,
                   "dp_impl": args.dp_impl if (world > 1 or dp_single) and mode == "train" else None,
                   "rccl_ranks": n_ranks, "backend": args.backend if world > 1 else None,
                   "view_assignment": ("rank r renders view order[(step * world + r) mod views]; order = the views sorted by traversed block-entries, "
                                       "zigzag, so that the views of one step cost about the same (trainer.balanced_view_order); parameters replicated"
                                       if balanced else "rank r renders view (step * world + r) mod views; parameters replicated"),
                   "views_per_rank": vps, "views_per_step": world * vps,
                   "N": N, "W": W, "H": H, "tile": ts},
        "fwd_mpix_per_s": round(P / (fwd_ms * 1e-3) / 1e6, 2), "fwd_ms": round(fwd_ms, 4),
        "roofline": roof, "cpu_baseline": cpu, "stages": stages,
        "workload_stats": {"N_visible": st["N_visible"], "M_pairs": M, "M_eff_pairs_traversed": M_eff,
                           "max_tile_list": st["max_tile_list"], "mean_tile_list": round(M / T, 1),
                           "mean_nContrib": round(mean_contrib, 1), "checkpoints_per_forward": S_fwd},
        "workspace": {"bytes": int(r.lib.gs_workspace_bytes(r.ctx)), "capN": st["capN"], "capM": st["capM"],
                      "overflow": int(st["overflow"]), "overflow_recoveries": overflow_recoveries},
        "step_ms_spread": step_spread, "densify": densify_info, "depth_cuts": cut_info,
        "replicas_identical": replicas_identical, "loss": loss, "exchange": exchange,
        "pre_visits_per_view": pre_visits,
    }
    out["accounting_violations"] = sanitize_fractions(out)          # (fractions above 1 / rates above the HBM peak: nulled, listed)
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(out) + "\n").encode())


def roofline_block(dom, dom_ms, dom_src, surv, des, config, mode, ts, M_eff, pix_per_unit, fast16):
    """The `roofline` object of the result line, for the dominant stage `dom` (a pure function of its arguments and the
    committed profiles, so that tests/test_bench_launcher_cpu.py can hold it to "every fraction follows and none exceeds 1").
    achieved = SURVEY 8(d)'s bytes of the stage as it runs (survey_bytes) / its average launch time; frac against the 8 TB/s
    HBM peak; traffic = counter bytes per launch of the kernel from a PMC summary of THIS config, mode and kernel sources."""
    dom_bytes, dom_note = surv[dom]
    by = "survey"
    if dom_bytes is None:          # (no stage that can dominate has none today; should one, its designed bytes stand in, said so)
        dom_bytes, by = des[dom], "designed (the survey's formula does not describe this stage: " + str(dom_note) + ")"
    if dom_bytes is None:          # (no byte model at all: nothing to price -- the block says so instead of the run dying)
        dom_bytes, by = 0, "none: neither formula describes this stage (" + str(dom_note) + ")"
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # (a summary counts for the tile size it was taken at: tools/profile_round.sh writes the bench line's tile into its files)
    # (the generic blend kernels -- GSPLAT_BLOCK_LISTS=0 at a tile size that is not a multiple of 16 -- are other kernels than the
    # ones the summaries of that tile size were taken on)
    traffic, traffic_source = pmc_traffic_bytes(dom, config, mode, ts) if fast16 else (None, "the generic blend kernels: no PMC summary")
    if traffic is None and ts != 16 and not traffic_source:
        traffic_source = "no PMC summary for this tile size"
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
            "traffic_over_algorithmic": round(traffic / dom_bytes, 3) if traffic and dom_bytes else None,
            "algorithmic_bytes": int(dom_bytes), "algorithmic_bytes_are": by, "algorithmic_bytes_note": dom_note,
            "designed_bytes": int(des[dom]) if des[dom] is not None else None, "avg_launch_ms": round(dom_ms, 4), "avg_launch_ms_source": dom_src}
    # Round 6 (the verdict's item 5): where the counters saw the kernel move clearly FEWER bytes than the formula counts (traffic
    # below 0.95 of the algorithmic bytes), a fraction built on the formula flatters the kernel: the fraction built on the counter
    # bytes stands beside it, and `frac_claimed` names the one that is the claim -- the smaller, i.e. the counters'.  (Traffic
    # ABOVE the algorithmic bytes is wasted re-reads: `frac` on the algorithmic bytes stays the claim.)
    if traffic and dom_bytes and dom_ms > 0:
        roof["frac_by_counters"] = round(traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        roof["frac_claimed"] = "frac_by_counters" if traffic / dom_bytes < 0.95 else "frac"
        roof["frac_claimed_note"] = ("frac_by_counters = PMC traffic per launch / avg_launch_ms / peak; it is the claim where the counters "
                                     "saw fewer bytes than the formula counts (traffic_over_algorithmic < 0.95)")
    flop_per_pair = {"blend_fwd": 24.0, "blend_bwd": 70.0}
    if dom in flop_per_pair and not fast16:
        # tiles larger than a block: the scan / cull / compact kernels blend a small share of (pixel, list entry) pairs, a
        # count of all of them is no measure of anything (it came to 4.5x the vector peak for 200x200 tiles)
        roof["algorithmic_flop_frac"] = None
        roof["algorithmic_flop_note"] = "not defined for tiles that are not 16x16: the model counts every pixel of a tile against every list entry"
    elif dom in flop_per_pair:
        # NOT a hardware utilisation: SURVEY 8(d)'s flop count of the REFERENCE arithmetic per pixel-splat (24 forward, 70
        # backward) x the pixel-splats of the M_eff traversed block-splats (dead pixels and entries the staging cull drops
        # included) over the kernel's time, against the f32 vector peak.  What the hardware did is in "counters" below.
        tf = flop_per_pair[dom] * pix_per_unit * M_eff / (dom_ms * 1e-3) / 1e12
        roof["algorithmic_tflops"] = round(tf, 2)
        roof["algorithmic_flop_frac"] = round(tf / VALU_PEAK_TFLOPS, 4)
        roof["algorithmic_flop_note"] = ("SURVEY 8(d) flop per pixel-splat of the reference arithmetic x traversed pixel-splats / time / "
                                         "157.3 TFLOP/s; a model figure, not a counter")
        roof["counters"] = sq_counters(dom, config, mode, M_eff * pix_per_unit, ts)
        if roof["counters"]:
            roof["issue_model_frac"] = roof["counters"].get("issue_model_frac")
    return roof


def pmc_traffic_bytes(stage, config, mode, tile=16):
    """HBM-side bytes per launch of the stage's dominant kernel from a committed rocprofv3 PMC summary
    (profiles/*hbm_traffic_pmc.json: FETCH_SIZE and WRITE_SIZE in separate passes of this same bench; FETCH_SIZE is
    doubled as the MI355X guide prescribes for gfx950).  Only a summary taken on THIS config and mode and on the
    kernel sources of this build counts; anything else is reported as (None, why)."""
    import glob
    key = KERNEL_OF_STAGE.get(stage)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic_pmc.json")), key=os.path.getmtime)
    if not key or not files:
        return None, None
    sha = csrc_sha()
    why = None
    for f in reversed(files):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if j.get("config") != config or j.get("mode") != mode or j.get("tile", 16) != tile:
            continue
        if j.get("csrc_sha") != sha:
            why = why or f"stale: {os.path.basename(f)} was taken on other kernel sources ({j.get('csrc_sha')} != {sha})"
            continue
        for name, v in j["kernels"].items():
            if key in name and "FETCH_SIZE_KB_per_launch" in v and "WRITE_SIZE_KB_per_launch" in v:
                return (int((2.0 * v["FETCH_SIZE_KB_per_launch"] + v["WRITE_SIZE_KB_per_launch"]) * 1024),
                        {"file": "profiles/" + os.path.basename(f), "commit": j.get("commit"), "csrc_sha": sha})
        return None, f"{os.path.basename(f)} (these kernel sources) holds no FETCH_SIZE / WRITE_SIZE pair for {key}"
    return None, why


SIMDS = 1024                   # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9               # MI355X engine clock (tools/microbench prints it; the SQ cycle counters agree with duration x 2.4 GHz)
NOMINAL_VALU_CYCLES = 2.0      # the guide's nominal issue interval of a wave64 VALU instruction on one SIMD


def isa_mix_cost(kernel_name):
    """Average issue cost (cycles per wave64 VALU instruction per SIMD) of a blend kernel's inner loop by its instruction MIX:
    profiles/*blend_isa_mix.json (tools/isa_mix.py: static class counts of the shipped code object x the measured class
    costs of profiles/r01c_microbench_issue_rates.txt), only from a file taken on these kernel sources."""
    import glob
    want = kernel_name.replace("void ", "").replace("gs::", "").replace(" ", "")
    sha = csrc_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*blend_isa_mix.json")), key=os.path.getmtime, reverse=True):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if j.get("csrc_sha") != sha:
            continue
        for name, v in j.get("kernels", {}).items():
            if name.replace(" ", "") == want and v.get("inner_loop"):
                il = v["inner_loop"]
                return il["mix_cycles_per_valu_inst"], {"file": "profiles/" + os.path.basename(f), "valu_by_class": il["valu_by_class"],
                                                       "class_cost_cycles": j.get("class_cost_cycles")}
    return None, None


def sq_counters(stage, config, mode, pixel_splats, tile=16):
    """What the SQ counters of a committed rocprofv3 summary (profiles/*sq_counters.json, same config / mode / kernel sources
    rule as pmc_traffic_bytes) say about the stage's dominant kernel: VALU wave-instructions executed per launch and per
    traversed pixel-splat (x64 = lane-instructions), and how much of the chip's VALU ISSUE capacity over the kernel's span
    they account for,
        issue_frac = SQ_INSTS_VALU / 1024 SIMDs x cost / (average duration x 2.4 GHz),
    twice: issue_nominal_frac with the guide's nominal 2 cycles per instruction, issue_model_frac with the kernel's own mix
    cost (isa_mix_cost: half-rate min / max / cmp / cndmask, quarter-rate exp / rcp, packed, DPP and lane-swap instructions
    cost more than 2).  Both are <= 1 by construction when the costs are true issue intervals.  (Rounds 2-4 printed
    valu_issue_busy = 4 SQ_ACTIVE_INST_VALU / SIMDs / busy cycles, which adds up per WAVE and read 1.1 - 1.3 for the forward:
    gone.)"""
    import glob
    key = KERNEL_OF_STAGE.get(stage)
    sha = csrc_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*sq_counters.json")), key=os.path.getmtime, reverse=True):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if j.get("config") != config or j.get("mode") != mode or j.get("csrc_sha") != sha or j.get("tile", 16) != tile:
            continue
        for name, v in j["kernels"].items():
            if key in name and v.get("SQ_INSTS_VALU"):
                insts = v["SQ_INSTS_VALU"]
                out = {"kernel": name.replace("void ", ""), "valu_wave_insts_per_launch": insts,
                       "valu_lane_insts_per_pixel_splat": round(insts * 64.0 / max(pixel_splats, 1.0), 2),
                       # (SQ_ACTIVE_INST_VALU counts quad-cycles summed over waves: a per-wave average, not an occupancy)
                       "cycles_per_valu_wave_inst_per_wave": round(4.0 * v.get("SQ_ACTIVE_INST_VALU", 0.0) / insts, 3)}
                dur_ns = v.get("avg_duration_ns")
                if dur_ns:
                    cycles = dur_ns * 1e-9 * CLOCK_HZ
                    per_simd = insts / SIMDS
                    mix, mix_src = isa_mix_cost(name)
                    # the model's cycles over the span: a RATIO -- the class costs are measured issue intervals of isolated
                    # instruction streams, good to a few per cent, so a kernel at its issue bound can come out a shade above 1
                    # (grown scene: 1.0045).  issue_model_frac is that ratio capped at 1, said so where it was.
                    ratio = per_simd * mix / cycles if mix else None
                    out.update({"avg_duration_us_of_the_profile_run": round(dur_ns / 1e3, 2),
                                "span_cycles": round(cycles), "valu_wave_insts_per_simd": round(per_simd, 1),
                                "issue_nominal_frac": round(per_simd * NOMINAL_VALU_CYCLES / cycles, 4),
                                "mix_cycles_per_valu_inst": mix,
                                "issue_model_cycles_over_span": round(ratio, 4) if ratio is not None else None,
                                "issue_model_frac": round(min(ratio, 1.0), 4) if ratio is not None else None,
                                "isa_mix": mix_src})
                    out["passes"] = ("SQ_INSTS_VALU: the counter pass of tools/profile_round.sh (6 steps, kernels serialised); "
                                     "avg_duration_ns: its --kernel-trace --stats pass (30 steps) of the same command -- two runs of one build")
                    if ratio is not None and ratio > 1.0:
                        out["issue_model_note"] = (f"the class-cost model prices the kernel's VALU instructions at {ratio:.4f} of its span: "
                                                   "at the bound within the model's accuracy (a few per cent); the fraction is capped at 1")
                out["source"] = {"file": "profiles/" + os.path.basename(f), "commit": j.get("commit"), "csrc_sha": sha}
                return out
    return None


def physical_cores(threads):
    """Distinct (package, core) pairs among the CPUs this process may run on (SMT siblings count once)."""
    seen = set()
    try:
        for cpu in threads:
            base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
    except OSError:
        return None
    return len(seen) or None


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(mode, params, cams, W, H, targets, budget_s=12.0):
    """The CPU oracle timed on the host cores on the same workload and the same mode, one view after the other until
    about budget_s seconds of CPU work are spent (at least one view, at most all of them).  It is a port of the reference
    arithmetic (the reference itself is Swift + MLX + Metal and cannot run off Apple hardware), written to be checked
    against, not to be fast: OpenMP over Gaussians / tiles / pixels, but a single-threaded stable sort and a serial
    per-pair reduction in the blend backward.  Built here with -O3 -march=native as BASELINE.md section 3 says (the -O2
    build stays the parity oracle)."""
    import numpy as np
    from oracle import oracle as orc
    affinity = sorted(os.sched_getaffinity(0))
    threads = len(affinity)
    cores = physical_cores(affinity) or threads
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    o = orc.Oracle(np.float32, native=True)
    z = np.zeros(W * H, np.float32)
    t_fwd = t_rest = 0.0
    views = 0
    t_begin = time.perf_counter()
    for v, cam in enumerate(cams):
        c = cam.as_dict()
        t0 = time.perf_counter()
        fw = o.render_forward(params, c, W, H, 16, 16, 4)
        t1 = time.perf_counter()
        if mode != "forward":
            if mode == "train":
                _, cot, _, _, _ = o.loss_forward_backward(fw["color"].reshape(H, W, 3), targets[v], 0.2)
            else:
                cot = np.full((H, W, 3), 1e-3, np.float32)
            o.render_backward(params, c, W, H, 16, 16, 4, fw, cot.reshape(-1, 3), z, z)
        t2 = time.perf_counter()
        t_fwd += t1 - t0
        t_rest += t2 - t1
        views += 1
        if time.perf_counter() - t_begin >= budget_s:
            break
    what = {"forward": "forward", "fwdbwd": "forward + backward", "train": "forward + loss + backward, no optimizer step"}[mode]
    base = {"cores": cores, "threads": int(os.environ.get("OMP_NUM_THREADS", threads)), "kind": "port", "cpu": cpu_model(), "build": "gcc -O3 -march=native -fopenmp -ffp-contract=off",
            "sample": f"{views} view(s) of the same workload, {what} (forward {t_fwd:.2f} s, rest {t_rest:.2f} s in total)",
            "fwd_mpix_per_s": round(views * W * H / t_fwd / 1e6, 3)}
    if mode == "forward":
        return dict(base, value=round(views * W * H / t_fwd / 1e6, 4), unit="Mpix/s")
    return dict(base, value=round(views / (t_fwd + t_rest), 4), unit="views/s")


if __name__ == "__main__":
    main()
