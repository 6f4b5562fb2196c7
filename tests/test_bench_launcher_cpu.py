"""bench.py's host logic that needs no GPU: the self-launcher for --gpus N (SURVEY 8(e) contract), the config -> mode
map, the byte formulas, and the provenance rule for roofline.traffic."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launcher_command_is_the_drivers():
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "7", "--warmup", "2"], port=29777)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert bench.launcher_command(2, [])[cmd.index("--master-port") + 1].isdigit()       # a free port by default


def test_config_selects_the_mode_baseline_names():
    assert bench.parse_args([]).mode == "train" and bench.parse_args([]).config == "c3_300k_800"
    assert bench.parse_args(["--config", "c1_10k_400"]).mode == "forward"
    assert bench.parse_args(["--config", "c2_100k_800"]).mode == "fwdbwd"
    assert bench.parse_args(["--config", "c2_100k_800", "--mode", "train"]).mode == "train"
    assert bench.parse_args(["--config", "c5_garden_2m"]).mode == "train"
    a = bench.parse_args([])
    assert (a.gpus, a.steps, a.warmup, a.views) == (1, 50, 10, 100)          # 100 train views


def test_gpus_beyond_the_box_are_refused_before_any_rank_starts():
    """Round 3's last gpurun call ended in "the 2-rank run failed with exit code 1" and nothing else: two RCCL ranks had been
    asked of a one-GPU box.  Now the parent counts the visible GPUs (without initialising one) and says so itself."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("GSPLAT_BENCH_DEVICE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 2
    assert '{"metric"' not in p.stdout
    assert "needs 2 visible GPUs" in p.stderr and "this box has 0" in p.stderr and "GSPLAT_BENCH_DEVICE" in p.stderr
    # ... and a rank started by somebody else's launcher on such a box says the same instead of dying in set_device
    env.update(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert p.returncode != 0 and ("needs a GPU" in p.stderr + p.stdout or "visible GPUs" in p.stderr + p.stdout)


def test_self_launch_starts_the_ranks_and_relays_why_they_failed():
    """The one-card rehearsal (GSPLAT_BENCH_DEVICE) skips the GPU count.  Here there is no GPU at all: the parent must start
    N rank processes through torch.distributed.run (never touching a GPU itself), the ranks refuse to run on the CPU, and the
    parent reports the failure with a non-zero exit code, no result line, and the tail of the ranks' stderr."""
    env = dict(os.environ, GSPLAT_BENCH_DEVICE="0")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--backend", "gloo", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode != 0
    assert '{"metric"' not in p.stdout
    assert "2-rank run failed" in p.stderr and "lines of the ranks' stderr" in p.stderr
    relayed = p.stderr[p.stderr.index("lines of the ranks' stderr"):]
    assert "needs a GPU" in relayed          # the reason is IN the relayed tail, not only somewhere above it


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert p.returncode != 0 and "3 ranks" in (p.stderr + p.stdout)


def test_byte_formulas():
    N, K, M, P, T = 300_000, 25, 2_000_000, 640_000, 2500
    alg = bench.algorithmic_bytes(N, K, M, P, T)
    assert alg["proj_fwd"] == N * 408 and alg["proj_bwd"] == N * 728 and alg["adam"] == N * 2408
    assert alg["blend_fwd"] == M * 48 + P * 24 and alg["blend_bwd"] == M * 136 + P * 44 + N * 44
    des = bench.designed_bytes(N, K, 7_500_000, 1_500_000, P, T, 20_000, True)
    assert des["bin"] == N * 96 + 7_500_000 * 18 + T * 8 and des["adam"] == 0
    assert des["blend_fwd"] == 1_500_000 * 52 + P * 28 + 20_000 * 4096


def test_traffic_needs_a_profile_of_this_config_mode_and_sources(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_sha", lambda: "abc")
    kern = {"void gs::blend_bwd_v2_kernel<64, false>": {"FETCH_SIZE_KB_per_launch": 100.0, "WRITE_SIZE_KB_per_launch": 50.0}}
    (prof / "r09_hbm_traffic_pmc.json").write_text(json.dumps(
        {"config": "c3_300k_800", "mode": "train", "csrc_sha": "abc", "commit": "deadbee", "kernels": kern}))
    t, src = bench.pmc_traffic_bytes("blend_bwd", "c3_300k_800", "train")
    assert t == int((2 * 100.0 + 50.0) * 1024) and src["file"] == "profiles/r09_hbm_traffic_pmc.json" and src["commit"] == "deadbee"
    assert bench.pmc_traffic_bytes("blend_bwd", "c2_100k_800", "fwdbwd") == (None, None)       # other workload: nothing
    monkeypatch.setattr(bench, "csrc_sha", lambda: "other")
    t, why = bench.pmc_traffic_bytes("blend_bwd", "c3_300k_800", "train")                      # kernels changed since
    assert t is None and "stale" in why


def test_round3_flags_and_counter_block(tmp_path, monkeypatch):
    a = bench.parse_args(["--tile", "200", "--dp-impl", "native"])
    assert a.tile == 200 and a.dp_impl == "native"
    assert bench.parse_args([]).tile == 16 and bench.parse_args([]).dp_impl in ("torch", "native")
    # what the SQ counters say, from a summary of THIS config / mode / kernel sources only
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_sha", lambda: "abc")
    kern = {"void gs::blend_bwd_v2_kernel<64, false>": {"SQ_INSTS_VALU": 150e6, "SQ_ACTIVE_INST_VALU": 160e6, "valu_issue_busy": 0.9}}
    (prof / "r09_sq_counters.json").write_text(json.dumps(
        {"config": "c3_300k_800", "mode": "train", "csrc_sha": "abc", "commit": "deadbee", "kernels": kern}))
    c = bench.sq_counters("blend_bwd", "c3_300k_800", "train", 4.0e8)
    assert c["valu_wave_insts_per_launch"] == 150e6 and c["valu_issue_busy"] == 0.9
    assert c["valu_lane_insts_per_pixel_splat"] == round(150e6 * 64 / 4.0e8, 2)
    assert abs(c["cycles_per_valu_wave_inst"] - 4.0 * 160 / 150) < 1e-3           # SQ_ACTIVE_INST_VALU counts quad-cycles
    assert bench.sq_counters("blend_bwd", "c2_100k_800", "fwdbwd", 1.0) is None
    monkeypatch.setattr(bench, "csrc_sha", lambda: "other")
    assert bench.sq_counters("blend_bwd", "c3_300k_800", "train", 1.0) is None


def test_physical_cores_counts_smt_siblings_once():
    threads = sorted(os.sched_getaffinity(0))
    n = bench.physical_cores(threads)
    assert n is None or 1 <= n <= len(threads)
