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
    prof.mkdir(exist_ok=True)
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


def test_survey_bytes_of_the_stages_as_they_run():
    """Round 4's verdict: c5's roofline named the FUSED projection backward + Adam and priced it with the projection backward's
    bytes alone (frac 0.199 for a kernel whose counter traffic is 0.59 of the peak), and two stage rates stood above the HBM peak
    (proj_fwd 8546 GB/s under colour riders, bin 7461 GB/s by the survey's 6-pass model).  survey_bytes is the one place that
    knows what a stage moves as it runs."""
    N, K, M, M_eff, P, T = 1_750_000, 25, 3_000_000, 1_300_000, 1237 * 822, 78 * 52
    E = N * (11 + 3 * K)
    alg = bench.algorithmic_bytes(N, K, M, P, T)
    plain = bench.survey_bytes(N, K, M, M_eff, P, T)
    assert plain["proj_bwd"][0] == alg["proj_bwd"] == N * 728 and plain["adam"][0] == alg["adam"] and plain["proj_fwd"][0] == N * 408
    fused = bench.survey_bytes(N, K, M, M_eff, P, T, fused_adam=True, colour_riders=True)
    # round 6: the parameters are read ONCE by the fused kernel (the survey's two formulas read them once each)
    assert fused["proj_bwd"][0] == alg["proj_bwd"] + alg["adam"] - 2 * E * 4 - N * (44 + 12 * K) == N * (728 + 2408 - 688 - 344)
    assert fused["adam"][0] is None and "fused" in fused["adam"][1]
    assert fused["proj_fwd"][0] is None and "rider" in fused["proj_fwd"][1]
    assert plain["bin"][0] is None and fused["bin"][0] is None and "6-pass" in plain["bin"][1]
    # round 6: a data-parallel step's projection backward / Adam are two kernels per stage name -- no per-launch bytes, said so;
    # and the projection's own launch under colour riders reads the geometry only (both used to print rates above the HBM peak
    # that rate_gbps dropped silently)
    dp = bench.survey_bytes(N, K, M, M_eff, P, T, dp_form=True, colour_riders=True)
    assert dp["proj_bwd"][0] is None and dp["adam"][0] is None and "two kernels" in dp["adam"][1] and dp["blend_bwd"] == plain["blend_bwd"]
    ddp = bench.designed_bytes(N, K, M, M_eff, P, T, 1000, False, colour_riders=True, dp_form=True)
    assert ddp["proj_bwd"] is None and ddp["adam"] is None and ddp["proj_fwd"] == N * (44 + 68)
    assert bench.designed_bytes(N, K, M, M_eff, P, T, 1000, True)["proj_fwd"] == N * (44 + 12 * K + 68)
    # blend: the formula on the traversed block-splats, not on everything binned
    assert plain["blend_bwd"][0] == M_eff * 136 + P * 44 + N * 44 and plain["blend_fwd"][0] == M_eff * 48 + P * 24
    # c5 as profiled in round 4 (profiles/r04_c5_*): 0.912 ms per launch -> the fraction the counters show, not 0.199
    frac = fused["proj_bwd"][0] / 0.912e-3 / 1e9 / bench.HBM_PEAK_GBS
    assert 0.45 < frac < 0.55
    # round 5's c5 line as the verdict recomputed it: 4.89 GB (N = 1.9975 M x 2448 B) in 0.7813 ms printed 0.78; its own PMC
    # traffic of 4.27 GB gives 0.68.  With the parameters read once the formula says 4.20 GB -> 0.67: formula and counters agree
    n5 = 1_997_500
    s5 = bench.survey_bytes(n5, K, M, M_eff, P, T, fused_adam=True)["proj_bwd"][0]
    assert abs(s5 / 0.7813e-3 / 1e9 / bench.HBM_PEAK_GBS - 0.67) < 0.01 and 0.95 < 4.27e9 / s5 < 1.05
    # a rate is printed only where there are bytes and time; one above the peak is returned raw so that sanitize_fractions
    # nulls it IN THE LINE and lists it (round 5 dropped it silently here)
    assert bench.rate_gbps(None, 0.1) is None and bench.rate_gbps(1e9, 0.0) is None
    assert bench.rate_gbps(131e6, 0.0153) == round(131e6 / 0.0153 / 1e6, 1) > bench.HBM_PEAK_GBS
    line = {"stages": {"proj_fwd": {"GBps_survey_bytes": bench.rate_gbps(131e6, 0.0153)}}}
    bad = bench.sanitize_fractions(line)
    assert line["stages"]["proj_fwd"]["GBps_survey_bytes"] is None and len(bad) == 1 and "8562" in bad[0]
    assert bench.rate_gbps(131e6, 0.0353) == round(131e6 / 0.0353 / 1e6, 1)


def _profiles(tmp_path, monkeypatch, kern_sq, mix=None, kern_pmc=None):
    prof = tmp_path / "profiles"
    prof.mkdir(exist_ok=True)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_sha", lambda: "abc")
    meta = {"config": "c3_300k_800", "mode": "train", "csrc_sha": "abc", "commit": "deadbee"}
    (prof / "r09_sq_counters.json").write_text(json.dumps(dict(meta, kernels=kern_sq)))
    if kern_pmc:
        (prof / "r09_hbm_traffic_pmc.json").write_text(json.dumps(dict(meta, kernels=kern_pmc)))
    if mix:
        (prof / "r09_blend_isa_mix.json").write_text(json.dumps({"csrc_sha": "abc", "class_cost_cycles": {"full": 2.4}, "kernels": mix}))


def test_round3_flags_and_counter_block(tmp_path, monkeypatch):
    a = bench.parse_args(["--tile", "200", "--dp-impl", "native"])
    assert a.tile == 200 and a.dp_impl == "native"
    assert bench.parse_args([]).tile == 16 and bench.parse_args([]).dp_impl in ("torch", "native")
    # what the SQ counters say, from a summary of THIS config / mode / kernel sources only
    name = "void gs::blend_bwd_v2_kernel<64, false>"
    _profiles(tmp_path, monkeypatch, {name: {"SQ_INSTS_VALU": 146.2e6, "SQ_ACTIVE_INST_VALU": 156.3e6, "avg_duration_ns": 270446.0}},
              mix={"blend_bwd_v2_kernel<64,false>": {"inner_loop": {"mix_cycles_per_valu_inst": 4.5, "valu_by_class": {"full": 30, "half": 20}}}})
    c = bench.sq_counters("blend_bwd", "c3_300k_800", "train", 4.0e8)
    assert c["valu_wave_insts_per_launch"] == 146.2e6 and "valu_issue_busy" not in c
    assert c["valu_lane_insts_per_pixel_splat"] == round(146.2e6 * 64 / 4.0e8, 2)
    assert abs(c["cycles_per_valu_wave_inst_per_wave"] - 4.0 * 156.3 / 146.2) < 1e-3           # SQ_ACTIVE_INST_VALU counts quad-cycles
    # the verdict's arithmetic: 146.2 M instructions on 1024 SIMDs in 270.4 us x 2.4 GHz = 649 k cycles -> 4.55 cycles each;
    # 0.44 of the nominal 2-cycle issue rate, 0.99 of what a 4.5-cycle mix allows
    cycles = 270446e-9 * 2.4e9
    assert abs(c["span_cycles"] - cycles) <= 1
    assert abs(c["issue_nominal_frac"] - 146.2e6 / 1024 * 2.0 / cycles) < 1e-4 and 0.43 < c["issue_nominal_frac"] < 0.45
    assert abs(c["issue_model_frac"] - 146.2e6 / 1024 * 4.5 / cycles) < 1e-4 and c["issue_model_frac"] <= 1.0
    assert c["isa_mix"]["file"] == "profiles/r09_blend_isa_mix.json"
    assert c["issue_model_cycles_over_span"] == c["issue_model_frac"] and "issue_model_note" not in c
    assert bench.sq_counters("blend_bwd", "c2_100k_800", "fwdbwd", 1.0) is None
    # a kernel at its issue bound can come out a shade above 1 by the class costs (grown scene: 1.0045): the RATIO is printed as
    # it is, the fraction is capped at 1 and says so
    _profiles(tmp_path, monkeypatch, {name: {"SQ_INSTS_VALU": 150.0e6, "SQ_ACTIVE_INST_VALU": 156.3e6, "avg_duration_ns": 270446.0}},
              mix={"blend_bwd_v2_kernel<64,false>": {"inner_loop": {"mix_cycles_per_valu_inst": 4.5, "valu_by_class": {"full": 30, "half": 20}}}})
    c = bench.sq_counters("blend_bwd", "c3_300k_800", "train", 4.0e8)
    assert c["issue_model_cycles_over_span"] > 1.0 and c["issue_model_frac"] == 1.0 and "capped" in c["issue_model_note"]
    assert bench.sanitize_fractions({"roofline": {"counters": c}}) == []
    monkeypatch.setattr(bench, "csrc_sha", lambda: "other")
    assert bench.sq_counters("blend_bwd", "c3_300k_800", "train", 1.0) is None


def test_every_emitted_fraction_follows_and_none_exceeds_one(tmp_path, monkeypatch):
    """The roofline block as bench.py builds it, on round 4's own c3 and c5 numbers (profiles/r04_*): every `frac` can be
    recomputed from the block's own fields, none exceeds 1, and sanitize_fractions nulls and lists anything that would."""
    nb, nf, npb = "void gs::blend_bwd_v2_kernel<64, false>", "void gs::blend_fwd_v2q_kernel<64, false>", "void gs::proj_bwd_fused_kernel<2>"
    _profiles(tmp_path, monkeypatch,
              {nb: {"SQ_INSTS_VALU": 146.2e6, "SQ_ACTIVE_INST_VALU": 156.3e6, "avg_duration_ns": 270446.0},
               nf: {"SQ_INSTS_VALU": 102.6e6, "SQ_ACTIVE_INST_VALU": 105.8e6, "avg_duration_ns": 173422.0}},
              mix={"blend_bwd_v2_kernel<64,false>": {"inner_loop": {"mix_cycles_per_valu_inst": 4.5, "valu_by_class": {}}},
                   "blend_fwd_v2q_kernel<64,false>": {"inner_loop": {"mix_cycles_per_valu_inst": 2.87, "valu_by_class": {}}}},
              kern_pmc={nb: {"FETCH_SIZE_KB_per_launch": 61.4e3 / 1.024, "WRITE_SIZE_KB_per_launch": 66.9e3 / 1.024},
                        npb: {"FETCH_SIZE_KB_per_launch": 1050e3 / 1.024, "WRITE_SIZE_KB_per_launch": 1976e3 / 1.024}})
    N, K, M, M_eff, P, T = 327_582, 25, 7_900_000, 1_385_815, 640_000, 2500
    surv = bench.survey_bytes(N, K, M, M_eff, P, T, fused_adam=True, colour_riders=True)
    des = bench.designed_bytes(N, K, M, M_eff, P, T, 20_600, True)
    for dom, ms in (("blend_bwd", 0.2704), ("blend_fwd", 0.1734)):
        roof = bench.roofline_block(dom, ms, "test", surv, des, "c3_300k_800", "train", 16, M_eff, 256.0, True)
        assert roof["algorithmic_bytes"] == surv[dom][0]
        assert abs(roof["achieved"] - roof["algorithmic_bytes"] / (roof["avg_launch_ms"] * 1e-3) / 1e9) < 0.5
        assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4 and roof["frac"] <= 1.0
        c = roof["counters"]
        assert c["issue_nominal_frac"] <= c["issue_model_frac"] <= 1.0 and roof["issue_model_frac"] == c["issue_model_frac"]
        assert roof["algorithmic_flop_frac"] <= 1.0
    bwd = bench.roofline_block("blend_bwd", 0.2704, "test", surv, des, "c3_300k_800", "train", 16, M_eff, 256.0, True)
    assert abs(bwd["frac"] - 0.107) < 2e-3 and abs(bwd["traffic_over_algorithmic"] - 0.83) < 0.02          # the verdict's figures
    # c5: the fused projection backward + Adam
    N5 = 1_745_000
    surv5 = bench.survey_bytes(N5, K, 3_000_000, 1_300_000, 1237 * 822, 4056, fused_adam=True)
    des5 = bench.designed_bytes(N5, K, 3_000_000, 1_300_000, 1237 * 822, 4056, 10_000, True)
    monkeypatch.setattr(bench, "pmc_traffic_bytes", lambda *a: (int(4.27e9), {"file": "x"}))
    r5 = bench.roofline_block("proj_bwd", 0.912, "test", surv5, des5, "c5_garden_2m", "train", 16, 1_300_000, 256.0, True)
    assert 0.47 <= r5["frac"] <= 0.53 and 1.1 <= r5["traffic_over_algorithmic"] <= 1.2 and "counters" not in r5
    # (round 4's c5 profile moved 4.27 GB at N = 1.745 M: 1.15x the round-6 formula -- traffic above the formula keeps `frac` the claim)
    assert r5["frac_claimed"] == "frac" and abs(r5["frac_by_counters"] - 4.27e9 / 0.912e-3 / 1e9 / 8000.0) < 1e-4
    # counters BELOW the formula (the blend backward: 0.83x): the counter fraction is the claim, and it is the smaller one
    assert bwd["frac_claimed"] == "frac_by_counters" and bwd["frac_by_counters"] < bwd["frac"]
    assert abs(bwd["frac_by_counters"] - bwd["traffic"] / 0.2704e-3 / 1e9 / 8000.0) < 1e-4
    # a stage neither formula prices (the data-parallel form's Adam; a several-ranks-on-one-card rehearsal once ranked it first and
    # the run died on None / float): the block says so and carries no rate
    none = bench.roofline_block("adam", 0.5, "test", {"adam": (None, "dp form")}, {"adam": None}, "c3_300k_800", "train", 16, M_eff, 256.0, True)
    assert none["achieved"] == 0.0 and none["frac"] == 0.0 and none["algorithmic_bytes"] == 0 and "none" in none["algorithmic_bytes_are"]
    # the net under it all
    line = {"roofline": {"frac": 0.5, "counters": {"issue_model_frac": 1.13}, "achieved": 9000.0, "unit": "GB/s"},
            "stages": {"proj_fwd": {"GBps_survey_bytes": 8546.0, "ms": 0.0153}, "bin": {"GBps_designed_bytes": 1045.0}}}
    bad = bench.sanitize_fractions(line)
    assert len(bad) == 3 and line["roofline"]["counters"]["issue_model_frac"] is None and line["stages"]["proj_fwd"]["GBps_survey_bytes"] is None
    assert line["roofline"]["achieved"] is None and line["roofline"]["frac"] == 0.5 and line["stages"]["bin"]["GBps_designed_bytes"] == 1045.0
    assert bench.sanitize_fractions(line) == []


def test_physical_cores_counts_smt_siblings_once():
    threads = sorted(os.sched_getaffinity(0))
    n = bench.physical_cores(threads)
    assert n is None or 1 <= n <= len(threads)


def test_visible_gpus_is_bounded_by_the_render_nodes_the_process_can_open(tmp_path):
    """Round 5's advisor: the kfd topology lists every GPU of the host; what this process can use is bounded by the render
    nodes it may open."""
    d = tmp_path / "dri"
    d.mkdir()
    for n in ("renderD128", "renderD129", "card0"):
        (d / n).write_text("")
    assert bench.usable_render_nodes(str(d)) == 2
    os.chmod(d / "renderD129", 0)
    if os.geteuid() != 0:          # (root may open anything)
        assert bench.usable_render_nodes(str(d)) == 1
    assert bench.usable_render_nodes(str(tmp_path / "none")) is None
