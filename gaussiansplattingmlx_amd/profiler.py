"""Host-side mirror of the reference's `IntervalProfiler` (Trainer/GaussianTrainer.swift:122-243): named, nestable
wall-clock sections with self / total time and a top-K report in the reference's format.

On this path everything the host does is queue launches, so a wall-clock section alone would time the enqueue, not the
work.  A profiled iteration therefore also records the library's per-stage HIP events (gs_profile_enable / _read) and
`deviceSections()` reports them under the reference's section names:

    train.forward                  projection + binning + blend forward        (GaussianTrainer.swift:669)
    train.loss.ssim                the fused L1 + DSSIM loss kernel            (:689-714)
    bwd.globalTileComposite        blend backward                              (GaussianRenderer.swift:157-172)
    bwd.projectionScreenFused      projection backward (+ the fused Adam step) (GaussianRenderer.swift:579-600)
    train.optimizer.applySingle    Adam, when it runs as its own kernel        (GaussianTrainer.swift:1069)
"""
from __future__ import annotations

import time

DEVICE_SECTIONS = {
    "train.forward": ("proj_fwd", "bin", "blend_fwd"),
    "train.loss.ssim": ("loss",),
    "bwd.globalTileComposite": ("blend_bwd",),
    "bwd.projectionScreenFused": ("proj_bwd",),
    "train.optimizer.applySingle": ("adam",),
}


class IntervalProfiler:
    class Metric:
        __slots__ = ("totalNanoseconds", "selfNanoseconds", "count")

        def __init__(self):
            self.totalNanoseconds = self.selfNanoseconds = self.count = 0

    def __init__(self, enabled: bool):
        self.enabled = bool(enabled)
        self.metrics: dict[str, IntervalProfiler.Metric] = {}
        self._stack: list[list[int]] = []            # [start ns, child ns]
        self.deviceMs: dict[str, tuple[float, int]] = {}

    def measure(self, name: str, body):
        if not self.enabled:
            return body()
        self._stack.append([time.perf_counter_ns(), 0])
        value = body()
        end = time.perf_counter_ns()
        start, child = self._stack.pop()
        elapsed = max(end - start, 0)
        m = self.metrics.setdefault(name, IntervalProfiler.Metric())
        m.totalNanoseconds += elapsed
        m.selfNanoseconds += max(elapsed - child, 0)
        m.count += 1
        if self._stack:
            self._stack[-1][1] += elapsed
        return value

    def setDeviceStages(self, stage_ms: dict):
        """stage_ms: renderer.profileRead() -- {stage: (ms, calls)}; kept under the reference's section names."""
        self.deviceMs = {}
        for section, stages in DEVICE_SECTIONS.items():
            ms = sum(stage_ms[s][0] for s in stages if s in stage_ms)
            calls = max((stage_ms[s][1] for s in stages if s in stage_ms), default=0)
            if calls:
                self.deviceMs[section] = (ms, calls)

    def deviceSections(self) -> dict:
        return dict(self.deviceMs)

    def makeReport(self, iteration: int, iterationNanoseconds: int, topK: int = 12, minMilliseconds: float = 0.0) -> str:
        if not self.enabled:
            return ""
        iterationMs = iterationNanoseconds / 1e6
        totalSelfNs = sum(m.selfNanoseconds for m in self.metrics.values())
        if totalSelfNs == 0:
            return f"[Profile] iter={iteration} wall={iterationMs:.3f} ms (no measured sections)"
        lines = [f"[Profile] iter={iteration} wall={iterationMs:.3f} ms (top {topK})"]
        shown = 0
        for name, m in sorted(self.metrics.items(), key=lambda kv: -kv[1].selfNanoseconds):
            if shown >= topK:
                break
            selfMs = m.selfNanoseconds / 1e6
            if selfMs < minMilliseconds:
                continue
            totalMs = m.totalNanoseconds / 1e6
            lines.append(f"  {name}: self {selfMs:.3f} ms, total {totalMs:.3f} ms "
                         f"({m.selfNanoseconds / totalSelfNs * 100.0:.1f}%, calls={m.count}, avg={totalMs / max(m.count, 1):.4f} ms)")
            shown += 1
        if shown == 0:
            lines.append("  (all sections are below threshold)")
        for name, (ms, calls) in self.deviceMs.items():
            lines.append(f"  [device] {name}: {ms:.3f} ms (calls={calls}, avg={ms / max(calls, 1):.4f} ms)")
        return "\n".join(lines)
