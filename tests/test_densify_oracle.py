"""Known-answer tests for the oracle's restatement of the densify / prune row.

The reference has no test for this row (parity unpinned); these pin the restatement to the kernel text of
Trainer/GaussianTrainer.swift:317-427 and the host sequence :766-907 on hand-worked cases."""
import numpy as np
import pytest

from oracle.oracle import Oracle


@pytest.fixture(scope="module")
def o():
    return Oracle(np.float32)


def _logit(p):
    return float(np.log(p / (1 - p)))


def test_accum_grad_norm(o):
    g = np.array([[3, 4, 0], [0, 0, 0], [1, 2, 2]], np.float32)
    np.testing.assert_array_equal(o.accum_grad_norm(g), np.array([5, 0, 3], np.float32))
    np.testing.assert_array_equal(o.accum_grad_norm(g, np.array([1, 2, 3], np.float32)), np.array([6, 2, 6], np.float32))


def test_classify_decision_table(o):
    big, small = np.log(0.02), np.log(0.005)            # max exp(scale) above / below maxScale = 0.01
    scales = np.array([[small, small, small],           # 0: low gradient                  -> keep
                       [small, big, small],             # 1: high gradient, large          -> split
                       [small, small, small],           # 2: high gradient, small          -> clone
                       [big, big, big],                 # 3: transparent (wins over split) -> prune
                       [big, big, big]], np.float32)    # 4: high gradient but denom path  -> see below
    opacity = np.array([_logit(0.5), _logit(0.5), _logit(0.5), _logit(0.004), _logit(0.5)], np.float32)
    accum = np.array([0.0001, 0.01, 0.01, 0.01, 0.01], np.float32) * 10
    a, c = o.classify_gaussians(accum, 10.0, scales, opacity)
    assert a.tolist() == [0, 1, 2, 3, 1] and c.tolist() == [1, 2, 2, 0, 2]
    a, c = o.classify_gaussians(accum, 10.0, scales, opacity, allowDensify=False)      # over budget: prune only
    assert a.tolist() == [0, 0, 0, 3, 0] and c.tolist() == [1, 1, 1, 0, 1]
    a, c = o.classify_gaussians(accum, 0.0, scales, opacity)                           # denom 0 -> avg grad 0
    assert a.tolist() == [0, 0, 0, 3, 0]
    # thresholds are strict: avg == threshold keeps, opacity == threshold keeps
    a, _ = o.classify_gaussians(np.array([0.0002], np.float32), 1.0, scales[1:2], opacity[1:2])
    assert a.tolist() == [0]


def test_offsets_and_map(o):
    actions = np.array([0, 1, 3, 2, 0, 3, 1], np.int32)
    counts = np.array([1, 2, 0, 2, 1, 0, 2], np.int32)
    off, st = o.densify_offsets(actions, counts)
    assert off.tolist() == [0, 1, 3, 3, 5, 6, 6]
    assert st == dict(total=8, keep=2, split=2, clone=1, prune=2)
    g, m = o.build_densify_output_map(actions, off, st["total"])
    assert g.tolist() == [0, 1, 1, 3, 3, 4, 6, 6]
    assert m.tolist() == [0, 1, 2, 0, 3, 0, 1, 2]


def test_gather_and_noise(o):
    rng = np.random.default_rng(3)
    N, K = 4, 4
    p = dict(xyz=rng.normal(size=(N, 3)), features_dc=rng.normal(size=(N, 1, 3)),
             features_rest=rng.normal(size=(N, K - 1, 3)), scales=rng.normal(-4, 0.3, (N, 3)),
             rotation=rng.normal(size=(N, 4)), opacity=rng.normal(size=N))
    p = {k: v.astype(np.float32) for k, v in p.items()}
    g = np.array([0, 1, 1, 3, 3], np.int32)             # keep 0, split 1, (2 pruned), clone 3
    m = np.array([0, 1, 2, 0, 3], np.int32)
    nz = rng.normal(size=(5, 3)).astype(np.float32)
    out = o.densify_gather(p, g, m, nz)
    for k in ("features_dc", "features_rest", "rotation", "opacity"):
        np.testing.assert_array_equal(out[k], p[k][g])
    red = np.float32(-np.log(1.6))
    np.testing.assert_array_equal(out["scales"][[0, 3, 4]], p["scales"][[0, 3, 3]])
    np.testing.assert_array_equal(out["scales"][[1, 2]], p["scales"][[1, 1]] + red)
    mean1 = np.exp(p["scales"][1].astype(np.float64)).mean()
    np.testing.assert_array_equal(out["xyz"][[0, 3]], p["xyz"][[0, 3]])                # untouched slots
    np.testing.assert_allclose(out["xyz"][1], p["xyz"][1] + mean1 * 0.1 * nz[1], rtol=2e-6)
    np.testing.assert_allclose(out["xyz"][2], p["xyz"][1] - mean1 * 0.1 * nz[2], rtol=2e-6)
    np.testing.assert_allclose(out["xyz"][4], p["xyz"][3] + 0.01 * nz[4], rtol=2e-6)
    # children of one split sit mirrored about the parent only when fed mirrored noise: the reference draws
    # independent noise per slot and flips the sign of the second (:872-881)
    out2 = o.densify_gather(p, g, m, None)              # prune-only branch: pure gather
    for k in p:
        np.testing.assert_array_equal(out2[k], p[k][g])


def test_split_and_prune_sequence(o):
    rng = np.random.default_rng(0)
    N = 2000
    p = dict(xyz=rng.normal(size=(N, 3)), features_dc=rng.normal(size=(N, 1, 3)),
             features_rest=rng.normal(size=(N, 24, 3)), scales=rng.normal(np.log(0.01), 0.5, (N, 3)),
             rotation=rng.normal(size=(N, 4)), opacity=rng.normal(-3, 3, N))
    acc = np.abs(rng.normal(0, 3e-4, N)) * 5
    new, st = o.split_and_prune(p, acc, 5.0, lambda t: rng.normal(size=(t, 3)))
    assert st["keep"] + st["split"] + st["clone"] + st["prune"] == N
    assert st["total"] == st["keep"] + 2 * st["split"] + 2 * st["clone"] == new["xyz"].shape[0]
    assert min(st.values()) > 0
    # nothing to do -> None
    none, st0 = o.split_and_prune(p, np.zeros(N), 5.0, None, minOpacity=0.0)
    assert none is None and st0["total"] == N and st0["keep"] == N
