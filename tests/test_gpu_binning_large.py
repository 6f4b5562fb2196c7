"""Binning at the sizes BASELINE config C5 (2 M Gaussians, 1237x822) actually runs at, held to the oracle bit for bit.

With N >= 2^20 Gaussians and 12-13 tile bits a (tile, index) pair no longer fits one 32-bit word
(binning.hip launch_binning: idxBits + tileBits > 32), so these inputs take the TWO-WORD pair path
(expand_kernel with idxBits == 0, wide_scatter_kernel<true> or the key + value radix_sort and tile_ranges_kernel), the
separate block-prefix launches above GS_FUSED_SCAN_MAX = 2048 scan blocks (bigScan) and the 8-slice expansion of blocks
with >= 32768 positions -- none of which the small bit-exact cases of test_gpu_parity.py reach.  The rects are synthetic
(as in test_tile_bin_equal_depth_ties_and_empty), so the oracle's cost is one stable sort of M pairs.

Order contract: the reference's stable sort of (tile, depth bits) keys emitted in Gaussian order
(slang/gaussian_tile_global_kernels.slang:73-126, 151-305; GaussianRenderer.swift:333-490).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _renderer(W, H, tile=(16, 16)):
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on an MI355X box")
    from gaussiansplattingmlx_amd.renderer import GaussianRenderer
    return GaussianRenderer(4, W, H, tile, False)


def _np(t):
    return t.detach().cpu().numpy()


def synthetic_rects(seed, N, W, H, near=6000, near_span=220.0, far_span=14.0, p_invisible=0.15, p_tied=0.3):
    """Pixel rects, radii and depths for N Gaussians: most cover one to four tiles, the `near` ones with the smallest
    depths cover ~14x14 tiles each (consecutive in depth order, so whole 256-Gaussian scan blocks exceed the 32768
    positions at which the expansion slices them), 15 % are invisible (radius 0), a few per cent lie off screen (radius > 0
    but no tile: the one-sided clamps of kernels.slang:158-172 leave min > max), and 30 % share one of eight depth values
    (ties fall back to the Gaussian index)."""
    rng = np.random.default_rng(seed)
    cx, cy = rng.uniform(-20, W + 20, N), rng.uniform(-20, H + 20, N)
    half = rng.uniform(0.5, far_span, N)
    depths = rng.uniform(2.0, 9.0, N).astype(np.float32)
    tied = rng.uniform(size=N) < p_tied
    depths[tied] = rng.choice(np.array([2.5, 3.0, 3.25, 4.0, 5.5, 6.0, 7.75, 8.0], np.float32), int(tied.sum()))
    nearIdx = rng.choice(N, near, replace=False)
    depths[nearIdx] = rng.uniform(0.3, 0.9, near).astype(np.float32)
    half[nearIdx] = rng.uniform(0.4 * near_span, 0.6 * near_span, near)
    rectMin = np.stack([cx - half, cy - half], 1)
    rectMax = np.stack([cx + half, cy + half], 1)
    # the projection's one-sided clamps (min >= 0, max <= size - 1)
    rectMin = np.maximum(rectMin, 0.0).astype(np.float32)
    rectMax = np.minimum(rectMax, np.array([W - 1.0, H - 1.0])).astype(np.float32)
    radii = np.where(rng.uniform(size=N) < p_invisible, 0.0, 3.0).astype(np.float32)
    return rectMin, rectMax, radii, depths


def _assert_same_lists(info, bn, what=""):
    assert info["M"] == bn.M and info["maxTilePairs"] == bn.B, what
    np.testing.assert_array_equal(_np(info["tileCounts"]).astype(np.uint32), bn.tileCounts, err_msg=what)
    rng_ = _np(info["tileRanges"]).astype(np.uint32)
    nz = bn.tileCounts > 0
    np.testing.assert_array_equal(rng_[nz], bn.tileRanges[nz], err_msg=what)
    np.testing.assert_array_equal(_np(info["sortedGaussIdx"]).astype(np.uint32), bn.sortedIdx, err_msg=what)


class _CutLists:
    """The uncut oracle lists with a per-tile depth cut applied: every tile keeps the entries whose key does not
    exceed its cut -- a prefix, since the lists are in key order."""

    def __init__(self, bn, T, seed, p_cut=0.7):
        rng = np.random.default_rng(seed)
        cutKey = np.full(T, 0xFFFFFFFF, np.uint64)
        has = np.nonzero(bn.tileCounts > 0)[0]
        chosen = has[rng.uniform(size=has.size) < p_cut]
        pos = (rng.uniform(size=chosen.size) * bn.tileCounts[chosen]).astype(np.int64)
        cutKey[chosen] = bn.sortedLow[bn.tileRanges[chosen, 0].astype(np.int64) + pos]
        self.store = (0xFFFFFFFF - cutKey).astype(np.uint32)          # the hint buffer's encoding; 0 = no cut
        keep = bn.sortedLow.astype(np.uint64) <= cutKey[bn.sortedHigh]
        self.sortedIdx = bn.sortedIdx[keep]
        self.tileCounts = np.bincount(bn.sortedHigh[keep], minlength=T).astype(np.uint32)
        self.M = int(keep.sum())
        self.B = int(self.tileCounts.max()) if T else 0
        ends = np.cumsum(self.tileCounts, dtype=np.uint64)
        self.tileRanges = np.stack([ends - self.tileCounts, ends], 1).astype(np.uint32)
        self.nCut = int(chosen.size)


CASES = {
    # 1024x1024 = exactly 4096 tiles (12 bits) + 2^21 index bits = 33: two words; one-pass tile sort (wide = 1) or two
    # 8-bit passes (wide = 0); 4297 scan blocks > 2048: bigScan; near Gaussians: sliced blocks
    "1.1M_4096tiles": dict(W=1024, H=1024, N=1_100_000, wides=(1, 0)),
    # 4160 tiles (13 bits) + 20 index bits = 33: two words through the key + value radix_sort and tile_ranges_kernel
    "600k_4160tiles": dict(W=1040, H=1024, N=600_000, wides=(1,)),
    # the C5 shape itself: 2 M Gaussians, 1237x822 (78 x 52 = 4056 tiles, partial edge tiles)
    "c5_2M_1237x822": dict(W=1237, H=822, N=2_000_000, wides=(1,)),
}


@pytest.mark.parametrize("case", list(CASES))
def test_two_word_pair_paths_bit_exact(oracle32, case):
    cfg = CASES[case]
    W, H, N = cfg["W"], cfg["H"], cfg["N"]
    T = ((W + 15) // 16) * ((H + 15) // 16)
    rectMin, rectMax, radii, depths = synthetic_rects(17, N, W, H)
    bn = oracle32.tile_bin(rectMin, rectMax, radii, depths, W, H, 16, 16)
    # the input does what it is here for: enough Gaussians for two-word pairs and the separate prefix launches,
    # blocks heavy enough to be sliced, ties, invisible and off-screen Gaussians
    idxBits = int(np.ceil(np.log2(N)))
    tileBits = int(np.ceil(np.log2(T)))
    assert idxBits + tileBits > 32 and N // 256 > 2048
    assert ((radii > 0) & (bn.tilesTouched == 0)).sum() > 100
    order = np.lexsort((np.arange(N), depths.view(np.uint32)))
    perBlock = np.add.reduceat(bn.tilesTouched[order].astype(np.int64), np.arange(0, N, 256))
    assert (perBlock >= 32768).sum() >= 4, perBlock.max()
    r = _renderer(W, H)
    for wide in cfg["wides"]:
        r.setTuning(wide_tile_sort=wide)
        info = r.buildGlobalTileSliceInfo((rectMin, rectMax), radii, depths)
        _assert_same_lists(info, bn, f"{case} wide={wide}")
        # ... and under forced depth cuts (cut expansion + compaction with the two-level segment prefix)
        cut = _CutLists(bn, T, seed=5)
        assert cut.nCut > T // 2 and cut.M < bn.M
        info = r.buildGlobalTileSliceInfo((rectMin, rectMax), radii, depths, tileCuts=cut.store)
        _assert_same_lists(info, cut, f"{case} wide={wide} cut")
    r.close()


@pytest.mark.parametrize("wide", [1, 0])
def test_cut_binning_small_packed_words(oracle32, wide):
    """The same cut lists on the ONE-word path (N = 40 000: 16 + 10 bits), few scan blocks (every block sums the counts
    itself), sliced because there are fewer blocks than CUs."""
    W, H, N = 640, 360, 40_000
    T = ((W + 15) // 16) * ((H + 15) // 16)
    rectMin, rectMax, radii, depths = synthetic_rects(23, N, W, H, near=300, near_span=120.0)
    bn = oracle32.tile_bin(rectMin, rectMax, radii, depths, W, H, 16, 16)
    r = _renderer(W, H)
    r.setTuning(wide_tile_sort=wide)
    _assert_same_lists(r.buildGlobalTileSliceInfo((rectMin, rectMax), radii, depths), bn)
    for seed in (1, 2):
        cut = _CutLists(bn, T, seed=seed, p_cut=0.5 * seed)
        _assert_same_lists(r.buildGlobalTileSliceInfo((rectMin, rectMax), radii, depths, tileCuts=cut.store), cut)
    # a cut word of 0 everywhere is "no cut"
    _assert_same_lists(r.buildGlobalTileSliceInfo((rectMin, rectMax), radii, depths, tileCuts=np.zeros(T, np.uint32)), bn)
    r.close()


@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("tiles,slack", [(12, 100), (12, 4096 * 3 + 5), (26, 4096 + 1), (100, 4096 * 3), (9, 1)])
def test_one_pass_tile_sort_with_a_reserve_that_is_not_a_multiple_of_eight_sort_tiles(oracle32, tiles, slack, packed):
    """wide_scatter_kernel maps sort tile t to block 8 + 8 (t mod perXcd) + t / perXcd with perXcd = ceil(active / 8): with
    a pair reserve of `tiles` sort tiles (4096 pairs each, not a multiple of 8) and M within 7 tiles of it, the last
    tiles' blocks lie beyond nbAll + 8 -- the grid has to be rounded up to a multiple of 8 (round 2 launched nbAll + 8
    and would have left those pairs unsorted; no bench or test reserve hit it)."""
    W, H = (1024, 1024) if not packed else (640, 360)
    capM = tiles * 4096
    target = capM - slack
    N = 1_100_000 if not packed else target + 5000          # packed: 10 tile bits + at most 19 index bits
    rng = np.random.default_rng(tiles * 7 + slack)
    # single-tile rects for `target` visible Gaussians (M = target exactly), the rest invisible
    gx, gy = (W + 15) // 16, (H + 15) // 16
    tx, ty = rng.integers(0, gx, N), rng.integers(0, gy, N)
    rectMin = np.stack([tx * 16 + 2.0, ty * 16 + 2.0], 1).astype(np.float32)
    rectMax = np.minimum(rectMin + 5.0, np.array([W - 1.0, H - 1.0], np.float32)).astype(np.float32)
    radii = np.zeros(N, np.float32)
    radii[rng.choice(N, target, replace=False)] = 2.0
    depths = rng.choice(np.linspace(1.0, 4.0, 97).astype(np.float32), N)
    bn = oracle32.tile_bin(rectMin, rectMax, radii, depths, W, H, 16, 16)
    assert bn.M == target and (target + 4095) // 4096 > 8 * (tiles // 8) and tiles % 8 != 0
    r = _renderer(W, H)
    r.reserve(N, capM)
    assert r.stats()["capM"] == capM
    r.setTuning(wide_tile_sort=1)
    _assert_same_lists(r.buildGlobalTileSliceInfo((rectMin, rectMax), radii, depths), bn)
    r.close()


@pytest.mark.parametrize("N,ties", [(16385, 0.3), (40_000, 0.0), (322_000, 0.0), (322_000, 0.3), (600_000, 0.9), (655_360, 0.3),
                                    (655_361, 0.3), (1_100_000, 0.3), (1_400_001, 0.0), (2_000_000, 0.0), (2_000_000, 0.9)])
def test_splitter_depth_sort_agrees_with_the_oracle(oracle32, N, ties):
    """The depth sort of more than 16384 records.  A context's FIRST sort takes the four LSD passes and leaves 127
    splitters; every later one buckets the records between the previous sort's splitters, scatters them stably and sorts
    every bucket locally (GS_TUNE_SPLITTER_DEPTH_SORT = 1, default; = 0: LSD passes always).  Above 655 360 records
    (160 sort tiles; round 4, GS_TUNE_SPLITTER_DEPTH_SORT = 2 -- bit-exact but slower than the LSD passes of those sizes,
    so not the default) there are 255 splitters / 512 buckets, a chunk scan of the histogram rows between the two
    passes, and from 1.4 M records on the local sort with 16384 records per workgroup: the sizes of the grown bench
    scene (1 M) and of BASELINE's config 5 (2 M), and both sides of each boundary.  The order must be the
    oracle's stable sort whatever the splitters are: fresh ones (the same input again), STALE ones (other depths: one
    bucket then holds most of the records and is streamed through global memory by a single workgroup), depths that
    differ in their low bits only, all-equal depths, and 30 % / 90 % ties (whole "equal to splitter" classes)."""
    W, H = 640, 360
    rectMin, rectMax, radii, depths = synthetic_rects(41 + N % 7, N, W, H, near=200, near_span=80.0, far_span=6.0, p_tied=ties)
    bn = oracle32.tile_bin(rectMin, rectMax, radii, depths, W, H, 16, 16)
    d2 = (np.float32(3.0) + (np.arange(N) % 251).astype(np.float32) * np.float32(2.4e-7)).astype(np.float32)
    bn2 = oracle32.tile_bin(rectMin, rectMax, radii, d2, W, H, 16, 16)
    d3 = np.full(N, 2.5, np.float32)
    bn3 = oracle32.tile_bin(rectMin, rectMax, radii, d3, W, H, 16, 16)
    d4 = (depths * np.float32(37.0) + np.float32(0.05)).astype(np.float32)             # another range altogether
    bn4 = oracle32.tile_bin(rectMin, rectMax, radii, d4, W, H, 16, 16)
    r = _renderer(W, H)
    on = 2 if N > 655_360 else 1        # (the 512-bucket form is not the default: slower than the LSD passes it replaces)
    r.setTuning(splitter_depth_sort=on)
    bin_ = lambda d: r.buildGlobalTileSliceInfo((rectMin, rectMax), radii, d)
    _assert_same_lists(bin_(depths), bn, "first sort: LSD passes + splitters")
    _assert_same_lists(bin_(depths), bn, "fresh splitters")
    _assert_same_lists(bin_(depths), bn, "... and theirs")
    _assert_same_lists(bin_(d2), bn2, "stale splitters, narrow range")
    _assert_same_lists(bin_(d2), bn2, "narrow range, own splitters")
    _assert_same_lists(bin_(d3), bn3, "all equal")
    _assert_same_lists(bin_(d3), bn3, "all equal, own splitters")
    _assert_same_lists(bin_(d4), bn4, "stale splitters, other range")
    _assert_same_lists(bin_(depths), bn, "back again")
    r.setTuning(splitter_depth_sort=0)
    _assert_same_lists(bin_(depths), bn, "LSD passes")
    r.setTuning(splitter_depth_sort=on)
    _assert_same_lists(bin_(d4), bn4, "knob back on: LSD + splitters")
    _assert_same_lists(bin_(d4), bn4, "splitters")
    r.close()
