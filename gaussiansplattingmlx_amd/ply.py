"""Host-side mirror of Data/PlyWriter.swift over the C ABI: the reference's static methods, same names.

Tensors are device tensors (or anything the renderer can move to the device); the interleave / de-interleave runs on
the GPU and the file bytes are those of the reference writer."""
from __future__ import annotations

import ctypes as C
import os

import torch

from .renderer import GaussianRenderer, _p


class PlyWriter:
    def __init__(self, renderer: GaussianRenderer):
        self.r = renderer

    def writeGaussianBinary(self, positions, features_dc, features_rest, opacities, scales, rotations, to):
        """PlyWriter.writeGaussianBinary(positions:features_dc:features_rest:opacities:scales:rotations:to:)
        (Data/PlyWriter.swift:116-146).  features_rest must be [N, M, 3] (the reference's precondition, :131)."""
        r = self.r
        t = [r._t(x) for x in (positions, features_dc, features_rest, opacities, scales, rotations)]
        N = int(t[0].shape[0])
        if t[2].dim() != 3 or t[2].shape[2] != 3:
            raise ValueError("features_rest must be [N, M, 3]")
        sizes = [3, 3, 3 * int(t[2].shape[1]), 1, 3, 4]
        for x, n in zip(t, sizes):
            if x.numel() != N * n:
                raise ValueError("Attribute array size mismatch")          # PlyWriter.swift:34-43
        K = int(t[2].shape[1]) + 1
        r._check(r.lib.gs_ply_write(r.ctx, os.fsencode(os.fspath(to)), N, K, *[_p(x) for x in t]))

    def probe(self, url):
        n, m, d = C.c_longlong(), C.c_int(), C.c_int()
        r = self.r
        r._check(r.lib.gs_ply_probe(r.ctx, os.fsencode(os.fspath(url)), C.byref(n), C.byref(m), C.byref(d)))
        return int(n.value), int(m.value), int(d.value)

    def loadGaussianBinaryPLYAsMLX(self, url):
        """PlyWriter.loadGaussianBinaryPLYAsMLX (Data/PlyWriter.swift:235-265): the six tensors, on the device, in
        the reference's shapes -- positions [N,3], features_dc [N,1,3], features_rest [N,M,3], opacities [N,1],
        scales [N,3], rotations [N,4]."""
        r = self.r
        N, M, D = self.probe(url)
        out = dict(positions=r._empty(N, 3), features_dc=r._empty(N, 1, 3), features_rest=r._empty(N, M, 3),
                   opacities=r._empty(N, 1), scales=r._empty(N, 3), rotations=r._empty(N, 4))
        if D != 3:
            raise ValueError("features_rest_shape must be M 3")
        r._check(r.lib.gs_ply_load(r.ctx, os.fsencode(os.fspath(url)), N, M + 1, _p(out["positions"]),
                                   _p(out["features_dc"]), _p(out["features_rest"]), _p(out["opacities"]),
                                   _p(out["scales"]), _p(out["rotations"])))
        return out

    def packRows(self, positions, features_dc, features_rest, opacities, scales, rotations) -> torch.Tensor:
        r = self.r
        t = [r._t(x) for x in (positions, features_dc, features_rest, opacities, scales, rotations)]
        N, K = int(t[0].shape[0]), int(t[2].shape[1]) + 1
        rows = r._empty(N, 14 + 3 * (K - 1))
        r._check(r.lib.gs_ply_pack_rows(r.ctx, N, K, *[_p(x) for x in t], _p(rows)))
        return rows
