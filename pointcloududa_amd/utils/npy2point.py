"""HIP-backed mask -> surface point cloud sampler (reference: ``src/utils/npy2point.py``).

``graipher`` / ``npy2point_datagenerator`` keep the reference's names and argument meaning
(npy2point.py:11-18, 101-125); ``masks_to_pointclouds`` is the batched device-side form the
synthetic-data path and a GPU loader use.  Farthest point sampling is bit-exact with the numpy
reference given the same vertex list and first index.  The vertex LIST comes from this build's own
canonical surface extraction (PyMCubes, which the reference calls, is not vendored): parity of the
extraction itself is unpinned -- see oracle/sampler.py.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import kernels as K


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("the sampler runs on a HIP device only (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def fps_indices(pts, k, first):
    """pts: [N,dim<=3] array-like; returns int64 numpy indices chosen by graipher started at `first`."""
    p = np.zeros((1, len(pts), 3), dtype=np.float64)
    a = np.asarray(pts, dtype=np.float64)
    p[0, :, :a.shape[1]] = a
    dev = _dev()
    idx = K.fps(torch.from_numpy(p).to(dev), torch.tensor([len(pts)], dtype=torch.int32, device=dev),
                torch.tensor([int(first)], dtype=torch.int32, device=dev), k)
    return idx[0].cpu().numpy().astype(np.int64)


def graipher(pts, K_, dim=2, first=None):
    """npy2point.py:11-18.  ``first`` replaces the reference's ``np.random.randint(len(pts))`` draw
    (drawn from numpy's global RNG, exactly as the reference does, when omitted)."""
    pts = np.asarray(pts, dtype=np.float64)
    if first is None:
        first = np.random.randint(len(pts))
    return pts[fps_indices(pts, K_, first)][:, :dim]


def masks_to_pointclouds(mask_u8: torch.Tensor, firsts: torch.Tensor, number_points: int = 300,
                         max_verts: int = 0, order: str = "lex") -> torch.Tensor:
    """mask_u8: uint8 [B,H,W] on the device (>0 = foreground); firsts: int32 [B].
    -> int32 [B,number_points,3] rows (z,y,x); all zeros where the mask has <= 50 foreground pixels.
    ``order``: "lex" = the canonical vertex list, "mc" = marching-cubes traversal order with the coincident
    duplicates of one vertex per crossing edge (both parity-unpinned against PyMCubes, oracle/sampler.py)."""
    b, h, w = mask_u8.shape
    if max_verts <= 0:
        max_verts = 3 * 8 * (h + w) * (2 if order == "mc" else 1)
    verts, counts = K.surface_vertices(mask_u8.contiguous(), max_verts, order)
    # npy2point.py:116: sample only when the binarised mask has more than 50 foreground pixels
    area = (mask_u8 > 0).flatten(1).sum(1)
    counts = torch.where(area > 50, counts, torch.zeros_like(counts)).to(torch.int32)
    n_used = int(counts.max())                       # one host sync: this is loader-side code
    if n_used > max_verts:
        raise RuntimeError("surface has more than max_verts=%d vertices" % max_verts)
    verts = verts[:, :max(n_used, 1)].contiguous()
    idx = K.fps(verts.to(torch.float64), counts, firsts.to(torch.int32), number_points)
    safe = idx.clamp(min=0).long()
    out = torch.gather(verts, 1, safe[..., None].expand(-1, -1, 3))
    return torch.where((idx >= 0)[..., None], out, torch.zeros_like(out))


def npy2point_datagenerator(mask=None, number_points=300, dim=3, crop_size=112, tocrop=False, fps=True, first=0,
                            order="lex"):
    """npy2point.py:101-125 for one [H,W,1] (or [H,W]) integer mask -> int array [number_points, 3]."""
    if tocrop or not fps:
        raise NotImplementedError("tocrop=True / fps=False are not used by the data generators")
    m = np.asarray(mask)
    if m.ndim == 3:
        m = m[..., 0]
    dev = _dev()
    mu8 = torch.from_numpy((m > 0).astype(np.uint8))[None].to(dev)
    out = masks_to_pointclouds(mu8, torch.tensor([int(first)], dtype=torch.int32, device=dev), number_points, order=order)
    return out[0].cpu().numpy().astype(np.int64)
