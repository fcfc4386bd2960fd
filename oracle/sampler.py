"""Mask -> surface point cloud sampler, restated in numpy (oracle; test-only).

Reference (relative to /root/reference/src):
  utils/npy2point.py:7-8     calc_distances      -> _sqdist
  utils/npy2point.py:11-18   graipher (FPS)      -> fps_indices / fps_points
  utils/npy2point.py:101-125 npy2point_datagenerator -> mask_to_pointcloud

PINNED: farthest-point sampling.  ``fps_points`` reproduces ``graipher`` bit for bit
given the same vertex list and the same first index (the reference draws it with
``np.random.randint(len(pts))``, npy2point.py:13; here it is an explicit argument).
``oracle/make_golden.py`` checks that against the imported reference.

PARITY UNPINNED: vertex extraction.  The reference calls
``mcubes.marching_cubes(vol, 0)`` (PyMCubes, third-party C++, not vendored, version
not pinned by the reference, not installed here), which defines which vertices exist
and their ORDER.  ``surface_vertices`` is this build's own canonical definition:
on the 3-slice stack of the binarised mask every z-edge joins equal values, so only
in-plane edges cross the surface; with isovalue 0 on a {0,1} volume the linear
interpolation collapses each crossing onto the background end of the edge.  We emit
each such background voxel ONCE, in lexicographic (z, y, x) order.

``surface_vertices_mc`` is a second mode that follows the marching-cubes traversal PyMCubes documents (Lorensen &
Cline cells, Bourke's corner / edge numbering, vertices created once per crossing edge and shared between the cells
around it): cells in (x, y, z)-nested order, inside a cell the edges in the order 6, 5, 10 (the three edges a cell
always owns) then 0, 1, 2, 3, 4, 7, 8, 9, 11 (owned only on the lower volume faces), one vertex per crossing edge --
coincident duplicates included (a background pixel next to k foreground pixels appears k times per plane).  It is
restated from the published algorithm and from memory of the un-vendored library's structure, with NO fixture of
PyMCubes behind it: STILL PARITY UNPINNED (which library version the reference ran is not recorded either).
"""
from __future__ import annotations

import numpy as np


def _sqdist(p0: np.ndarray, pts: np.ndarray) -> np.ndarray:
    return ((p0 - pts) ** 2).sum(axis=1)


def fps_indices(pts: np.ndarray, k: int, first: int) -> np.ndarray:
    """Indices chosen by graipher: start at ``first``; then k-1 times take the argmax
    (first occurrence) of the running minimum squared distance.  float64 arithmetic."""
    pts = np.asarray(pts, dtype=np.float64)
    idx = np.empty(k, dtype=np.int64)
    idx[0] = first
    dist = _sqdist(pts[first], pts)
    for i in range(1, k):
        j = int(np.argmax(dist))
        idx[i] = j
        dist = np.minimum(dist, _sqdist(pts[j], pts))
    return idx


def fps_points(pts: np.ndarray, k: int, first: int) -> np.ndarray:
    """What graipher returns: the sampled coordinates, float64 [k, dim]."""
    pts = np.asarray(pts, dtype=np.float64)
    return pts[fps_indices(pts, k, first)]


def surface_vertices(mask2d: np.ndarray) -> np.ndarray:
    """Canonical (build-defined) vertex list for a [H,W] mask: int64 [M,3] rows (z,y,x),
    z in {0,1,2}, (y,x) = background pixels 4-adjacent to a foreground pixel."""
    fg = np.asarray(mask2d) > 0
    h, w = fg.shape
    pad = np.zeros((h + 2, w + 2), dtype=bool)
    pad[1:-1, 1:-1] = fg
    near = pad[:-2, 1:-1] | pad[2:, 1:-1] | pad[1:-1, :-2] | pad[1:-1, 2:]
    ys, xs = np.nonzero(near & ~fg)                 # row-major == lexicographic (y,x)
    m = ys.shape[0]
    out = np.empty((3 * m, 3), dtype=np.int64)
    for z in range(3):
        out[z * m:(z + 1) * m, 0] = z
        out[z * m:(z + 1) * m, 1] = ys
        out[z * m:(z + 1) * m, 2] = xs
    return out


# Bourke numbering on the cell with corner (i, j, k); axes (x, y, z) = array axes (slice, row, column).
_MC_CORNER = ((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1))
_MC_EDGE = ((0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7))
_MC_ORDER = (6, 5, 10, 0, 1, 2, 3, 4, 7, 8, 9, 11)
# an edge is created by the FIRST cell (in traversal order) that contains it: 6, 5, 10 always by this cell; the others only
# where none of the up to three earlier cells around the edge exists, i.e. when ALL the named cell indices are 0
_MC_NEW_IF = {0: "jk", 1: "k", 2: "k", 3: "ik", 4: "j", 7: "i", 8: "ij", 9: "j", 11: "i"}


def surface_vertices_mc(mask2d: np.ndarray) -> np.ndarray:
    """Marching-cubes-order vertex list (see the module docstring; parity unpinned) of the 3-slice stack of a
    [H,W] mask at isovalue 0: int64 [M,3] rows (slice, row, column).  'inside' = value <= 0 (background); with
    values in {0,1} the interpolated vertex sits on the background end of the crossing edge."""
    fg = (np.asarray(mask2d) > 0)
    h, w = fg.shape
    vol = np.broadcast_to(fg[None], (3, h, w))
    ni, nj, nk = 2, h - 1, w - 1
    if nj <= 0 or nk <= 0:
        return np.zeros((0, 3), dtype=np.int64)
    ii, jj, kk = np.meshgrid(np.arange(ni), np.arange(nj), np.arange(nk), indexing="ij")
    cell = (ii * nj + jj) * nk + kk
    corner = lambda c: vol[c[0]:c[0] + ni, c[1]:c[1] + nj, c[2]:c[2] + nk]
    keys, rows = [], []
    for rank, e in enumerate(_MC_ORDER):
        a, b = _MC_CORNER[_MC_EDGE[e][0]], _MC_CORNER[_MC_EDGE[e][1]]
        fa, fb = corner(a), corner(b)
        emit = fa != fb
        if e in _MC_NEW_IF:
            for ax in _MC_NEW_IF[e]:
                emit = emit & ({"i": ii, "j": jj, "k": kk}[ax] == 0)
        sel = np.nonzero(emit)
        at_a = ~fa[sel]                                  # the background end of the edge
        pos = [np.where(at_a, idx + a[d], idx + b[d]) for d, idx in enumerate(sel)]
        keys.append(cell[sel] * 12 + rank)
        rows.append(np.stack(pos, axis=1))
    keys, rows = np.concatenate(keys), np.concatenate(rows)
    return rows[np.argsort(keys, kind="stable")].astype(np.int64)


def mask_to_pointcloud(mask_hw1: np.ndarray, number_points: int = 300, first: int = 0,
                       fps: bool = True, order: str = "lex") -> np.ndarray:
    """npy2point_datagenerator (npy2point.py:101-125) with the canonical extraction above.

    mask_hw1: [H,W,1] (or [H,W]) integer labels.  Returns int64 [number_points,3] (z,y,x);
    all zeros when the binarised mask has <= 50 foreground pixels (npy2point.py:116).
    ``first`` indexes the vertex list and stands in for np.random.randint.
    """
    m = np.asarray(mask_hw1)
    if m.ndim == 3:
        m = m[..., 0]
    m = np.where(m > 0, 1, 0)
    verts = np.zeros((number_points, 3), dtype=np.int64)
    if m.sum() > 50:
        v = surface_vertices(m) if order == "lex" else surface_vertices_mc(m)
        if fps and len(v) > 0:
            v = fps_points(v, number_points, first % len(v))
        verts = np.array(v, dtype=np.int64)
    return verts
