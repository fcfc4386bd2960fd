"""Farthest point sampling + surface extraction: integer / index work, bit-exact against the golden
vectors produced by the reference's graipher (tests/golden/fps.npz) and against the numpy oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD

pytestmark = pytest.mark.gpu


def test_fps_bit_exact_vs_reference_golden(dev):
    from pointcloududa_amd.utils import npy2point as S
    g = np.load(os.path.join(GOLD, "fps.npz"))
    for name in ("rand", "lattice", "dup", "surface"):
        pts = g[name + "_pts"]
        for trial in range(2):
            idx = S.fps_indices(pts, 300, int(g["%s_%d_first" % (name, trial)]))
            assert np.array_equal(idx, g["%s_%d_idx" % (name, trial)]), (name, trial)


def test_surface_and_batched_sampler_vs_oracle(dev):
    from oracle import sampler as OS
    from oracle.synth import synth_labels
    from pointcloududa_amd import kernels as K
    from pointcloududa_amd.utils import npy2point as S
    rng = np.random.default_rng(3)
    lab = synth_labels(4, 4, 256, rng)
    lab[3] = 0
    lab[3, 10:14, 10:14] = 1                     # 16 px: below the > 50 px threshold -> zeros
    lab[2, :, 0:3] = 2                           # foreground touching the image border
    m = torch.from_numpy((lab > 0).astype(np.uint8)).to(dev)
    verts, counts = K.surface_vertices(m, 3 * 4096)
    for i in range(4):
        ref = OS.surface_vertices(lab[i])
        assert int(counts[i]) == len(ref)
        assert np.array_equal(verts[i, :len(ref)].cpu().numpy(), ref)
    firsts = np.array([0, 17, 123456, 5], dtype=np.int32)
    out = S.masks_to_pointclouds(m, torch.from_numpy(firsts).to(dev)).cpu().numpy()
    for i in range(4):
        ref = OS.mask_to_pointcloud(lab[i][..., None], 300, first=int(firsts[i]))
        assert np.array_equal(out[i], ref), i
    assert np.array_equal(S.npy2point_datagenerator(lab[1][..., None], first=9),
                          OS.mask_to_pointcloud(lab[1][..., None], 300, first=9))


def test_marching_cubes_order_mode_vs_oracle(dev):
    """second vertex-list mode (cell traversal order, one vertex per crossing edge incl. coincident duplicates)"""
    from oracle import sampler as OS
    from oracle.synth import synth_labels
    from pointcloududa_amd import kernels as K
    from pointcloududa_amd.utils import npy2point as S
    rng = np.random.default_rng(5)
    lab = synth_labels(4, 4, 256, rng)
    lab[2, :, 0:3] = 2                           # foreground on the volume faces (edges owned by boundary cells)
    lab[2, 0:2, :] = 1
    lab[3] = (rng.random((256, 256)) > 0.7)      # salt and pepper: many coincident duplicates
    m = torch.from_numpy((lab > 0).astype(np.uint8)).to(dev)
    verts, counts = K.surface_vertices(m, 3 * 70000, order="mc")
    for i in range(4):
        ref = OS.surface_vertices_mc(lab[i])
        assert int(counts[i]) == len(ref), i
        assert np.array_equal(verts[i, :len(ref)].cpu().numpy(), ref), i
    small, cnt = K.surface_vertices(m[:1], 100, order="mc")           # truncated output: count still the full one
    assert int(cnt[0]) == len(OS.surface_vertices_mc(lab[0])) and np.array_equal(small[0].cpu().numpy(), OS.surface_vertices_mc(lab[0])[:100])
    firsts = np.array([0, 17, 123456], dtype=np.int32)
    out = S.masks_to_pointclouds(m[:3], torch.from_numpy(firsts).to(dev), order="mc").cpu().numpy()
    for i in range(3):
        assert np.array_equal(out[i], OS.mask_to_pointcloud(lab[i][..., None], 300, first=int(firsts[i]), order="mc")), i
    assert np.array_equal(S.npy2point_datagenerator(lab[1][..., None], first=9, order="mc"),
                          OS.mask_to_pointcloud(lab[1][..., None], 300, first=9, order="mc"))


def test_fps_edge_cases(dev):
    from oracle import sampler as OS
    from pointcloududa_amd import kernels as K
    rng = np.random.default_rng(8)
    # ragged batch: different counts, one empty cloud, k larger than the number of distinct points
    counts = [700, 0, 5, 1200]
    pts = np.zeros((4, 1200, 3))
    for i, c in enumerate(counts):
        pts[i, :c] = rng.integers(0, 50, (c, 3))
    first = [3, 0, 4, 1199]
    idx = K.fps(torch.from_numpy(pts).to(dev), torch.tensor(counts, dtype=torch.int32, device=dev),
                torch.tensor(first, dtype=torch.int32, device=dev), 300).cpu().numpy()
    assert (idx[1] == -1).all()
    for i in (0, 2, 3):
        assert np.array_equal(idx[i], OS.fps_indices(pts[i, :counts[i]], 300, first[i])), i
