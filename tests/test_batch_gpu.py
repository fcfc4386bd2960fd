"""Loader-side batch assembly on the device, bit-exact against the numpy restatement."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("crop", [0, 224, 100])
def test_assemble_batch_bit_exact(dev, crop):
    from oracle import batch as OB
    from pointcloududa_amd.utils.batch import assemble_batch, to_categorical
    rng = np.random.default_rng(9)
    img = rng.normal(0, 1, (3, 256, 256, 3)).astype(np.float32)
    m = rng.integers(0, 5, (3, 256, 256, 1)).astype(np.int64)
    v = rng.integers(0, 256, (3, 300, 3)).astype(np.int64)
    ri, ro, rv = OB.assemble_batch(img, m, v, num_classes=5, crop_size=crop)
    gi, go, gv = assemble_batch(torch.from_numpy(img).to(dev), torch.from_numpy(m).to(dev), 5, crop,
                                verts=torch.from_numpy(v).to(dev))
    assert go.dtype == torch.uint8 and np.array_equal(gi.cpu().numpy(), ri) and np.array_equal(go.cpu().numpy(), ro)
    assert np.array_equal(gv.cpu().numpy(), rv)
    assert np.array_equal(to_categorical(torch.from_numpy(m).to(dev), 5).cpu().numpy(), OB.to_categorical(m, 5))


def test_assemble_batch_resamples_vertices_from_the_full_mask(dev):
    from oracle.sampler import mask_to_pointcloud
    from oracle.synth import synth_batch
    from pointcloududa_amd.utils.batch import assemble_batch
    _, mask, _, _, _ = synth_batch(2, 1, 4, 256, seed=12)
    lab = np.argmax(mask, axis=1).astype(np.int64)                       # [B,H,W] labels
    img = np.zeros((2, 256, 256, 1), dtype=np.float32)
    _, _, gv = assemble_batch(torch.from_numpy(img).to(dev), torch.from_numpy(lab).to(dev), 4, 224, resample_verts=True)
    ref = np.stack([mask_to_pointcloud((lab[i] > 0).astype(np.uint8), first=0) for i in range(2)]).astype(np.float32) / 255.0
    assert np.array_equal(gv.cpu().numpy(), ref)
