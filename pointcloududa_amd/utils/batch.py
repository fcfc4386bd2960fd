"""Loader-side batch assembly on the device.

What the reference's data generators do to a batch AFTER augmentation (``src/data_generator_mmwhs.py:255-274``,
``src/utils/utils.py:7-29``, ``crop_volume`` ``:134-137``): re-sample the surface point cloud of every
(augmented) mask, centre-crop, move channels first, one-hot the labels, scale the vertices by 1/255.
CSV/NIfTI reading and imgaug stay on the CPU (out of scope); the raw ``[B,H,W,C]`` images and integer masks are
uploaded once and everything else happens here."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import kernels as K
from .npy2point import masks_to_pointclouds


def to_categorical(mask: torch.Tensor, num_classes: int) -> torch.Tensor:
    """``utils.py:7-29`` (channel_first): integer mask ``[B,H,W]`` / ``[B,1,H,W]`` / ``[B,H,W,1]`` on the device ->
    one-hot uint8 ``[B,num_classes,H,W]``.  Unlike the reference's asserts, value checks are not made here (they
    would cost a host sync); labels outside ``[0, num_classes)`` give an all-zero pixel."""
    if mask.dim() == 4 and mask.shape[1] == 1:
        mask = mask[:, 0]
    if mask.dim() == 4 and mask.shape[-1] == 1:
        mask = mask[..., 0]
    b, h, w = mask.shape
    dummy = torch.empty((b, h, w, 1), dtype=torch.float32, device=mask.device)
    return K.assemble_batch(dummy, mask.to(torch.int32), num_classes, 0)[1]


def assemble_batch(images_hwc: torch.Tensor, masks: torch.Tensor, num_classes: int = 5, crop_size: int = 0,
                   verts: Optional[torch.Tensor] = None, resample_verts: bool = False,
                   firsts: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """images ``[B,H,W,C]`` fp32, masks ``[B,H,W]`` (or ``[B,H,W,1]``) integer labels, both on the device ->
    ``(images [B,C,h,w] fp32, masks one-hot uint8 [B,K,h,w], verts fp32 [B,300,3] / 255 or None)``.

    ``resample_verts`` reproduces the augmentation branch (``:255-263``): the point cloud is re-sampled from the
    FULL-size mask (before the crop) with the HIP sampler; ``firsts`` [B] int32 replaces the reference's random
    first FPS index (default 0).  Otherwise ``verts`` (the integer vertices stored with the data set) are only
    scaled."""
    if masks.dim() == 4 and masks.shape[-1] == 1:
        masks = masks[..., 0]
    lab = masks.to(torch.int32)
    if resample_verts:
        if firsts is None:
            firsts = torch.zeros(lab.shape[0], dtype=torch.int32, device=lab.device)
        verts = masks_to_pointclouds((lab > 0).to(torch.uint8), firsts)
    img, onehot = K.assemble_batch(images_hwc, lab, num_classes, crop_size)
    # tensor / tensor: a true IEEE division (the scalar form multiplies by a rounded 1/255, off in the last bit)
    v = None if verts is None else verts.to(torch.float32) / torch.full((), 255.0, dtype=torch.float32, device=verts.device)
    return img, onehot, v
