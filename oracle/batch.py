"""Batch assembly of the reference's data generators, restated in numpy (oracle; test-only).

Reference: data_generator_mmwhs.py:265-274 (crop, moveaxis, to_categorical, verts / 255),
ImageProcessor.crop_volume :134-137, utils/utils.py:7-29 (to_categorical)."""
from __future__ import annotations

import numpy as np


def crop_volume(vol: np.ndarray, crop_size: int) -> np.ndarray:
    """data_generator_mmwhs.py:134-137 (note: called with crop_size // 2)."""
    return np.array(vol[:, int(vol.shape[1] / 2) - crop_size: int(vol.shape[1] / 2) + crop_size,
                        int(vol.shape[2] / 2) - crop_size: int(vol.shape[2] / 2) + crop_size])


def to_categorical(mask: np.ndarray, num_classes: int) -> np.ndarray:
    """utils.py:7-29, channel_first."""
    assert num_classes > 1
    if mask.ndim == 4 and mask.shape[1] == 1:
        mask = np.squeeze(mask, axis=1)
    if mask.shape[-1] == 1:
        mask = np.squeeze(mask, axis=-1)
    return np.moveaxis(np.eye(num_classes, dtype="uint8")[mask], -1, 1)


def assemble_batch(images_hwc: np.ndarray, masks_hw1: np.ndarray, verts_int, num_classes: int = 5, crop_size: int = 0):
    """data_generator_mmwhs.py:265-274 for channel_first."""
    images, masks = np.asarray(images_hwc), np.asarray(masks_hw1)
    if crop_size:
        images = crop_volume(images, crop_size // 2)
        masks = crop_volume(masks, crop_size // 2)
    images = np.moveaxis(images, -1, 1)
    masks = to_categorical(masks, num_classes)
    verts = np.array(verts_int, np.float32) / 255.0
    return images, masks, verts
