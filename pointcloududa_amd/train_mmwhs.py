"""The epoch-level functions of ``src/train_mmwhs.py`` on the HIP path, under the reference's names and signatures
(``train_epoch`` :144-377, ``valid_model_with_one_dataset`` :55-100, ``valid_model`` :102-141): softmax / sigmoid mode,
the w1 / w2 / w4 weights, ``-etpls`` / ``-Tetpls`` / ``-d4aux`` and ``-sgd`` are read from ``args`` like the script does.
See ``train_mscmrseg.py`` of this package for the calling convention."""
from __future__ import annotations

from . import _epoch as E

args = None      # the script's module-global namespace (train_mmwhs.py:815-872)


def _args(a):
    a = a if a is not None else args
    if a is None:
        raise ValueError("set pointcloududa_amd.train_mmwhs.args (or pass args=...) first: the reference reads a module-global")
    return a


def train_epoch(model_gen, model_dis2, model_dis4, model_dis1=None, optim_gen=None, optim_dis2=None, optim_dis4=None,
                optim_dis1=None, trainA_iterator=None, trainB_iterator=None, *, args=None):
    """train_mmwhs.py:144-377"""
    return E.train_epoch("mmwhs", _args(args), model_gen, model_dis2, model_dis4, model_dis1, optim_gen, optim_dis2,
                         optim_dis4, optim_dis1, trainA_iterator, trainB_iterator)


def valid_model_with_one_dataset(seg_model, data_generator, hd=False, *, args=None):
    """train_mmwhs.py:55-100"""
    return E.valid_model_with_one_dataset("mmwhs", _args(args), seg_model, data_generator, hd)


def valid_model(seg_model, validA_iterator, validB_iterator, testB_generator, *, args=None):
    """train_mmwhs.py:102-141"""
    return E.valid_model("mmwhs", _args(args), seg_model, validA_iterator, validB_iterator, testB_generator)
