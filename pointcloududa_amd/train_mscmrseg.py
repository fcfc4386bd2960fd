"""The epoch-level functions of ``src/train_mscmrseg.py`` on the HIP path, under the reference's names and signatures:

    from pointcloududa_amd import train_mscmrseg as T
    T.args = args                                  # the script's module-global namespace (-d1 -d2 -d4 -dr -wp ...)
    result = T.train_epoch(model_gen, model_dis2, model_dis4, model_dis1, optim_gen, optim_dis2, optim_dis4, optim_dis1,
                           trainA_iterator, trainB_iterator)           # train_mscmrseg.py:142-345
    valid  = T.valid_model(model_gen, validA_iterator, validB_iterator, testB_generator)      # :102-139

The iterators are the reference's host-numpy generators, the optimisers its ``torch.optim`` objects (``_epoch.py``).
"""
from __future__ import annotations

from . import _epoch as E

args = None      # set by the caller, as the reference's ``if __name__ == '__main__'`` block does (train_mscmrseg.py:675-697)


def _args(a):
    a = a if a is not None else args
    if a is None:
        raise ValueError("set pointcloududa_amd.train_mscmrseg.args (or pass args=...) first: the reference reads a module-global")
    return a


def train_epoch(model_gen, model_dis2, model_dis4, model_dis1=None, optim_gen=None, optim_dis2=None, optim_dis4=None,
                optim_dis1=None, trainA_iterator=None, trainB_iterator=None, *, args=None):
    """train_mscmrseg.py:142-345"""
    return E.train_epoch("mscmrseg", _args(args), model_gen, model_dis2, model_dis4, model_dis1, optim_gen, optim_dis2,
                         optim_dis4, optim_dis1, trainA_iterator, trainB_iterator)


def valid_model_with_one_dataset(seg_model, data_generator, hd=False, *, args=None):
    """train_mscmrseg.py:53-99"""
    return E.valid_model_with_one_dataset("mscmrseg", _args(args), seg_model, data_generator, hd)


def valid_model(seg_model, validA_iterator, validB_iterator, testB_generator, *, args=None):
    """train_mscmrseg.py:102-139"""
    return E.valid_model("mscmrseg", _args(args), seg_model, validA_iterator, validB_iterator, testB_generator)
