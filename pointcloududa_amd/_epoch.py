"""Epoch-level callers of the hot path with the reference's signatures ("callers").

``train_epoch`` / ``valid_model`` / ``valid_model_with_one_dataset`` of ``src/train_mscmrseg.py:53-345`` and
``src/train_mmwhs.py:55-377`` keep working after the import swap: same positional / keyword arguments, the same
host-numpy iterators (``(x[B,Cin,H,W] f32, y[B,C,H,W] uint8 one-hot, z[B,300,3] f32)`` per batch,
``data_generator_mscmrseg.py:274-319``), externally built ``torch.optim`` objects, the same result dictionaries.

What differs is where the work happens, not what is computed:

* the loop body is ``AdversarialTrainer.step`` (the HIP kernels); a step never synchronises with the host, the
  per-step metrics stay on the device and ONE transfer at the end of the epoch produces the epoch means the
  reference computes with ``.item()`` / ``.cpu().numpy()`` every step (train_mscmrseg.py:207-216,270-322);
* the reference's ``torch.tensor(np).cuda()`` per batch (:201,219) becomes a copy through pinned staging buffers on a
  copy stream, one batch ahead of the step that consumes it;
* the ``torch.optim.Adam`` / ``SGD`` objects the caller built are ADOPTED: their hyper-parameters (read again at
  every epoch: the scripts decay ``param_group['lr']`` between epochs, :585-589) and state drive the fused flat
  optimisers, and after the epoch their ``state`` holds the fused optimisers' state again, so
  ``optimizer.state_dict()`` in the caller's checkpoint code (callbacks.py:78-80) stays genuine.

The scripts read a module-global ``args``; here it is a keyword argument (or the module attribute ``args`` of
``pointcloududa_amd.train_mscmrseg`` / ``train_mmwhs``, set by the caller exactly as the scripts' ``__main__`` does).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Iterator, Optional

import numpy as np
import torch

from .optim import FusedAdam, FusedSGD
from .train_step import AdversarialTrainer, TrainCfg


# ------------------------------------------------------------------------------------------------ batches
class DeviceBatches:
    """Host-numpy batches -> device tensors, one batch ahead: pinned staging buffers, ``non_blocking`` copies on a
    copy stream, an event per batch that the consumer's stream waits on.  Replaces ``torch.tensor(x).cuda()`` per
    batch (train_mscmrseg.py:68,201,219).  Tensors that already live on the device pass through."""

    def __init__(self, iterator: Iterable, device: torch.device, depth: int = 2):
        self.it: Iterator = iter(iterator)
        self.dev = device
        self.depth = max(1, depth)
        self.stream = torch.cuda.Stream(device=device) if device.type == "cuda" else None
        self.slots = [dict() for _ in range(self.depth + 1)]      # pinned buffers, reused round-robin
        self.free_ev = [None] * (self.depth + 1)
        self.queue, self.n = [], 0
        self.done = False          # latched at the first StopIteration: the reference's generators START A NEW EPOCH when
        #                            asked again (data_generator_mscmrseg.py:281-283 resets the count as it raises)

    @staticmethod
    def _as_tensor(a):
        if torch.is_tensor(a):
            return a
        a = np.asarray(a)
        if a.dtype == np.float64:
            a = a.astype(np.float32)
        if a.dtype == np.bool_:
            a = a.astype(np.uint8)
        return torch.from_numpy(np.ascontiguousarray(a))

    def _put(self, slot, key, a):
        t = self._as_tensor(a)
        if t.device == self.dev:
            return t
        if self.stream is None:
            return t.to(self.dev)
        buf = slot.get(key)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = slot[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        buf.copy_(t)
        return buf.to(self.dev, non_blocking=True)

    def _fetch(self) -> bool:
        if self.done:
            return False
        try:
            item = next(self.it)
        except StopIteration:
            self.done = True
            return False
        i = self.n % len(self.slots)
        if self.free_ev[i] is not None:
            self.free_ev[i].synchronize()          # the slot's previous copies have left its pinned buffers
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                out = tuple(None if a is None else self._put(self.slots[i], j, a) for j, a in enumerate(item))
                ev = torch.cuda.Event()
                ev.record(self.stream)
            self.free_ev[i] = ev
        else:
            out, ev = tuple(None if a is None else self._put(self.slots[i], j, a) for j, a in enumerate(item)), None
        self.queue.append((out, ev))
        self.n += 1
        return True

    def __iter__(self):
        return self

    def __next__(self):
        while len(self.queue) < self.depth and self._fetch():
            pass
        if not self.queue:
            raise StopIteration
        out, ev = self.queue.pop(0)
        if ev is not None:
            torch.cuda.current_stream(self.dev).wait_event(ev)
            for t in out:      # the caching allocator must not hand the block to the copy stream's next batch early
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(torch.cuda.current_stream(self.dev))
        self._fetch()          # the next batch's copies go out under this batch's step
        return out


# ------------------------------------------------------------------------------------------------ optimisers
def _hyper(opt):
    if len(opt.param_groups) != 1:
        raise ValueError("adopt: one param_group per optimiser (the reference builds each from model.parameters())")
    return opt.param_groups[0]


def adopt_hyperparameters(fused, torch_opt) -> None:
    """(re-)read lr / betas / eps / momentum / weight decay from the caller's ``torch.optim`` object"""
    g = _hyper(torch_opt)
    fused.lr = float(g["lr"])
    fused.wd = float(g.get("weight_decay", 0.0))
    if isinstance(fused, FusedAdam):
        fused.betas, fused.eps = tuple(float(b) for b in g["betas"]), float(g["eps"])
        if g.get("amsgrad", False) or g.get("maximize", False):
            raise NotImplementedError("adopt: Adam(amsgrad / maximize) is not what the reference builds")
    else:
        if float(g.get("dampening", 0)) != 0 or g.get("nesterov", False) or g.get("maximize", False):
            raise NotImplementedError("adopt: SGD(dampening / nesterov / maximize) is not what the reference builds")
        if float(g.get("momentum", 0.0)) != fused.momentum:
            if (fused.buf is None) != (float(g.get("momentum", 0.0)) == 0.0):
                raise ValueError("adopt: momentum switched between zero and non-zero after the first epoch")
            fused.momentum = float(g.get("momentum", 0.0))


def adopt_optimizer(torch_opt, module, skip_prefixes=()):
    """A fused flat optimiser over ``module`` that continues ``torch_opt``: same hyper-parameters, same state.
    ``torch_opt`` must have been built from ``module.parameters()`` (train_mscmrseg.py:427-455), as a
    ``torch.optim.Adam`` or ``torch.optim.SGD``; the flat optimisers of this package pass through."""
    if torch_opt is None or isinstance(torch_opt, (FusedAdam, FusedSGD)):
        return torch_opt
    g = _hyper(torch_opt)
    ids = [id(p) for p in g["params"]]
    if ids != [id(p) for p in module.parameters()]:
        raise ValueError("adopt: the optimiser was not built from this module's parameters()")
    if isinstance(torch_opt, torch.optim.Adam) and not isinstance(torch_opt, torch.optim.AdamW):
        fused = FusedAdam(module, lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g.get("weight_decay", 0.0),
                          skip_prefixes=skip_prefixes or None)
    elif isinstance(torch_opt, torch.optim.SGD):
        fused = FusedSGD(module, lr=g["lr"], momentum=g.get("momentum", 0.0), weight_decay=g.get("weight_decay", 0.0),
                         skip_prefixes=skip_prefixes)
    else:
        raise TypeError("adopt: %s is neither torch.optim.Adam nor torch.optim.SGD" % type(torch_opt).__name__)
    adopt_hyperparameters(fused, torch_opt)
    sd = torch_opt.state_dict()
    if sd["state"]:
        fused.load_torch_state_dict(sd)
    return fused


def _state_token(torch_opt):
    """identifies the state OBJECTS a torch.optim optimiser holds right now: ``load_state_dict`` replaces every one of them"""
    return tuple((k_, id(v), v.data_ptr() if torch.is_tensor(v) else None)
                 for st in torch_opt.state.values() for k_, v in sorted(st.items()))


def export_state(fused, torch_opt):
    """the fused optimiser's state back into the caller's object: ``torch_opt.state_dict()`` is what the reference's
    checkpoint callback saves (callbacks.py:78-80).  Returns a token of the exported state objects: if the caller's
    optimiser holds other ones at the next epoch (``optimizer.load_state_dict(checkpoint)`` in between: resume, rollback),
    that state is imported again instead of being overwritten (``_trainer_for``)."""
    if torch_opt is None or torch_opt is fused:
        return None
    sd = fused.torch_state_dict()
    sd["param_groups"] = torch_opt.state_dict()["param_groups"]       # the caller's own groups (keys of ITS torch version)
    torch_opt.load_state_dict(sd)
    return _state_token(torch_opt)


# ------------------------------------------------------------------------------------------------ train_epoch
def _cfg_from(args, variant, opts) -> TrainCfg:
    g = lambda k, d: getattr(args, k, d)
    lr = lambda o, d: float(_hyper(o)["lr"]) if (o is not None and hasattr(o, "param_groups")) else (o.lr if o is not None else d)
    og, o1, o2, o4 = opts
    mom = 0.99 if variant == "mscmrseg" else 0.95
    for o in (o1, o2, o4):
        if o is not None:
            mom = float(_hyper(o).get("momentum", mom)) if hasattr(o, "param_groups") else o.momentum
            break
    return TrainCfg(variant=variant, d1=bool(g("d1", False)), d2=bool(g("d2", False)), d4=bool(g("d4", False)),
                    dr=float(g("dr", 0.01)), wp=float(g("wp", 1.0)), lr=lr(og, 1e-3), d1lr=lr(o1, 2.5e-5), d2lr=lr(o2, 2.5e-5),
                    d4lr=lr(o4, 2.5e-5), d_momentum=mom, softmax=bool(g("softmax", True)), w1=float(g("w1", 1.0)),
                    w2=float(g("w2", 1.0)), w4=float(g("w4", 1.0)), n_class=int(g("n_class", 0)) or _n_class(args),
                    etpls=bool(g("etpls", False)), Tetpls=bool(g("Tetpls", False)), d4aux=bool(g("d4aux", False)),
                    gen_sgd=isinstance(og, (torch.optim.SGD, FusedSGD)))


def _n_class(args) -> int:
    return int(getattr(args, "nclass", 0) or getattr(args, "n_class", 0) or 4)


def _trainer_for(model_gen, model_dis1, model_dis2, model_dis4, opts, args, variant) -> AdversarialTrainer:
    """One trainer per (models, optimisers) combination, kept on the segmenter between epochs."""
    og, o1, o2, o4 = opts
    key = (id(model_dis1), id(model_dis2), id(model_dis4), id(og), id(o1), id(o2), id(o4), variant,
           bool(getattr(args, "d1", False)), bool(getattr(args, "d2", False)), bool(getattr(args, "d4", False)))
    cached = getattr(model_gen, "_pcuda_trainer", None)
    if cached is not None and cached[0] == key:
        tr = cached[1]
    else:
        cfg = _cfg_from(args, variant, opts)
        cfg.n_class = int(model_gen.classifier.weight.shape[0]) if hasattr(model_gen, "classifier") else cfg.n_class
        tr = AdversarialTrainer(model_gen, model_dis1, model_dis2, model_dis4, cfg)
        # the trainer built its own optimisers from cfg; continue the caller's instead (hyper-parameters + state)
        pairs = (("opt_gen", og, tr.gen), ("opt_d1", o1, tr.dis1), ("opt_d2", o2, tr.dis2), ("opt_d4", o4, tr.dis4))
        for name, o, mod in pairs:
            if o is None or mod is None:
                continue
            cur = getattr(tr, name)
            if isinstance(o, (FusedAdam, FusedSGD)):
                setattr(tr, name, o)
                continue
            want = FusedAdam if isinstance(o, torch.optim.Adam) else FusedSGD
            if not isinstance(cur, want):
                # torch.optim.SGD skips parameters whose .grad is None: the never-used encoder.conv1_1, and the point
                # head when no loss is attached to it (no weight decay, no momentum buffer for them)
                skip = (("encoder.conv1_1.",) + (() if (cfg.d4 or cfg.d4aux) else ("pointNet.",))) if mod is tr.gen else ()
                cur = adopt_optimizer(o, mod, skip_prefixes=skip)
                setattr(tr, name, cur)
            adopt_hyperparameters(cur, o)
            sd = o.state_dict()
            if sd["state"]:
                cur.load_torch_state_dict(sd)
        tr._torch_opts = {"opt_gen": og, "opt_d1": o1, "opt_d2": o2, "opt_d4": o4}
        tr._exported = {n_: _state_token(o_) for n_, o_ in tr._torch_opts.items() if o_ is not None and hasattr(o_, "param_groups")}
        model_gen._pcuda_trainer = (key, tr)
    # every epoch: the scripts mutate param_group['lr'] between epochs (train_mscmrseg.py:585-589), and -dr / -wp style
    # weights are read from args at every step in the reference
    for name, o in tr._torch_opts.items():
        f = getattr(tr, name)
        if f is not None and o is not None and hasattr(o, "param_groups"):
            adopt_hyperparameters(f, o)
            if _state_token(o) != tr._exported.get(name):
                # the caller replaced the optimiser's state since the last epoch (load_state_dict of a checkpoint): continue
                # from THAT state, as the reference's loop would
                f.load_torch_state_dict(o.state_dict())
                tr._exported[name] = _state_token(o)
    tr.cfg.dr, tr.cfg.wp = float(getattr(args, "dr", tr.cfg.dr)), float(getattr(args, "wp", tr.cfg.wp))
    tr._wp.fill_(tr.cfg.wp)
    for k in ("w1", "w2", "w4"):
        setattr(tr.cfg, k, float(getattr(args, k, getattr(tr.cfg, k))))
    return tr


_MEAN_KEYS = ("seg_dice", "ver_s_loss", "ver_t_loss", "entropy_loss", "entropy_loss_T")


def train_epoch(variant, args, model_gen, model_dis2, model_dis4, model_dis1=None, optim_gen=None, optim_dis2=None,
                optim_dis4=None, optim_dis1=None, trainA_iterator=None, trainB_iterator=None) -> Dict[str, float]:
    """One epoch of the reference's ``train_epoch`` (train_mscmrseg.py:142-345 / train_mmwhs.py:144-377) -> its result
    dictionary (epoch means: ``seg_loss``, ``seg_dice``, ``dis{1,2,4}_acc{1,2}``, ``ver_s_loss``, ``ver_t_loss``;
    MM-WHS also ``entropy_loss`` / ``entropy_loss_T``).  A mean over no values is NaN, as ``np.mean([])`` is there."""
    if args is None:
        raise ValueError("train_epoch: no args (the scripts' module-global namespace: d1, d2, d4, dr, wp, ...)")
    d1, d2, d4 = (bool(getattr(args, k, False)) for k in ("d1", "d2", "d4"))
    dev = next(model_gen.parameters()).device
    tr = _trainer_for(model_gen, model_dis1 if d1 else None, model_dis2 if d2 else None, model_dis4 if d4 else None,
                      (optim_gen, optim_dis1 if d1 else None, optim_dis2 if d2 else None, optim_dis4 if d4 else None),
                      args, variant)
    tr.train()
    # the reference walks zip(trainA_iterator, trainB_iterator): next(A), then next(B), until the first StopIteration.
    # Read ahead over the PAIR, so that the callers' generators see exactly that sequence of next() calls (the one that
    # runs out first is asked once and never again; the other one is not advanced past what zip would have taken)
    pairs = DeviceBatches((tuple(a) + (b[0], None, b[2]) for a, b in zip(trainA_iterator, trainB_iterator)), dev)   # B's mask is never read (:219)
    acc: Dict[str, list] = {}
    steps = 0
    for item in pairs:
        if len(item) != 6:
            raise ValueError("train_epoch: the iterators yield (image, mask, vertices) triples (data_generator_mscmrseg.py:319)")
        img_a, mask_a, vert_a, img_b, _, vert_b = item
        if mask_a.dtype != torch.uint8:
            mask_a = mask_a.to(torch.uint8)
        out = tr.step(img_a, mask_a, vert_a, img_b, vert_b)
        for k, v in out.items():
            acc.setdefault(k, []).append(v)
        steps += 1
    for name, o in tr._torch_opts.items():
        if getattr(tr, name) is not None and o is not None and hasattr(o, "param_groups"):
            tr._exported[name] = export_state(getattr(tr, name), o)
    res: Dict[str, float] = {}
    if steps:
        keys = sorted(acc)
        means = torch.stack([torch.stack([t.float() for t in acc[k]]).mean() for k in keys]).tolist()     # ONE sync per epoch
        m = dict(zip(keys, means))
        res["seg_loss"] = m["loss_bce"] + m["loss_jac"]                  # mean of per-step (bce + jaccard).item(), :211
        for k in _MEAN_KEYS:
            if k in m:
                res[k] = m[k]
        for d, on in (("dis2", d2), ("dis1", d1), ("dis4", d4)):
            if on:
                res[d + "_acc1"] = m[d + "_hit_src"]                     # mean(sigmoid(D) >= .5) on source, :270-291
                res[d + "_acc2"] = 1.0 - m[d + "_hit_tgt"]               # 1 - mean(...) on target, :294-322
    else:
        res["seg_loss"] = res["seg_dice"] = math.nan
    for k in ("ver_s_loss", "ver_t_loss") + (("entropy_loss", "entropy_loss_T") if variant == "mmwhs" else ()):
        res.setdefault(k, math.nan)                                      # np.mean of an empty list in the reference
    return res


# ------------------------------------------------------------------------------------------------ validation
def valid_model_with_one_dataset(variant, args, seg_model, data_generator, hd: bool = False) -> Dict[str, float]:
    """train_mscmrseg.py:53-99 / train_mmwhs.py:55-100 over a host-numpy generator; ``hd=True`` (medpy Hausdorff, CPU) is
    out of scope.  Result keys as in the respective script: dice, loss and ``valid_vert_loss`` (MS-CMRSeg) or
    ``vert_loss`` (MM-WHS)."""
    if hd:
        raise NotImplementedError("Hausdorff distance (medpy, CPU, evaluation only) is out of scope: SURVEY section 2.1 row 7")
    from . import validate as V
    dev = next(seg_model.parameters()).device

    def batches():
        for x, y, z in DeviceBatches(data_generator, dev):
            yield x, (y if y.dtype == torch.uint8 else y.to(torch.uint8)), z
    ms = variant == "mscmrseg"
    d4 = bool(getattr(args, "d4", False)) or (not ms and bool(getattr(args, "d4aux", False)))
    r = V.valid_model_with_one_dataset(seg_model, batches(), d4=d4, variant=variant, softmax=bool(getattr(args, "softmax", True)))
    seg_model.eval()                                   # the reference leaves the model in eval mode (:60)
    if not ms:
        r["vert_loss"] = r.pop("valid_vert_loss")
    return r


def valid_model(variant, args, seg_model, validA_iterator, validB_iterator, testB_generator) -> Dict[str, float]:
    """train_mscmrseg.py:102-139 / train_mmwhs.py:102-141: the three data sets in the reference's order, its result keys."""
    seg_model.eval()
    a = valid_model_with_one_dataset(variant, args, seg_model, validA_iterator)
    b = valid_model_with_one_dataset(variant, args, seg_model, validB_iterator)
    t = valid_model_with_one_dataset(variant, args, seg_model, testB_generator)
    if variant == "mscmrseg":
        return {"val_dice": a["dice"], "val_loss": a["loss"], "valid_vert_loss": a["valid_vert_loss"],
                "val_lge_dice": b["dice"], "val_lge_loss": b["loss"], "test_lge_dice": t["dice"], "test_lge_loss": t["loss"]}
    return {"val_dice": a["dice"], "val_loss": a["loss"], "val_vert_loss": a["vert_loss"], "val_lge_dice": b["dice"],
            "val_lge_loss": b["loss"], "val_lge_vert_loss": b["vert_loss"], "test_lge_dice": t["dice"],
            "test_lge_loss": t["loss"]}
