"""The 5-phase adversarial train step on the HIP kernels.

Host-side mirror of the body of ``train_epoch``'s loop (train_mscmrseg.py:183-330;
train_mmwhs.py:187-360): same phases, same losses, same freeze/unfreeze schedule, same optimiser
settings -- with three MI355X-first changes that do not alter the arithmetic:

* the per-step host metrics (Dice, discriminator accuracies, ``.item()`` calls at
  train_mscmrseg.py:207-216,270-322) are computed on the device and returned as 0-dim tensors, so a
  step never synchronises with the host;
* gradients live in one flat buffer per network; with ``torch.distributed`` initialised each rank
  runs the step on its shard of the batch and the flat buffers are all-reduced over RCCL before the
  optimiser kernels (G after phase 2, the D's after phase 4);
* loss weights (``dr``, ``wp``, ``w1/w2/w4``) are folded into the backward seeds instead of being
  multiplied in as separate scalar ops.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import kernels as K
from .optim import FusedAdam, FusedSGD, _FlatOptimizer as _FlatOpt, _collectives_on
from .utils import loss as L


@dataclass
class TrainCfg:
    variant: str = "mscmrseg"        # "mscmrseg" | "mmwhs"
    d1: bool = True
    d2: bool = True
    d4: bool = True
    dr: float = 0.01                 # -dr  (train_mscmrseg.py:692)
    wp: float = 1.0                  # -wp  (:693)
    lr: float = 1e-3
    d1lr: float = 2.5e-5
    d2lr: float = 2.5e-5
    d4lr: float = 2.5e-5
    d_momentum: float = 0.99
    softmax: bool = True             # mmwhs -softmax
    w1: float = 1.0
    w2: float = 1.0
    w4: float = 1.0
    n_class: int = 4
    # optional branches of the MM-WHS loop (ignored by the mscmrseg variant, which has none of them)
    etpls: bool = False              # -etpls: source entropy mean added to the supervised loss, with -d2 (train_mmwhs.py:227-230)
    Tetpls: bool = False             # -Tetpls: target entropy mean added to the adversarial loss (:245-247)
    d4aux: bool = False              # -d4aux: point head trained (and its target loss reported) without d4 (:220,248,256)
    gen_sgd: bool = False            # -sgd: SGD(momentum .95, weight decay 5e-4) for the segmenter (:453-459)


class AdversarialTrainer:
    def __init__(self, model_gen, model_dis1=None, model_dis2=None, model_dis4=None, cfg: Optional[TrainCfg] = None,
                 process_group=None):
        self.cfg = cfg or TrainCfg()
        c = self.cfg
        self.gen = model_gen
        self.dis1 = model_dis1 if c.d1 else None
        self.dis2 = model_dis2 if c.d2 else None
        self.dis4 = model_dis4 if c.d4 else None
        self.group = process_group
        if c.gen_sgd and c.variant == "mmwhs":
            # torch.optim.SGD skips parameters whose .grad is None (no weight decay either): the reference's unused
            # encoder.conv1_1 always, and the point head when no loss is attached to it
            skip = ["encoder.conv1_1."] + ([] if (c.d4 or c.d4aux) else ["pointNet."])
            self.opt_gen = FusedSGD(self.gen, lr=c.lr, momentum=0.95, weight_decay=0.0005, skip_prefixes=tuple(skip))
        else:
            # (exported torch-format state omits what the reference never updates: its unused encoder.conv1_1, and the point
            # head when no loss is attached to it)
            skip = ["encoder.conv1_1."] + ([] if (c.d4 or c.d4aux or not hasattr(self.gen, "pointNet")) else ["pointNet."])
            self.opt_gen = FusedAdam(self.gen, lr=c.lr, betas=(0.9, 0.99), skip_prefixes=tuple(skip))
        mk = lambda m, lr: FusedSGD(m, lr=lr, momentum=c.d_momentum, weight_decay=0.0005)
        self.opt_d1 = mk(self.dis1, c.d1lr) if self.dis1 is not None else None
        self.opt_d2 = mk(self.dis2, c.d2lr) if self.dis2 is not None else None
        self.opt_d4 = mk(self.dis4, c.d4lr) if self.dis4 is not None else None
        dev = next(self.gen.parameters()).device
        self._one = torch.ones((), dtype=torch.float32, device=dev)
        self._wp = torch.full((), float(c.wp), dtype=torch.float32, device=dev)
        self.last = {}
        self.d_streams = os.environ.get("PCUDA_DSTREAMS", "1") != "0"   # discriminator passes on concurrent streams
        self._streams = {}
        self.broadcast_parameters()
        self.d_overlap = os.environ.get("PCUDA_DOVERLAP", "1") != "0"   # discriminator update under the G backward
        self.early_fwd2 = os.environ.get("PCUDA_EARLY2", "1") != "0"    # target forward ahead of the source backward
        self.d_batch = os.environ.get("PCUDA_DBATCH", "1") != "0"       # d1 / d2: source + target as one batch
        self.bucketed = os.environ.get("PCUDA_BUCKET", "1") != "0"      # data parallel: first all-reduce bucket from inside the backward pass
        # d1 / d2 update on the target batch replays the frozen adversarial pass of phase 2 (same weights, same input
        # values -> the same activations) instead of running the network forward on it again
        self.d_reuse = os.environ.get("PCUDA_DREUSE", "1") != "0"
        # ... with the source batch's activations written in front of the cached target ones: one backward pass over 2B
        self.d_joint = os.environ.get("PCUDA_DJOINT", "1") != "0"
        self._segment = None      # "compute": step() leaves out the collectives and the optimiser steps (step_graphed)
        self._timeline = os.environ.get("PCUDA_TIMELINE", "0") == "1"     # (read per trainer, not at import time)
        # PCUDA_EXP_SKIP_DUPDATE=1: the discriminators' update passes (phases 3-4) are left out -- a timing experiment
        # (profiles/r06_experiment_dstream_cost.txt); bench.py marks such a line "experiment"
        self._exp_skip_dupdate = os.environ.get("PCUDA_EXP_SKIP_DUPDATE", "0") == "1"

    def _side_streams(self, names):
        """One side stream PER DISCRIMINATOR, keyed by its name: a network's frozen pass (phase 2), its input-gradient
        pass and its update passes (phases 3-4) all run on the same stream, so its weight repack, its BatchNorm
        running-statistic updates and its gradient buffer are ordered by the stream itself."""
        for nm in names:
            if nm not in self._streams:
                self._streams[nm] = torch.cuda.Stream()
        return [self._streams[nm] for nm in names]

    def broadcast_parameters(self, src: int = 0):
        """Data parallel: every replica starts from rank ``src``'s parameters, BatchNorm running statistics and
        optimiser state (one broadcast per flat buffer).  No-op without an initialised group of more than one rank."""
        if not _collectives_on(self.group):
            return
        import torch.distributed as dist
        for opt in [self.opt_gen] + self._d_opts():
            bufs = [opt.p] + [t for t in (getattr(opt, "m", None), getattr(opt, "v", None), getattr(opt, "buf", None),
                                          getattr(opt, "step_t", None)) if t is not None]
            bufs += [b for b in opt.module.buffers() if b.numel() > 0]
            for t in bufs:
                dist.broadcast(t, src=src, group=self.group)
            # the host-side step count decides whether SGD's next step INITIALISES its momentum buffer (optim.py): a
            # fresh replica next to a resumed rank 0 would overwrite the buffer it has just received
            steps = torch.tensor([opt.steps], dtype=torch.int64, device=opt.p.device)
            dist.broadcast(steps, src=src, group=self.group)
            opt.steps = int(steps.item())
            # the flat parameter buffer changed behind torch's back (no ``_version`` bump): packed convolution weights
            # cached by an earlier forward pass are stale on every rank but ``src``
            opt.module._wgen = getattr(opt.module, "_wgen", 0) + 1
            K.repack_owner(opt.module)

    def _dis(self):
        return [m for m in (self.dis1, self.dis2, self.dis4) if m is not None]

    def _replays(self, m):
        return self.d_reuse and bool(getattr(m, "can_replay", False))

    def _d_opts(self):
        return [o for o in (self.opt_d1, self.opt_d2, self.opt_d4) if o is not None]

    def train(self):
        self.gen.train()
        for m in self._dis():
            m.train()

    # ------------------------------------------------------------------ one loop iteration
    # PCUDA_TIMELINE=1 (diagnostics, scripts/step_timeline.py): timed events on the streams at the schedule's joints -- where
    # each stream is when, without a profiler in the host's way.  Off: ``_mark`` returns at once.
    _MARKS_KEPT = 8 * 32      # the last ~8 steps' marks: a training run with the variable left set does not grow without bound

    def _mark(self, label, stream=None):
        if not self._timeline:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream if stream is not None else torch.cuda.current_stream())
        marks = self.__dict__.setdefault("_marks", [])
        marks.append((label, ev))
        if len(marks) > 2 * self._MARKS_KEPT:
            del marks[:len(marks) - self._MARKS_KEPT]

    def step(self, img_a, mask_a_u8, vert_a, img_b, vert_b, drop_mask=None, keep=False) -> Dict[str, torch.Tensor]:
        c, out = self.cfg, {}
        self._mark("step")
        ms = c.variant == "mscmrseg"
        mode = "sigmoid" if (ms or not c.softmax) else "softmax"
        one = self._one
        for o in [self.opt_gen] + self._d_opts():
            o.zero_grad()
        for m in self._dis():
            m.requires_grad_(False)
        self.gen.requires_grad_(True)

        # 1. supervised pass on the source batch (train_mscmrseg.py:200-213)
        o_s, _, vert_s = self.gen(img_a)
        self._mark("fwd_src.end")
        l_main, l_jac = L.seg_loss(o_s, mask_a_u8, mode)
        seeds_t, seeds_g = [l_main, l_jac], [one, one]
        aux = c.d4aux and not ms
        pre_s = None
        if not ms:
            # entropy map / probabilities of the source batch (train_mmwhs.py:223-226): the map's mean is a per-step
            # metric, and with -d2 -etpls a term of the supervised loss (:227-230); the maps themselves are the
            # discriminators' source inputs of phases 3-4 (taken detached)
            with_grad = c.d2 and c.etpls
            ent_s, pred_s, m_s = L.entropy_map(o_s if with_grad else o_s.detach(), mode, True, want_prob=True, want_mean=True)
            out["entropy_loss"] = m_s.detach()
            if with_grad:
                seeds_t.append(m_s)
                seeds_g.append(one)
            pre_s = (ent_s.detach(), pred_s.detach())
        if c.d4 or aux:
            l_pt = L.batch_NN_loss(vert_s, vert_a)
            out["ver_s_loss"] = l_pt.detach()
            seeds_t.append(l_pt)
            seeds_g.append(self._wp)
        out["loss_bce"], out["loss_jac"] = l_main.detach(), l_jac.detach()
        # 2. (forward half) adversarial pass on the target batch (:218-241).  The reference runs the source batch's
        # backward pass first; nothing in it feeds the target forward (weights change only in the optimisers, BatchNorm's
        # running statistics are updated by the forward passes, in this same order), so the forward goes first and
        # the frozen discriminators' passes over its outputs -- a few-tile island between the two halves of the
        # segmenter's work -- run on their streams UNDER the source batch's backward pass.
        early = self.early_fwd2
        if early:
            o_t, vert_t, prep, ev, adv_t, adv_g = self._phase2_forward(img_b, vert_b, drop_mask, out, o_s, pre_s)
        torch.autograd.backward(seeds_t, seeds_g)
        self._mark("bwd_src.end")
        out["seg_dice"] = K.dice_metric(o_s.detach(), mask_a_u8)      # :215-216, on the device
        if keep:
            self.last = {"oS": o_s.detach(), "vertS": None if vert_s is None else vert_s.detach(),
                         "grad_seg": self.opt_gen.g.clone()}
        if not early:
            o_t, vert_t, prep, ev, adv_t, adv_g = self._phase2_forward(img_b, vert_b, drop_mask, out, o_s, pre_s)
        # 2. (backward half, :242-247).  Data-parallel: the all-reduce of every segmenter gradient except the encoder's
        # (92 % of the 76 MB) starts from inside the backward pass, as soon as those kernels are launched, and runs on
        # RCCL's stream under the encoder's backward kernels; the encoder's slice follows after the pass.
        g_works, split = [], 0
        if adv_t:
            eng = getattr(self.gen, "_engine", None)
            if eng is not None and self.bucketed and _collectives_on(self.group) and self._segment is None:
                split = self.opt_gen.split_after("encoder.")
                if split:
                    eng.after_deep_grads = lambda: g_works.append(
                        self.opt_gen.all_reduce_grads_async(self.group, lo=split)[0])
            try:
                torch.autograd.backward(adv_t, adv_g)
                self._mark("bwd_adv.end")
            finally:
                if eng is not None:
                    eng.after_deep_grads = None
            if not g_works:
                split = 0          # the hook did not fire: one all-reduce of the whole buffer below
        if keep:
            self.last.update({"oT": o_t.detach(), "vertT": None if vert_t is None else vert_t.detach(),
                              "grad_total": self.opt_gen.g.clone()})
        self._phase345(o_s, vert_s, vert_t, prep, ev, drop_mask, out, keep, g_works, split)
        return out

    def _phase2_forward(self, img_b, vert_b, drop_mask, out, o_s, pre_s=None):
        c = self.cfg
        ms = c.variant == "mscmrseg"
        mode = "sigmoid" if (ms or not c.softmax) else "softmax"
        one = self._one
        o_t, _, vert_t = self.gen(img_b)
        self._mark("fwd_tgt.end")
        norm = not ms
        pred_t = ent_t = tap_t = None
        if ms:
            if c.d2 and c.d1:
                tap_t, ent_t = L.logits_and_entropy(o_t, "sigmoid", False)
            elif c.d2:
                ent_t = L.entropy_map(o_t, "sigmoid", False)
            else:
                tap_t = o_t
        else:
            ent_t, pred_t, m_t = L.entropy_map(o_t, mode, True, want_prob=True, want_mean=True)
            out["entropy_loss_T"] = m_t.detach()                    # train_mmwhs.py:243-244
        adv_t, adv_g = [], []
        if not ms and c.Tetpls:                                     # :245-247 (weight 1, not dr-scaled)
            adv_t.append(m_t); adv_g.append(one)
        # the frozen discriminators' forward passes (and, through autograd's per-node streams, their input-gradient
        # passes) run next to each other like phases 3-4 below
        heads = []
        # (PCUDA_DJOINT=1: with room for the source batch in front, the update then walks source + target as ONE batch)
        run_d = lambda m, x: m.forward_cached(x, room=1 if self.d_joint else 0) if self._replays(m) else m(x)
        if c.d2:
            heads.append(("adv2", lambda: run_d(self.dis2, ent_t), c.dr * (1.0 if ms else c.w2)))
        if c.d4 or (c.d4aux and not ms):
            out["ver_t_loss"] = L.batch_NN_loss(vert_t.detach(), vert_b)
        if c.d4:
            heads.append(("adv4", lambda: self.dis4(vert_t.transpose(2, 1), drop_mask)[0], c.dr * (1.0 if ms else c.w4)))
        if c.d1:
            heads.append(("adv1", lambda: run_d(self.dis1, tap_t if ms else pred_t), c.dr * (1.0 if ms else c.w1)))
        cur = torch.cuda.current_stream()
        side = (self._side_streams(["d" + nm[-1] for nm, _, _ in heads]) if (self.d_streams and len(heads) > 1)
                else [None] * len(heads))
        for (nm, fwd, wgt), st in zip(heads, side):
            if st is not None:
                st.wait_stream(cur)
            with torch.cuda.stream(st if st is not None else cur):
                self._mark(nm + ".fwd.start")
                l = L.bce_logits_const(fwd(), 1.0, weight=wgt)
                self._mark(nm + ".fwd.end")
            adv_t.append(l); adv_g.append(one); out[nm] = l.detach()
        # (no join here: autograd orders the backward nodes across streams, and the caller's stream waits for the
        # discriminator streams at the end of phase 4)
        # Inputs of the discriminator update (phases 3-4): detached outputs of the two forward passes.  They exist
        # now, and the update touches nothing the adversarial backward pass below reads or writes (frozen D weights,
        # separate gradient buffers), so its streams fork HERE: the discriminators' few-tile kernels then run under the
        # segmenter's backward pass instead of after it.
        prep = ev = None
        if self._dis():
            o_s_d, o_t_d = o_s.detach(), o_t.detach()
            if ms:
                ent_s = L.entropy_map(o_s_d, "sigmoid", False) if c.d2 else None
                in1_s, in1_t = o_s_d, o_t_d
            else:
                ent_s, pred_s = pre_s if pre_s is not None else L.entropy_map(o_s_d, mode, True, want_prob=True)
                in1_s, in1_t = pred_s, (None if pred_t is None else pred_t.detach())
            prep = (ent_s, None if ent_t is None else ent_t.detach(), in1_s, in1_t)
            if self.d_overlap and self.d_streams:
                ev = torch.cuda.Event()
                ev.record(cur)
        return o_t, vert_t, prep, ev, adv_t, adv_g

    def _phase345(self, o_s, vert_s, vert_t, prep, ev, drop_mask, out, keep, g_works, split):
        c = self.cfg
        # The segmenter's update needs the all-reduced gradient, but nothing in phases 3-5 reads the segmenter's
        # parameters or gradient buffer (the discriminators train on the detached outputs of phases 1-2): the rest of
        # the all-reduce (the encoder's slice, or everything if no bucket went out during the backward pass) starts
        # now, and Adam is applied after the discriminator passes, which hide it.  Single process: no collective, same
        # order of arithmetic as the reference.
        compute_only = self._segment == "compute"      # step_graphed with collectives: gradients only, see there
        if compute_only:
            g_work, g_scale = None, 1.0
        else:
            w_rest, g_scale = self.opt_gen.all_reduce_grads_async(self.group, lo=0, hi=(split or None))
            g_work = [w for w in g_works + [w_rest] if w is not None] or None
            if g_work is None:
                self.opt_gen.step(g_scale)

        # 3./4. discriminators: source batch as 1, target batch as 0 (:250-322)
        if self._dis():
            for m in self._dis():
                m.requires_grad_(True)
            self.gen.requires_grad_(False)
            ent_s, ent_t_d, in1_s, in1_t = prep
            # The discriminators are independent networks with their own gradient buffers: each runs its source and
            # target passes on its own HIP stream (forked from / joined to the caller's), so that the point-cloud
            # discriminator's many small kernels and the tails of the persistent convolution kernels fill each other's
            # gaps.  Same arithmetic per network as the reference's interleaved order (:250-322).
            passes = []
            if c.d2:
                passes.append(("d2", "dis2", lambda e, i1, v: self.dis2(e)))
            if c.d1:
                passes.append(("d1", "dis1", lambda e, i1, v: self.dis1(i1)))
            if c.d4:
                passes.append(("d4", "dis4", lambda e, i1, v: self.dis4(v.detach().transpose(2, 1), drop_mask)[0]))
            if self._exp_skip_dupdate:     # MEASUREMENT ONLY (what the update passes cost the step): not the reference's step
                passes = []
            main = torch.cuda.current_stream()
            side = (self._side_streams([nm for nm, _, _ in passes]) if (self.d_streams and len(passes) > 1)
                    else [None] * len(passes))
            d_works = {}
            for (nm, hit, fwd), st in zip(passes, side):
                if st is not None:
                    if ev is not None:
                        st.wait_event(ev)
                    else:
                        st.wait_stream(main)
                with torch.cuda.stream(st if st is not None else main):
                    self._mark(nm + ".update.start")
                    dnet = getattr(self, "dis" + nm[1])
                    if nm != "d4" and self._replays(dnet) and getattr(dnet, "_cache", None) is not None:
                        # source batch: a forward pass; target batch: the activations of phase 2's frozen pass (the
                        # weights have not moved and the input values are the same).  Both backward passes add into
                        # the network's gradient buffer, as the reference's two backward calls do (:262-263,:296-297).
                        ls = []
                        if dnet._cache["full"] is not None:
                            # the source batch's activations go in front of the target batch's in the cached pass's
                            # buffers: one backward pass over 2B samples, the arithmetic of the one-batch form below
                            dnet.forward_fill(ent_s if nm == "d2" else in1_s)
                            d, bsz = dnet.replay(), o_s.shape[0]
                            parts = (("src", 1.0, d[:bsz]), ("tgt", 0.0, d[bsz:]))
                        else:
                            parts = (("src", 1.0, fwd(ent_s, in1_s, None)), ("tgt", 0.0, dnet.replay()))
                        for tag, label, part in parts:
                            l, acc = L.bce_logits_const(part, label, 1.0, want_acc=True)
                            ls.append(l)
                            out[nm + "_loss_" + tag], out[hit + "_hit_" + tag] = l.detach(), acc
                        torch.autograd.backward(ls, [self._one, self._one])
                        dnet.drop_cache()
                    elif nm != "d4" and self.d_batch:
                        # d1 / d2 have no batch statistics: the source and the target batch go through the network as
                        # ONE batch of 2B samples (twice the tiles per launch on the 17x17 / 9x9 maps, half the launches);
                        # the two mean losses of :262-263,:296-297 are taken over the halves of the output, and their
                        # gradients add up in the weight-gradient kernels exactly as the two backward calls' do
                        bsz = o_s.shape[0]
                        d = fwd(torch.cat([ent_s, ent_t_d], 0) if nm == "d2" else None,
                                torch.cat([in1_s, in1_t], 0) if nm == "d1" else None, None)
                        ls = []
                        for tag, label, part in (("src", 1.0, d[:bsz]), ("tgt", 0.0, d[bsz:])):
                            l, acc = L.bce_logits_const(part, label, 1.0, want_acc=True)
                            ls.append(l)
                            out[nm + "_loss_" + tag], out[hit + "_hit_" + tag] = l.detach(), acc
                        torch.autograd.backward(ls, [self._one, self._one])
                    else:
                        for tag, label, e, i1, v in (("src", 1.0, ent_s, in1_s, vert_s),
                                                     ("tgt", 0.0, ent_t_d, in1_t, vert_t)):
                            l, acc = L.bce_logits_const(fwd(e, i1, v), label, 1.0, want_acc=True)
                            l.backward()
                            out[nm + "_loss_" + tag], out[hit + "_hit_" + tag] = l.detach(), acc
                    # data-parallel: this network's all-reduce starts behind its own passes (on its stream), under the
                    # other discriminators' kernels
                    d_works[nm] = (None, 1.0) if compute_only else getattr(self, "opt_" + nm).all_reduce_grads_async(self.group)
                    self._mark(nm + ".update.end")
            for st in side:
                if st is not None:
                    main.wait_stream(st)
            if keep:      # (single-process diagnostics: with collectives on these would be the summed gradients)
                for nm, o in (("grad_d1", self.opt_d1), ("grad_d2", self.opt_d2), ("grad_d4", self.opt_d4)):
                    if o is not None:
                        o.finish_all_reduce(d_works[nm[5:]][0])
                        self.last[nm] = o.g.clone()
            # 5. update (:325-330)
            if g_work is not None:
                self.opt_gen.finish_all_reduce(g_work)
                if keep:
                    self.last["grad_reduced"] = self.opt_gen.g.clone()     # (diagnostics: the summed gradient)
                self.opt_gen.step(g_scale)
                g_work = None
            for nm in ("d1", "d2", "d4"):
                o = getattr(self, "opt_" + nm)
                if o is not None and not compute_only and nm in d_works:   # (absent only under PCUDA_EXP_SKIP_DUPDATE)
                    work, scale = d_works[nm]
                    o.finish_all_reduce(work)
                    o.step(scale)
            self._mark("opt_d.end")
        if g_work is not None:      # no discriminator configured
            self.opt_gen.finish_all_reduce(g_work)
            self.opt_gen.step(g_scale)

    # ------------------------------------------------------------------ the same iteration as one hipGraph
    def apply_updates(self, grad_scale: float = 1.0):
        """phase 5 on its own (:247,:325-330): every optimiser's step on the gradients as they stand (x grad_scale)"""
        for o in [self.opt_gen] + self._d_opts():
            o.step(grad_scale)

    def step_graphed(self, img_a, mask_a_u8, vert_a, img_b, vert_b) -> Dict[str, torch.Tensor]:
        """``step`` replayed from captured hipGraphs.

        The ~850 kernel launches of a step cost the host ~24 ms to issue from Python (bench.py: host_issue_ms_per_step);
        one process that is enough to stay ahead of a 47 ms GPU step, eight ranks on one host share its cores.  A
        replay is one launch.  Every call advances exactly one step: the first two run eagerly (every lazily created
        buffer, packed-weight cache and kernel attribute exists afterwards), the third captures, later calls copy the
        batch into the captured input buffers and replay.

        Data parallel: a collective cannot sit inside the capture here (this torch's RCCL watchdog queries the
        collective's event, which HIP forbids for an event recorded on a capturing stream: the process aborts --
        measured, round 4).  The step is therefore captured in TWO graphs around the exchange: graph A = phases 1-4
        (every forward / backward pass, gradients complete in the flat buffers; ``_segment = "compute"``), then the
        all-reduces of the four flat buffers issued eagerly -- four calls, any backend -- then graph B = phase 5 (the
        optimiser kernels with the 1 / world mean folded in, and the weight repacks).  What this gives up against the
        eager schedule is the overlap of the segmenter's all-reduce with the discriminator passes (``comm_exposed_ms``
        of bench.py prices it); the arithmetic is the eager step's, kernel for kernel.

        Learning rates are baked into the capture (re-capture after changing them: ``self._graph = None``); Adam's step
        count lives on the device.  Falls back to ``step`` if capture is not possible."""
        coll = _collectives_on(self.group)
        batch = (img_a, mask_a_u8, vert_a, img_b, vert_b)
        if getattr(self, "_graph", None) is None:
            self._gcalls = getattr(self, "_gcalls", 0) + 1
            if getattr(self, "_graph_failed", False) or self._gcalls <= 2:
                return self.step(*batch)       # calls 1 and 2: eager (every lazily created buffer exists afterwards)
            try:
                torch.cuda.synchronize()
                self._gin = tuple(t.clone() for t in batch)
                graph = torch.cuda.CUDAGraph()
                self._segment = "compute" if coll else None
                try:
                    # (thread-local capture mode: in a process group, RCCL's watchdog thread polls the events of earlier
                    # collectives while this thread records; in the default global mode its calls fail or invalidate the capture)
                    with torch.cuda.graph(graph, capture_error_mode="thread_local"):   # records only: nothing executes
                        self._gout = self.step(*self._gin)
                finally:
                    self._segment = None
                self._graph_b = None
                if coll:
                    import torch.distributed as dist
                    scale = 1.0 / dist.get_world_size(self.group)
                    gb = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gb, capture_error_mode="thread_local"):
                        self.apply_updates(scale)
                    self._graph_b = gb
                self._graph = graph
            except Exception as e:   # noqa: BLE001 -- any capture failure means: stay eager
                import warnings
                warnings.warn("hipGraph capture of the train step failed (%s: %s); running eagerly" % (type(e).__name__, e))
                self._graph, self._graph_b, self._graph_failed = None, None, True
                torch.cuda.synchronize()
                return self.step(*batch)
        for dst, src in zip(self._gin, batch):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self._graph.replay()
        if self._graph_b is not None:
            works = [o.all_reduce_grads_async(self.group)[0] for o in [self.opt_gen] + self._d_opts()]
            _FlatOpt.finish_all_reduce(works)
            self._graph_b.replay()
        return self._gout

    @staticmethod
    def to_host(out: Dict[str, torch.Tensor], cfg: "TrainCfg") -> Dict[str, float]:
        """One synchronisation for a whole step (or epoch): device scalars -> the reference's metrics."""
        h = {k: float(v) for k, v in out.items()}
        h["seg_loss"] = h["loss_bce"] + h["loss_jac"]
        ms = cfg.variant == "mscmrseg"
        adv = 0.0
        for k, w in (("adv2", 1.0 if ms else cfg.w2), ("adv4", 1.0 if ms else cfg.w4), ("adv1", 1.0 if ms else cfg.w1)):
            if k in h:
                adv += cfg.dr * w * h[k]
        if not ms and cfg.Tetpls and "entropy_loss_T" in h:
            adv += h["entropy_loss_T"]
        h["adv_loss"] = adv
        for d in ("dis1", "dis2", "dis4"):
            if d + "_hit_src" in h:
                h[d + "_acc1"] = h[d + "_hit_src"]             # mean(sigmoid(D) >= .5) on source
                h[d + "_acc2"] = 1.0 - h[d + "_hit_tgt"]       # 1 - mean(...) on target
        return h
