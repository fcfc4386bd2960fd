"""The 5-phase adversarial train step, restated on CPU (oracle; test-only).

Reference: train_mscmrseg.py:183-330 (``variant="mscmrseg"``) and
train_mmwhs.py:187-360 (``variant="mmwhs"``) -- the body of ``train_epoch``'s loop.
The scripts themselves cannot be imported (``kornia`` at line 5 / 1), so
``oracle/make_golden.py`` pins this restatement against a line-by-line re-typing of
that loop around the *imported* reference modules.

torch.optim is used for the parameter updates exactly as the reference does
(Adam(betas=(0.9,0.99)) for G, SGD(momentum, weight_decay=5e-4) for the D's:
train_mscmrseg.py:427-455; train_mmwhs.py:453-489).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import torch

from . import losses as L
from . import metrics as M
from .nets import (Params, SegCfg, disc_forward, is_trainable, pointnet_cls_forward,
                   seg_forward)


def _f(t) -> float:
    """python float of a (possibly graph-attached) scalar"""
    return float(t.detach()) if isinstance(t, torch.Tensor) else float(t)


@dataclass
class StepCfg:
    variant: str = "mscmrseg"        # or "mmwhs"
    d1: bool = True
    d2: bool = True
    d4: bool = True
    dr: float = 0.01                 # -dr   (train_mscmrseg.py:692)
    wp: float = 1.0                  # -wp   (:693)
    lr: float = 1e-3                 # -lr   (:682)
    d1lr: float = 2.5e-5
    d2lr: float = 2.5e-5
    d4lr: float = 2.5e-5
    d_momentum: float = 0.99         # mscmrseg :437; mmwhs default 0.95 (train_mmwhs.py:856-859)
    # mmwhs-only knobs
    softmax: bool = True             # -softmax (train_mmwhs.py:212)
    w1: float = 1.0
    w2: float = 1.0
    w4: float = 1.0
    etpls: bool = False
    Tetpls: bool = False
    d4aux: bool = False
    gen_sgd: bool = False            # -sgd: SGD(momentum .95, weight decay 5e-4) for the segmenter (train_mmwhs.py:453-459)
    disc_ext: bool = False           # UncertaintyDiscriminator(ext=...)
    pn_feature_transform: bool = False
    pn_ext: bool = False
    pn_drop: float = 0.0             # parity runs use 0 (reference default 0.3 draws from torch RNG)
    n_class: int = 4


def _leafify(p: Params) -> Params:
    out = {}
    for k, v in p.items():
        t = v.detach().clone()
        if is_trainable(k):
            t.requires_grad_(True)
        out[k] = t
    return out


def _trainables(p: Optional[Params]):
    return [] if p is None else [v for k, v in p.items() if is_trainable(k)]


def _freeze(p: Optional[Params], flag: bool):
    for t in _trainables(p):
        t.requires_grad_(not flag)


class OracleTrainer:
    """Holds G / D1 / D2 / D4 parameter dicts + optimisers and runs reference steps."""

    def __init__(self, seg_cfg: SegCfg, cfg: StepCfg, gen: Params, dis1: Optional[Params],
                 dis2: Optional[Params], dis4: Optional[Params]):
        self.seg_cfg, self.cfg = seg_cfg, cfg
        self.gen = _leafify(gen)
        self.dis1 = _leafify(dis1) if (cfg.d1 and dis1 is not None) else None
        self.dis2 = _leafify(dis2) if (cfg.d2 and dis2 is not None) else None
        self.dis4 = _leafify(dis4) if (cfg.d4 and dis4 is not None) else None
        if cfg.gen_sgd and cfg.variant == "mmwhs":
            self.opt_gen = torch.optim.SGD(_trainables(self.gen), lr=cfg.lr, momentum=.95, weight_decay=.0005)
        else:
            self.opt_gen = torch.optim.Adam(_trainables(self.gen), lr=cfg.lr, betas=(0.9, 0.99))
        mk = lambda p, lr: torch.optim.SGD(_trainables(p), lr=lr, momentum=cfg.d_momentum,
                                           weight_decay=0.0005)
        self.opt_d1 = mk(self.dis1, cfg.d1lr) if self.dis1 is not None else None
        self.opt_d2 = mk(self.dis2, cfg.d2lr) if self.dis2 is not None else None
        self.opt_d4 = mk(self.dis4, cfg.d4lr) if self.dis4 is not None else None

    # -- pieces ------------------------------------------------------------
    def _d_img(self, p, x):
        return disc_forward(p, x, ext=self.cfg.disc_ext)

    def _d_pts(self, verts):
        c = self.cfg
        return pointnet_cls_forward(self.dis4, verts.transpose(2, 1),
                                    feature_transform=c.pn_feature_transform, ext=c.pn_ext,
                                    drop=c.pn_drop, training=True)[0]

    def _pred(self, logits):
        c = self.cfg
        if c.variant == "mscmrseg":
            return None
        return torch.softmax(logits, 1) if c.softmax else torch.sigmoid(logits)

    def _entropy(self, logits):
        c = self.cfg
        if c.variant == "mscmrseg":
            return L.entropy_map(logits, "sigmoid", normalise=False)
        return L.entropy_map(logits, "softmax" if c.softmax else "sigmoid", normalise=True)

    # -- one iteration of the train_epoch loop -------------------------------
    def step(self, img_a: np.ndarray, mask_a: np.ndarray, vert_a: Optional[np.ndarray],
             img_b: np.ndarray, vert_b: Optional[np.ndarray], keep: bool = False) -> Dict[str, float]:
        c, out = self.cfg, {}
        ms = c.variant == "mscmrseg"
        for o in (self.opt_gen, self.opt_d1, self.opt_d2, self.opt_d4):
            if o is not None:
                o.zero_grad()
        for p in (self.dis1, self.dis2, self.dis4):
            _freeze(p, True)
        _freeze(self.gen, False)

        # 1. supervised pass on the source batch (train_mscmrseg.py:200-213)
        xa = torch.as_tensor(img_a, dtype=torch.float32)
        ya = torch.as_tensor(mask_a, dtype=torch.float32)
        o_s, vert_s = seg_forward(self.gen, xa, self.seg_cfg, training=True)
        if ms or not c.softmax:
            l_bce, l_jac = L.seg_loss_sigmoid(o_s, ya)
        else:
            l_bce, l_jac = L.seg_loss_softmax(o_s, ya)
        l_pt = 0.0
        if c.d4 or (c.d4aux and not ms):
            l_pt = L.batch_nn_loss(vert_s, torch.as_tensor(vert_a, dtype=torch.float32))
            out["ver_s_loss"] = float(l_pt.detach())
        loss1 = l_bce + l_jac + c.wp * l_pt
        emap_s = None
        if not ms:
            emap_s = self._entropy(o_s)                                  # train_mmwhs.py:224
            ent_s = torch.mean(torch.sum(emap_s, dim=1))
            out["entropy_loss"] = _f(ent_s)
            if c.d2 and c.etpls:
                loss1 = loss1 + ent_s
        out["seg_loss"] = _f(l_bce + l_jac)
        out["loss_bce"], out["loss_jac"] = _f(l_bce), _f(l_jac)
        loss1.backward()
        hard = M.soft_to_hard_pred(o_s.detach().numpy(), 1)             # :215-216
        out["seg_dice"] = M.dice_coef_multilabel(np.asarray(mask_a), hard, c.n_class)
        if keep:
            self.kept = {"grad_seg": {k: v.grad.clone() for k, v in self.gen.items()
                                      if is_trainable(k) and v.grad is not None},
                         "oS": o_s.detach().clone(),
                         "vertS": None if vert_s is None else vert_s.detach().clone()}

        # 2. adversarial pass on the target batch (:218-247)
        xb = torch.as_tensor(img_b, dtype=torch.float32)
        o_t, vert_t = seg_forward(self.gen, xb, self.seg_cfg, training=True)
        pred_s, pred_t = self._pred(o_s), self._pred(o_t)
        emap_t = self._entropy(o_t) if (c.d2 or not ms) else None
        adv = 0.0
        if not ms:
            ent_t = torch.mean(torch.sum(emap_t, dim=1))
            out["entropy_loss_T"] = _f(ent_t)
            if c.Tetpls:
                adv = adv + ent_t
        a2 = a4 = a1 = 0.0
        if c.d2:
            a2 = c.dr * L.bce_logits_const(self._d_img(self.dis2, emap_t), 1.0)
        if c.d4 or (c.d4aux and not ms):
            out["ver_t_loss"] = float(L.batch_nn_loss(vert_t, torch.as_tensor(vert_b, dtype=torch.float32)))
        if c.d4:
            a4 = c.dr * L.bce_logits_const(self._d_pts(vert_t), 1.0)
        if c.d1:
            a1 = c.dr * L.bce_logits_const(self._d_img(self.dis1, o_t if ms else pred_t), 1.0)
        if ms:
            adv = a2 + a4 + a1
        else:
            adv = adv + c.w2 * a2 + c.w4 * a4 + c.w1 * a1
        out["adv_loss"] = _f(adv)
        out["adv2"], out["adv4"], out["adv1"] = _f(a2), _f(a4), _f(a1)
        if torch.is_tensor(adv):
            adv.backward()
        if keep:
            self.kept["grad_total"] = {k: v.grad.clone() for k, v in self.gen.items()
                                       if is_trainable(k) and v.grad is not None}
            self.kept["oT"] = o_t.detach().clone()
            self.kept["vertT"] = None if vert_t is None else vert_t.detach().clone()
            self.kept["emapT"] = None if emap_t is None else emap_t.detach().clone()
        self.opt_gen.step()

        # 3./4. discriminators on source-as-1, target-as-0 (:250-322)
        for p in (self.dis1, self.dis2, self.dis4):
            _freeze(p, False)
        _freeze(self.gen, True)
        o_s, o_t = o_s.detach(), o_t.detach()
        if ms:
            in1_s, in1_t = o_s, o_t
            emap_s = self._entropy(o_s) if c.d2 else None
        else:
            in1_s, in1_t = pred_s.detach(), pred_t.detach()
            emap_s = emap_s.detach()
        emap_t = None if emap_t is None else emap_t.detach()

        def d_phase(tag, label, e, i1, v):
            if c.d2:
                d = self._d_img(self.dis2, e)
                l = L.bce_logits_const(d, label); l.backward()
                out["d2_loss_" + tag] = _f(l)
                out["dis2_acc_" + tag] = M.disc_accuracy(d.detach().numpy(), label == 1.0)
            if c.d1:
                d = self._d_img(self.dis1, i1)
                l = L.bce_logits_const(d, label); l.backward()
                out["d1_loss_" + tag] = _f(l)
                out["dis1_acc_" + tag] = M.disc_accuracy(d.detach().numpy(), label == 1.0)
            if c.d4:
                d = self._d_pts(v.detach())
                l = L.bce_logits_const(d, label); l.backward()
                out["d4_loss_" + tag] = _f(l)
                out["dis4_acc_" + tag] = M.disc_accuracy(d.detach().numpy(), label == 1.0)

        if c.d1 or c.d2 or c.d4:
            d_phase("src", 1.0, emap_s, in1_s, vert_s)
            if keep:      # (the two passes' gradients largely cancel at initialisation: tests scale errors by the parts)
                for nm, p in (("grad_d1_src", self.dis1), ("grad_d2_src", self.dis2), ("grad_d4_src", self.dis4)):
                    if p is not None:
                        self.kept[nm] = {k: v.grad.clone() for k, v in p.items()
                                         if is_trainable(k) and v.grad is not None}
            d_phase("tgt", 0.0, emap_t, in1_t, vert_t)
            if keep:
                for nm, p in (("grad_d1", self.dis1), ("grad_d2", self.dis2), ("grad_d4", self.dis4)):
                    if p is not None:
                        self.kept[nm] = {k: v.grad.clone() for k, v in p.items()
                                         if is_trainable(k) and v.grad is not None}
            # 5. update (:325-330)
            for o in (self.opt_d1, self.opt_d2, self.opt_d4):
                if o is not None:
                    o.step()
        return out
