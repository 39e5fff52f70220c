"""Flat parameter / gradient / Adam-state buffers for one half of the model (encoder or decoder).

Parameters are re-pointed (``p.data``) into one contiguous fp32 buffer and their ``.grad`` into
a second one, so that the global gradient norm, the clip scaling, the Adam update and the
data-parallel all-reduce are each ONE kernel / ONE collective per half instead of one per
tensor (20 M parameters in ~100 tensors at the 64x64 configuration).  The nn.Parameter objects
themselves are untouched, so optimizers and state_dicts that reference them stay valid.
"""
import weakref

import torch

from .abi import call, lib, ptr, stream
from .functional import bump_weight_epoch, register_pack_group

F32 = torch.float32


class FlatGroup:
    def __init__(self, params, inherit=None):
        """``inherit``: the group these parameters lived in before they were moved (``model.to()``, ``.float()``,
        ``load_state_dict(assign=True)`` ... after the first step): its Adam moments and step count carry over, so
        losing ownership never silently restarts the optimiser."""
        self.params = [p for p in params]
        assert self.params, "empty parameter group"
        dev = self.params[0].device
        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.numel = total
        self.flat_p = torch.zeros(total, dtype=F32, device=dev)
        self.flat_g = torch.zeros(total, dtype=F32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=F32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=F32, device=dev)
        self.step = 0                                      # host mirror (not advanced by graph replays)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)   # authoritative step count
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                n = p.numel()
                self.flat_p[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
        self._ws = torch.empty(lib.itcv_sumsq_workspace(total), dtype=torch.uint8, device=dev)
        self._opt, self._parent = None, inherit
        register_pack_group(self.params)       # their packed conv operands are refreshed by one launch per direction
        self.attach_grads()
        if inherit is not None:
            self._inherit(inherit)

    def _inherit(self, old):
        where = {id(p): (o, p.numel()) for p, o in zip(old.params, old.offsets)}
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                src = where.get(id(p))
                if src is not None and src[1] == p.numel():
                    n = p.numel()
                    self.exp_avg[o:o + n].copy_(old.exp_avg[src[0]:src[0] + n])
                    self.exp_avg_sq[o:o + n].copy_(old.exp_avg_sq[src[0]:src[0] + n])
                    # the (possibly stale) gradients too: the clip norm runs over them (solvers/intro.py:113-115)
                    self.flat_g[o:o + n].copy_(old.flat_g[src[0]:src[0] + n])
            self.step_dev.copy_(old.step_dev)
        self.step = old.step

    # ---- torch.optim.Adam state mirror: optimizer.state_dict() / load_state_dict() keep working ----------------
    def bind_optimizer(self, opt):
        """Expose the moments as views inside ``opt.state`` (torch.optim.Adam's own keys) so ``opt.state_dict()``
        saves them, and adopt whatever state ``opt`` already holds or later loads (``load_state_dict``).  The step
        count lives on the device (graph replays advance it); it is copied into the state's ``step`` entries right
        before a ``state_dict()`` call."""
        if self._opt is opt:
            return
        self._opt = opt
        prev = getattr(opt, "_itcv_group", None)
        prev = prev() if prev is not None else None
        opt._itcv_group = weakref.ref(self)
        # state mirrored by the group these parameters came from was carried over by _inherit (with the device-side
        # step count, which the mirror's ``step`` entries lag behind): only re-point the views then
        self._adopt(opt, copy=not (prev is not None and prev is self._parent))
        if hasattr(opt, "register_state_dict_pre_hook"):
            opt.register_state_dict_pre_hook(lambda o: self._refresh_steps(o) if self._current(o) else None)
            opt.register_load_state_dict_post_hook(lambda o: self._adopt(o) if self._current(o) else None)

    def _current(self, opt):
        ref = getattr(opt, "_itcv_group", None)
        return ref is not None and ref() is self

    def _mine(self, t, buf, o, n):
        return t is not None and t.data_ptr() == buf.data_ptr() + 4 * o and t.numel() == n and t.device == buf.device

    def _adopt(self, opt, copy=True):
        step = None
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                n = p.numel()
                st = opt.state.get(p)
                if copy and st and "exp_avg" in st and not self._mine(st["exp_avg"], self.exp_avg, o, n):
                    self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                    step = int(st["step"]) if step is None else max(step, int(st["step"]))
                st = opt.state[p]
                st["exp_avg"] = self.exp_avg[o:o + n].view(p.shape)
                st["exp_avg_sq"] = self.exp_avg_sq[o:o + n].view(p.shape)
                st.setdefault("step", torch.tensor(0.0))
            if step is not None:
                self.step = step
                self.step_dev.fill_(step)
        self._refresh_steps(opt)

    def _refresh_steps(self, opt):
        if opt is not self._opt:
            return
        step = float(int(self.step_dev.item()))
        for p in self.params:
            st = opt.state.get(p)
            if st is not None:
                st["step"] = torch.tensor(step)

    def owns(self, params):
        ps = list(params)
        return len(ps) == len(self.params) and all(a is b for a, b in zip(ps, self.params)) and all(
            p.data_ptr() == self.flat_p.data_ptr() + 4 * o for p, o in zip(self.params, self.offsets))

    def attach_grads(self):
        for p, o in zip(self.params, self.offsets):
            want = self.flat_g.data_ptr() + 4 * o
            if p.grad is None or p.grad.data_ptr() != want:
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    def zero_grad(self):
        call("itcv_fill", ptr(self.flat_g), self.numel, 0.0, stream())
        self.attach_grads()

    def sumsq_into(self, out_f64_slot):
        call("itcv_sumsq", ptr(self.flat_g), self.numel, ptr(out_f64_slot), ptr(self._ws), self._ws.numel(), stream())

    def scale_grads(self, coef_dev):
        call("itcv_scale_by_dev", ptr(self.flat_g), self.numel, ptr(coef_dev), stream())

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8):
        self.step += 1
        call("itcv_adam_step_dev", ptr(self.flat_p), ptr(self.flat_g), ptr(self.exp_avg), ptr(self.exp_avg_sq),
             self.numel, float(lr), float(betas[0]), float(betas[1]), float(eps), ptr(self.step_dev), stream())
        bump_weight_epoch(self.params)   # these parameters changed behind torch's back: drop their packed copies


def clip_grad_norm(groups, clip):
    """torch.nn.utils.clip_grad_norm_ over the union of ``groups`` (solvers/intro.py:113-115): returns the
    total norm as a 1-element device tensor; no host synchronisation."""
    dev = groups[0].flat_g.device
    sumsq = torch.empty(len(groups), dtype=torch.float64, device=dev)
    for i, g in enumerate(groups):
        g.sumsq_into(sumsq[i:i + 1])
    out = torch.empty(2, dtype=F32, device=dev)          # [norm, coef]
    call("itcv_clip_coef", ptr(sumsq), len(groups), float(clip), ptr(out[0:1]), ptr(out[1:2]), stream())
    for g in groups:
        g.scale_grads(out[1:2])
    return out[0:1]


def plain_adam_hparams(opt):
    """(lr, betas, eps) if ``opt`` is a torch.optim.Adam whose update is the plain one the fused
    kernel implements, else None (the caller then falls back to ``opt.step()``)."""
    if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
        return None
    g = opt.param_groups[0]
    if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
        return None
    if g.get("capturable", False) or g.get("differentiable", False) or isinstance(g["lr"], torch.Tensor):
        return None
    return g["lr"], tuple(g["betas"]), g["eps"]
