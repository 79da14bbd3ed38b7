"""torch.optim.Adam for models on the fused training path (studiosr_amd/fasttrain.py): when every parameter is a view of the plan's flat
buffer P and every gradient a view of its flat buffer G -- what a fused training step leaves behind -- the update of the reference Trainer's
optimizer (studiosr/engine/trainer.py:133-139: Adam(lr, betas, weight_decay)) is ONE sr_tr_adam launch over the flat buffers instead of
torch's multi-tensor kernels over ~800 parameter tensors.  Same arithmetic, same `state_dict()` layout (per-parameter `step`, `exp_avg`,
`exp_avg_sq`: the moments are views of two flat buffers), so checkpoints move between this class and torch.optim.Adam.  Anything else --
another model, gradients that are not views of G (generic engine, gradient clipping that re-allocates, ...) -- takes torch's own step."""
from __future__ import annotations

import torch

from . import _lib as L


class Adam(torch.optim.Adam):
    def __init__(self, params, model=None, **kw) -> None:
        self._sr_model = model
        params = list(params)
        # (No `fused=True` default for the steps the flat path does not cover: torch._fused_adam_ updates the parameters WITHOUT bumping their
        # version counters (torch 2.10), which the packed-weight caches of the model key on; torch's default foreach step bumps them.  Whatever the
        # caller chooses, step() below invalidates those caches itself.)
        super().__init__(params, **kw)
        self._flat = None  # (plan, m, v)
        self._steps = 0

    def _plan(self):
        m = self._sr_model
        plan = getattr(m, "_fast_plan", None) if m is not None else None
        if plan is None or not plan.fp.intact() or len(self.param_groups) != 1:
            return None
        g = self.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or len(g["params"]) != len(plan.fp.params):
            return None
        G = plan.fp.G
        lo, hi = G.data_ptr(), G.data_ptr() + 4 * G.numel()
        for p in g["params"]:
            if p.grad is None or not (lo <= p.grad.data_ptr() < hi) or p.grad.data_ptr() - lo != 4 * plan.fp.off(p):
                return None
        return plan

    def _flat_state(self, plan):
        fp = plan.fp
        if self._flat is not None and self._flat[0] is plan:
            return self._flat
        m, v = torch.zeros_like(fp.P), torch.zeros_like(fp.P)
        step = None
        for p in fp.params:  # adopt whatever state exists (a resumed checkpoint, earlier torch steps), then alias it to the flat buffers
            st = self.state[p]
            o, n = fp.off(p), p.numel()
            if "exp_avg" in st:
                m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                step = float(st["step"]) if step is None else step
            st["exp_avg"], st["exp_avg_sq"] = m[o:o + n].view_as(p), v[o:o + n].view_as(p)
        self._steps = int(step or 0)
        self._step_t = torch.tensor(float(self._steps), dtype=torch.float32)
        for p in fp.params:
            self.state[p]["step"] = self._step_t  # one shared counter tensor
        self._flat = (plan, m, v)
        return self._flat

    @torch.no_grad()
    def step(self, closure=None):
        plan = self._plan()
        if plan is None:
            if self._flat is not None:  # leaving the flat path: give every parameter its own step tensor again
                g0 = self.param_groups[0]
                on_device = bool(g0.get("capturable") or g0.get("fused"))  # as torch's Adam._init_group: torch._fused_adam_ reads the step tensors on the device
                for p in self._flat[0].fp.params:
                    self.state[p]["step"] = torch.tensor(float(self._steps), dtype=torch.float32, device=p.device if on_device else "cpu")
                self._flat = None
            out = super().step(closure)
            self._invalidate_packed()
            return out
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        _, m, v = self._flat_state(plan)
        g = self.param_groups[0]
        self._steps += 1
        self._step_t.fill_(float(self._steps))
        b1, b2 = g["betas"]
        fp = plan.fp
        L.check(L.lib().sr_tr_adam(fp.P.data_ptr(), fp.G.data_ptr(), m.data_ptr(), v.data_ptr(), fp.n, float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                                   float(g["weight_decay"]), self._steps, torch.cuda.current_stream(fp.P.device).cuda_stream), "sr_tr_adam")
        self._invalidate_packed()
        return loss

    def _invalidate_packed(self) -> None:
        """The parameters changed, possibly behind torch's version counters (sr_tr_adam, torch._fused_adam_): drop the model's packed-weight caches."""
        m = self._sr_model
        if m is not None and hasattr(m, "invalidate_packed"):
            m.invalidate_packed()

    def load_state_dict(self, state_dict):
        self._flat = None
        return super().load_state_dict(state_dict)
