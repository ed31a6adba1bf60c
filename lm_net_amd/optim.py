"""AdamW over LM_Net's flat parameter / gradient buffers: ONE kernel per step.

The reference trains with ``torch.optim.AdamW(model.parameters(), lr, weight_decay)`` (``train.py:156``) and a
``CosineAnnealingLR`` on top of it (``train.py:160``).  ``LM_Net`` already writes every gradient into one flat
fp32 buffer (``LM_Net._new_grads``); this optimizer lays the PARAMETERS out the same way (each ``p.data`` becomes a
view of one buffer, same offsets as its gradient) so that the whole update is a single ``lmn_adamw_step`` launch over
~4 M floats instead of ~90 multi-tensor launches over 514 tensors.

It is a ``torch.optim.Optimizer`` (one param group), so LR schedulers, ``zero_grad`` and ``state_dict`` work as
with the reference's optimizer; ``state_dict()`` is emitted in ``torch.optim.AdamW``'s per-parameter layout and
``load_state_dict`` accepts it, so optimizer checkpoints written by the reference load here and vice versa.
"""
import torch

from . import hip


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        net = getattr(model, "module", model)          # accept the DDP wrapper
        if not hasattr(net, "_ensure_grad_layout"):
            raise TypeError("FusedAdamW needs an lm_net_amd.LM_Net (flat gradient layout)")
        self.net = net
        params = list(net.parameters())
        if not params or not params[0].is_cuda:
            raise RuntimeError("FusedAdamW: move the model to the GPU first (the HIP path has no CPU fallback)")
        if any(not p.requires_grad for p in params):
            # the one-launch update runs over the WHOLE flat buffer (moments, weight decay and step for every element);
            # torch.optim.AdamW skips parameters without a gradient -- that case is not supported here
            raise ValueError("FusedAdamW updates every parameter of LM_Net's flat buffer: parameters with "
                             "requires_grad=False are not supported (use torch.optim.AdamW for partial fine-tuning)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flatten_parameters()
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.step_count = 0

    # ------------------------------------------------------------------ layout
    def _flatten_parameters(self):
        L = self.net._ensure_grad_layout()
        flat = torch.zeros(L["total"], device=L["device"], dtype=torch.float32)
        with torch.no_grad():
            for p in L["order"]:
                a, b = L["offs"][id(p)]
                flat[a:b].copy_(p.detach().reshape(-1))
                p.data = flat[a:b].view(p.shape)
        self.flat_p, self._layout = flat, L

    def _flat_grad(self):
        """The model's flat gradient buffer if every p.grad is still its view, else a gathered copy."""
        L, net = self._layout, self.net
        flat = getattr(net, "_grad_flat", None)
        ok = flat is not None and flat.numel() == L["total"] and all(p.requires_grad for p in L["order"][::37])
        if ok:
            for p in (L["order"][0], L["order"][-1], L["order"][len(L["order"]) // 2]):
                a, _ = L["offs"][id(p)]
                if p.grad is None or p.grad.data_ptr() != flat.data_ptr() + 4 * a:
                    ok = False
                    break
        if ok:
            return flat
        g = torch.zeros_like(self.flat_p)
        for p in L["order"]:
            if p.grad is not None:
                a, b = L["offs"][id(p)]
                g[a:b].copy_(p.grad.reshape(-1))
        return g

    # ------------------------------------------------------------------ Optimizer API
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = self._layout
        p0 = L["order"][0]
        if p0.data_ptr() != self.flat_p.data_ptr() + 4 * L["offs"][id(p0)][0]:
            raise RuntimeError("FusedAdamW: parameter storage was replaced after the optimizer was built "
                               "(model.to()/load via .data=); rebuild the optimizer")
        grp = self.param_groups[0]
        b1, b2 = grp["betas"]
        self.step_count += 1
        hip.adamw_step(self.flat_p, self._flat_grad(), self.exp_avg, self.exp_avg_sq, grp["lr"], b1, b2, grp["eps"],
                       grp["weight_decay"], 1.0 - b1 ** self.step_count, 1.0 - b2 ** self.step_count)
        return loss

    def state_dict(self):
        """torch.optim.AdamW layout: state[i] = {step, exp_avg, exp_avg_sq} in model.parameters() order."""
        L = self._layout
        state = {}
        for i, p in enumerate(self.param_groups[0]["params"]):
            a, b = L["offs"][id(p)]
            state[i] = dict(step=torch.tensor(float(self.step_count)),
                            exp_avg=self.exp_avg[a:b].view(p.shape).clone(),
                            exp_avg_sq=self.exp_avg_sq[a:b].view(p.shape).clone())
        grp = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        grp["params"] = list(range(len(self.param_groups[0]["params"])))
        return dict(state=state, param_groups=[grp])

    def load_state_dict(self, sd):
        L = self._layout
        params = self.param_groups[0]["params"]
        for k, v in sd["param_groups"][0].items():
            if k != "params":
                self.param_groups[0][k] = v
        steps = set()
        for i, st in sd["state"].items():
            p = params[int(i)]
            a, b = L["offs"][id(p)]
            self.exp_avg[a:b].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[a:b].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FusedAdamW: per-parameter step counts differ; one shared count is kept")
        self.step_count = steps.pop() if steps else 0
