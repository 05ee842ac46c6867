"""Fused Adam over the model's flat parameter buffer (reference: torch.optim.Adam(betas=(0.9,0.99)), main_dist.py:50,
stepped at utils.py:413).  One HIP launch updates all 37.6 M parameters (28 B/param of HBM traffic) instead of ~170
per-tensor updates; the step counter lives on the device so the launch is hipGraph-capturable."""
import torch

from ._lib import lib, check, stream_ptr


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        net = model.module if hasattr(model, "module") else model
        self.net = net
        params = net._ordered_params()
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        flat = net.store.flat
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self._step2 = torch.zeros(2, dtype=torch.int32, device=flat.device)      # [steps taken, the update kernel's completion ticket]
        self.step_count = self._step2[:1]
        self.grad_scale = grad_scale
        # attached: a backward may leave its last weight gradients (the stem's) running on the side stream; step() updates everything
        # else under them (mdl._Plan.run_backward / ZSGNet.join_grads)
        net._fused_opt = self

    def zero_grad(self, set_to_none: bool = False):
        """One memset of the flat gradient buffer; the p.grad views stay (backward accumulates into them).  With
        set_to_none=True the views are dropped instead and the next backward re-creates them."""
        if set_to_none:
            for p in self.net._ordered_params():
                p.grad = None
            return
        st = self.net.store
        self.net.join_grads()
        self.net._grad_reduced = False          # (DDP: one reduced backward per zero_grad, see _Plan.run_backward)
        if st.grad is not None and st.grad.is_cuda:
            check(lib.zsg_memset_f32(st.grad.data_ptr(), st.grad.numel(), 0.0, stream_ptr()), "zero_grad")

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        st = self.net.store
        ov = getattr(self.net, "_adam_overlap", None)
        self.net._adam_overlap = None
        self.net.join_weight_readers()          # (a forward without backward may still be reading the weights on the side stream)
        hp = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), float(self.grad_scale))
        if ov is not None:
            # [off, end): gradients complete behind the event; [0, off) (stem / first block) after the side stream's last weight gradients
            ev, side, off = ov
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            n = st.flat.numel()
            check(lib.zsg_adam_step_range(st.flat.data_ptr() + 4 * off, st.grad.data_ptr() + 4 * off, self.m.data_ptr() + 4 * off,
                                          self.v.data_ptr() + 4 * off, n - off, *hp, self.step_count.data_ptr(), 0, stream_ptr()), "zsg_adam_step_range")
            cur.wait_stream(side)
            check(lib.zsg_adam_step_range(st.flat.data_ptr(), st.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), off, *hp,
                                          self.step_count.data_ptr(), 1, stream_ptr()), "zsg_adam_step_range")
            return
        check(lib.zsg_adam_step(st.flat.data_ptr(), st.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), st.flat.numel(), *hp,
                                self.step_count.data_ptr(), stream_ptr()), "zsg_adam_step")

    def state_dict(self):
        d = super().state_dict()
        d["zsg"] = dict(m=self.m, v=self.v, step=self.step_count)
        return d

    def load_state_dict(self, sd):
        """Restores the moments / step counter AND the param_groups (lr as left by the scheduler, betas, eps, weight decay).
        A plain torch.optim.Adam state dict (a reference checkpoint) has per-parameter OIHW moments that do not map onto
        the flat OHWI buffer: refuse it loudly instead of silently restarting the moments."""
        z = sd.get("zsg")
        if z is None:
            raise ValueError("FusedAdam.load_state_dict: no 'zsg' entry — this is not a FusedAdam state (a torch.optim.Adam "
                             "state of the reference cannot be mapped onto the flat parameter buffer); resume with load_opt=False")
        self.m.copy_(z["m"])
        self.v.copy_(z["v"])
        self.step_count.copy_(z["step"].to(self.step_count.dtype))
        for g, gs in zip(self.param_groups, sd.get("param_groups", [])):
            for k in ("lr", "betas", "eps", "weight_decay"):
                if k in gs:
                    g[k] = gs[k]
