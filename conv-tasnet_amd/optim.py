"""FlatAdam: torch.optim.Adam semantics on ONE flat fp32 buffer, stepped by a single HIP kernel pair.

Replaces the tail of the reference training step, ``clip_grad_norm_`` + ``Adam.step`` over 294 tensors
(src/solver.py:194-196, src/train.py:92-95).  Parameters become views into ``flat_params``; their ``.grad``
are views into ``flat_grads`` -- which is also the single all-reduce payload of the data-parallel step.
``state_dict()`` / ``load_state_dict()`` use torch.optim.Adam's layout, so ``optim_dict`` of a reference
checkpoint loads here and vice versa.
"""
import torch

from ._lib import lib


def _round4(n):
    return (n + 3) // 4 * 4


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, direct_grads=True):
        if weight_decay != 0:
            raise NotImplementedError("FlatAdam: weight_decay != 0 is not on the hot path (reference uses l2 = 0)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0))
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdam takes a single parameter group")
        # direct_grads: the HIP backward stages write each parameter gradient straight into flat_grads (no
        # AccumulateGrad add per tensor).  Semantics: a backward pass OVERWRITES the gradient, so accumulate over
        # several backward calls only with direct_grads=False.
        self.direct_grads = bool(direct_grads)
        self._written = set()       # parameters whose sink a backward stage has overwritten since the last zero_grad()
        self._flatten()
        self._step = 0
        self.last_total_norm = None

    # ---- flat storage -----------------------------------------------------------------
    def _flatten(self):
        ps = self.param_groups[0]["params"]
        dev = ps[0].device
        if dev.type != "cuda":
            raise lib_error("FlatAdam needs parameters on the GPU (move the model first)")
        self._offsets = []
        off = 0
        for p in ps:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FlatAdam: fp32 parameters on one device only")
            self._offsets.append(off)
            off += _round4(p.numel())          # keep every tensor 16-byte aligned for the float4 GEMM loads
        self.numel = off
        self.flat_params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grads = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self._ws = torch.empty(lib.ctn_optim_parts(), dtype=torch.float64, device=dev)
        self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
        for p, o in zip(ps, self._offsets):
            n = p.numel()
            self.flat_params[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_params[o:o + n].view(p.shape)
            p.grad = self.flat_grads[o:o + n].view(p.shape)
            if self.direct_grads:
                p._ctn_grad_sink = p.grad
                p._ctn_sink_owner = self

    def _grad_views_intact(self):
        base = self.flat_grads.data_ptr()
        for p, o in zip(self.param_groups[0]["params"], self._offsets):
            if p.grad is None or p.grad.data_ptr() != base + 4 * o:
                return False
        return True

    def zero_grad(self, set_to_none=False):
        """One memset; the .grad views stay attached (set_to_none is ignored on purpose)."""
        gb = getattr(self, "_ctn_buckets", None)
        if gb is not None and (gb.works or gb.covered):
            gb.reset()              # buckets of a backward pass that never reached allreduce_gradients(): parallel.GradientBuckets.reset
        self.flat_grads.zero_()
        self._written.clear()
        if not self._grad_views_intact():
            for p, o in zip(self.param_groups[0]["params"], self._offsets):
                p.grad = self.flat_grads[o:o + p.numel()].view(p.shape)

    # ---- step -------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=0.0, grad_scale=1.0):
        """Adam step; with max_grad_norm > 0 the clip_grad_norm_ rule is fused in front of it.

        grad_scale multiplies the gradient first (1/world after a summing all-reduce).
        The total gradient norm (after grad_scale) is left in ``last_total_norm`` (device scalar)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._grad_views_intact():      # somebody re-assigned .grad: gather into the flat buffer
            for p, o in zip(self.param_groups[0]["params"], self._offsets):
                seg = self.flat_grads[o:o + p.numel()]
                if p.grad is None:
                    seg.zero_()
                elif p.grad.data_ptr() != seg.data_ptr():
                    seg.copy_(p.grad.reshape(-1))
        from . import ops
        ops.join_side_stream(self.flat_grads.device)      # weight-gradient kernels run on a second stream
        g = self.param_groups[0]
        self._written.clear()
        self._step += 1
        b1, b2 = g["betas"]
        lib.call("ctn_clip_adam_step", self.flat_params.data_ptr(), self.flat_grads.data_ptr(), self.exp_avg.data_ptr(),
                 self.exp_avg_sq.data_ptr(), self.numel, float(grad_scale), float(max_grad_norm), float(g["lr"]),
                 float(b1), float(b2), float(g["eps"]), self._step, self._norm.data_ptr(), self._ws.data_ptr(),
                 torch.cuda.current_stream().cuda_stream)
        self.last_total_norm = self._norm
        return loss

    # ---- torch.optim.Adam-compatible (de)serialisation -----------------------------------
    def state_dict(self):
        ps = self.param_groups[0]["params"]
        state = {}
        if self._step > 0:
            for i, (p, o) in enumerate(zip(ps, self._offsets)):
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self._step)),
                            "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        grp = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        grp["params"] = list(range(len(ps)))
        return {"state": state, "param_groups": [grp]}

    def load_state_dict(self, sd):
        ps = self.param_groups[0]["params"]
        grp = sd["param_groups"][0]
        for k in ("lr", "betas", "eps"):
            if k in grp:
                self.param_groups[0][k] = tuple(grp[k]) if k == "betas" else grp[k]
        steps = set()
        for i, (p, o) in enumerate(zip(ps, self._offsets)):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is None:
                continue
            n = p.numel()
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError("FlatAdam: per-parameter step counts differ")
        self._step = steps.pop() if steps else 0


def lib_error(msg):
    from ._lib import CtnError
    return CtnError(msg)
