"""One fixed-shape training step's device work (zero_grad + forward + PIT loss + backward) as a HIP graph.

The reference's step (src/solver.py:181-198) issues ~450 kernel launches per minibatch from Python.  Training
minibatches of Conv-TasNet all have one shape (4 s segments, src/data.py:287-296), so the launch sequence is recorded
once with stream capture -- including the weight-gradient kernels forked onto the second stream -- and replayed with a
single ``hipGraphLaunch``; the host then only copies the next minibatch into the static input buffers.

The gradient all-reduce and the clip + Adam kernels stay outside the graph (three launches): the step counter and the
learning rate are host values of ``FlatAdam.step`` and RCCL keeps its own stream semantics.

    opt  = FlatAdam(model.parameters(), lr=1e-3)
    step = GraphedBackprop(model, opt, sample_batch=(mix, lens, src))
    for mix, lens, src in loader:
        loss = step(mix, lens, src)          # device scalar; gradients are in opt.flat_grads
        opt.step(max_grad_norm=5.0, grad_scale=parallel.allreduce_gradients(opt))

Batches of a different shape (the last, shorter minibatch of an epoch; validation) must take the eager path --
``matches()`` tells which.
"""
import torch

from . import ops
from .pit_criterion import cal_loss


class GraphedBackprop:
    def __init__(self, model, optimizer, sample_batch, warmup=2):
        mix, lens, src = sample_batch
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise ValueError("GraphedBackprop needs the model on the GPU")
        if not hasattr(optimizer, "flat_grads"):
            raise ValueError("GraphedBackprop needs FlatAdam (gradients must live at fixed addresses)")
        self.model, self.opt = model, optimizer
        self.mix = mix.to(dev, torch.float32).clone()
        self.lens = lens.to(dev).clone()
        self.src = src.to(dev, torch.float32).clone()
        self.loss = None
        self.max_snr = None
        was_training = model.training
        model.train()
        # Overlapped gradient buckets (parallel.enable_overlap) issue collectives from inside the backward pass: not in the
        # warm-up (their works / covered ranges would be left over for the first real step) and not in the capture (a collective
        # baked into the graph would replay every step beside finish()'s own reduction of the slice).  The graphed step reduces
        # the whole flat gradient with ONE collective outside the graph (parallel.allreduce_gradients).
        saved_buckets = ops._GRAD_BUCKETS
        ops.set_grad_buckets(None)
        if getattr(optimizer, "_ctn_buckets", None) is not None:
            optimizer._ctn_buckets = None
        # eager warm-up on a side stream: sizes every cached workspace and lets the allocator settle before capture
        cur = torch.cuda.current_stream(dev)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup)):
                self._body()
        cur.wait_stream(s)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.max_snr = self._body()
        del saved_buckets           # (stays uninstalled: every later backward of this optimiser would be un-captured bucket work)
        model.train(was_training)

    def _body(self):
        self.opt.zero_grad()
        est = self.model(self.mix)
        loss, max_snr, _, _ = cal_loss(self.src, est, self.lens)
        loss.backward()
        ops.join_side_stream(self.mix.device)      # the forked weight-gradient stream must rejoin inside the capture
        return loss.detach(), max_snr.detach()

    def matches(self, mix, lens, src):
        return tuple(mix.shape) == tuple(self.mix.shape) and tuple(src.shape) == tuple(self.src.shape) \
            and tuple(lens.shape) == tuple(self.lens.shape)

    def __call__(self, mix, lens, src):
        """Replays the recorded step on this minibatch.  Returns the loss (device scalar, overwritten by the next call)."""
        if not self.matches(mix, lens, src):
            raise ValueError("GraphedBackprop was captured for mixture %s / sources %s; got %s / %s"
                             % (tuple(self.mix.shape), tuple(self.src.shape), tuple(mix.shape), tuple(src.shape)))
        if mix.data_ptr() != self.mix.data_ptr():
            self.mix.copy_(mix, non_blocking=True)
        if lens.data_ptr() != self.lens.data_ptr():
            self.lens.copy_(lens, non_blocking=True)
        if src.data_ptr() != self.src.data_ptr():
            self.src.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.loss
