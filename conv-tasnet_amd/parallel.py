"""Data parallelism, MI355X style: one process per GPU, torch.distributed ('nccl' == RCCL over xGMI).

Replaces nn.DataParallel of the reference (src/train.py:83-85: per-step broadcast of all 294 tensors, scatter,
gather, reduce-add on GPU0) by: replicated parameters, an independent shard of the minibatch per rank, and ONE
all-reduce of the flat fp32 gradient buffer (34.8 MB at the paper config) per step.  On the 8-GPU xGMI mesh a
single large message lets RCCL use all 7 links per GPU; the averaging (1/world) is folded into the optimiser
kernel's grad_scale.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(backend=None):
    """Initialise from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (world, rank, device)."""
    world, rank, local = env_world()
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        # one GPU per rank; ranks beyond the visible devices wrap around (only useful to rehearse the multi-rank
        # code path on a single-GPU box together with CTN_DIST_BACKEND=gloo -- RCCL needs distinct devices)
        local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("CTN_DIST_BACKEND") or ("nccl" if use_cuda else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return world, rank, device


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def broadcast_parameters(model_or_flat, src=0):
    """Make every replica start from rank `src`'s weights."""
    if world_size() == 1:
        return
    if torch.is_tensor(model_or_flat):
        dist.broadcast(model_or_flat, src)
        return
    for p in model_or_flat.parameters():
        dist.broadcast(p.data, src)
    for b in model_or_flat.buffers():
        dist.broadcast(b.data, src)


class GradientBuckets:
    """The gradient all-reduce in buckets, issued while the backward pass is still running.

    FlatAdam's flat gradient is contiguous per TemporalBlock (9 tensors each, model order), so the blocks of one repeat
    are one slice of it.  With this object installed (enable_overlap) the composite backward of the stack (ops.TcnGln /
    ops.TcnCln) is issued repeat by repeat -- each call ends with the weight-gradient stream joined into the main stream --
    and after each one `bucket_ready(sinks)` starts an asynchronous all-reduce of that repeat's slice: with the RCCL backend
    it runs on the process group's own stream behind an event of the current stream, i.e. beside the backward kernels of
    the next repeat (34.8 MB at the paper config = 4 x 8.4 MB + the front / back-end layers).  `finish()` reduces whatever
    no bucket covered and waits for all of them.  Every element is summed over the ranks exactly once, so the replicas
    stay identical; with two ranks the result is bitwise the single-collective one (a + b == b + a)."""

    def __init__(self, optimizer, blocks_per_bucket):
        self.opt = optimizer
        self.blocks_per_bucket = int(blocks_per_bucket)
        self.works, self.covered = [], []

    def reset(self):
        """Forget the buckets of a backward pass that was not followed by allreduce_gradients() (an exception, a skipped step,
        a diagnostic backward): wait for what was started -- every rank started the same collectives, or the job is lost
        anyway -- and drop the covered ranges, so that the next finish() reduces every slice of the NEW gradients.
        Called by FlatAdam.zero_grad().  A backward pass under enable_overlap must be followed by allreduce_gradients() on
        every rank (or by zero_grad() on every rank)."""
        for w in self.works:
            w.wait()
        self.works, self.covered = [], []

    def bucket_ready(self, sinks):
        flat = self.opt.flat_grads
        base, esz = flat.data_ptr(), flat.element_size()
        lo = min(s.data_ptr() for s in sinks)
        hi = max(s.data_ptr() + esz * ((s.numel() + 3) // 4 * 4) for s in sinks)
        lo, hi = (lo - base) // esz, min((hi - base) // esz, flat.numel())
        if lo < 0 or hi > flat.numel() or sum((s.numel() + 3) // 4 * 4 for s in sinks) != hi - lo:
            return                                  # not one contiguous slice of the flat buffer: leave it to finish()
        self.works.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        self.covered.append((lo, hi))

    def finish(self):
        flat = self.opt.flat_grads
        pos = 0
        for lo, hi in sorted(self.covered) + [(flat.numel(), flat.numel())]:
            if lo > pos:
                self.works.append(dist.all_reduce(flat[pos:lo], op=dist.ReduceOp.SUM, async_op=True))
            pos = max(pos, hi)
        for w in self.works:
            w.wait()                                # (RCCL: the current stream waits for the collective's stream)
        self.works, self.covered = [], []


def enable_overlap(optimizer, blocks_per_bucket):
    """Install bucketed, overlapped gradient all-reduce for a FlatAdam optimiser (no-op for one process).
    blocks_per_bucket = TemporalBlocks per bucket, normally X (one bucket per repeat)."""
    from . import ops
    if world_size() == 1 or getattr(optimizer, "flat_grads", None) is None or os.environ.get("CTN_DP_OVERLAP", "1") == "0":
        ops.set_grad_buckets(None)
        optimizer._ctn_buckets = None
        return None
    gb = GradientBuckets(optimizer, blocks_per_bucket)
    optimizer._ctn_buckets = gb
    ops.set_grad_buckets(gb)
    return gb


def allreduce_gradients(optimizer_or_params):
    """Sum gradients over ranks: FlatAdam's flat buffer (one collective, or the buckets of enable_overlap that were
    started during the backward pass plus the remainder), else one flattened bucket.

    Returns the scale (1/world) the caller applies (FlatAdam.step(grad_scale=...)); for plain parameter lists the
    gradients are already averaged in place and 1.0 is returned."""
    w = world_size()
    if w == 1:
        return 1.0
    flat = getattr(optimizer_or_params, "flat_grads", None)
    if flat is not None:
        if flat.is_cuda:
            from . import ops
            ops.join_side_stream(flat.device)              # side-stream weight gradients must have landed
        gb = getattr(optimizer_or_params, "_ctn_buckets", None)
        if gb is not None:
            gb.finish()
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return 1.0 / w
    grads = [p.grad for p in optimizer_or_params if p.grad is not None]
    if grads:
        bucket = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
        bucket.div_(w)
        off = 0
        for g in grads:
            g.copy_(bucket[off:off + g.numel()].view_as(g))
            off += g.numel()
    return 1.0


def shard_batch(tensors, rank=None, world=None):
    """Contiguous equal shard of dim 0 for this rank (64 utterances -> 8 per rank)."""
    world = world or world_size()
    if world == 1:
        return tensors
    rank = dist.get_rank() if rank is None else rank
    out = []
    for t in tensors:
        n = t.shape[0]
        if n % world:
            raise ValueError("global batch %d is not divisible by world size %d" % (n, world))
        per = n // world
        out.append(t[rank * per:(rank + 1) * per])
    return tuple(out)
