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


def allreduce_gradients(optimizer_or_params):
    """Sum gradients over ranks: one collective on FlatAdam's flat buffer, else one flattened bucket.

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
