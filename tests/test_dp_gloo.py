"""N>1 path on CPU: two gloo ranks, sharded minibatch, one gradient all-reduce == the full-batch gradient."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ctn_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import conv_tasnet_amd  # noqa: F401
    from conv_tasnet_amd import parallel
    w, r, dev = parallel.init_distributed(backend="gloo")
    assert (w, r) == (world, rank) and dev.type == "cpu"
    cfg = O.Config(N=16, L=20, B=8, H=16, P=3, X=2, R=1, C=2)
    sd = O.init_params(cfg, seed=5 + rank)            # deliberately different per rank ...
    params = [torch.nn.Parameter(v) for v in sd.values()]
    holder = torch.nn.ParameterList(params)
    parallel.broadcast_parameters(holder, src=0)      # ... until rank 0's weights are broadcast
    mix, lens, src = O.synth_batch(40, 4, 2000)        # the same global batch on every rank
    smix, slens, ssrc = parallel.shard_batch((mix, lens, src))
    assert smix.shape[0] == 2
    leaves = dict(zip(sd.keys(), params))
    loss, _, _, _ = O.cal_loss(ssrc, O.forward(cfg, leaves, smix), slens)
    loss.backward()
    scale = parallel.allreduce_gradients(params)
    assert scale == 1.0
    torch.save({"grads": [p.grad.clone() for p in params], "w0": params[0].detach().clone(), "loss": float(loss)},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_allreduce_equals_full_batch(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["w0"], r1["w0"])                       # broadcast happened
    for a, b in zip(r0["grads"], r1["grads"]):
        assert torch.equal(a, b)                                 # every rank holds the same reduced gradient
    # single-process reference: full batch of 4, loss = -mean over the batch
    cfg = O.Config(N=16, L=20, B=8, H=16, P=3, X=2, R=1, C=2)
    sd = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, seed=5).items()}
    mix, lens, src = O.synth_batch(40, 4, 2000)
    loss, _, _, _ = O.cal_loss(src, O.forward(cfg, sd, mix), lens)
    loss.backward()
    for g, p in zip(r0["grads"], sd.values()):
        assert float((g - p.grad).abs().max()) <= 1e-5 * float(p.grad.abs().max()) + 1e-8
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - float(loss)) < 1e-5


def test_shard_batch_rejects_ragged_split():
    from conv_tasnet_amd import parallel
    with pytest.raises(ValueError):
        parallel.shard_batch((torch.zeros(5, 3),), rank=0, world=2)
    a, = parallel.shard_batch((torch.arange(8).view(8, 1),), rank=1, world=4)
    assert a.flatten().tolist() == [2, 3]


def _loss_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from conv_tasnet_amd import parallel
    from conv_tasnet_amd.solver import Solver
    parallel.init_distributed(backend="gloo")
    n_local = (5, 3)[rank]
    x = torch.tensor([1.0 + rank], requires_grad=True)
    loss = (x * x).sum() * (2.0 + rank)                   # rank 0: 2, rank 1: 12
    back, report = Solver._global_loss(None, loss, n_local)
    back.backward()
    torch.save({"report": float(report), "grad": float(x.grad), "dloss": float((2.0 + rank) * 2 * (1.0 + rank))},
               os.path.join(out_dir, "l%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_global_minibatch_loss_weights_ragged_shards(tmp_path):
    """Solver._global_loss: the reported loss is the utterance-weighted mean over ranks, and the back-propagated loss is
    scaled so that the 1/world average of the gradient all-reduce equals the gradient of that mean."""
    world, port = 2, _free_port()
    mp.spawn(_loss_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / ("l%d.pt" % k)) for k in range(2)]
    assert r[0]["report"] == r[1]["report"] and abs(r[0]["report"] - (5 * 2.0 + 3 * 12.0) / 8) < 1e-6
    # (1/world) * sum_r scale_r dloss_r/dx_r  ==  d/dx of the weighted mean:  scale_r = n_r * world / n
    for k, n in enumerate((5, 3)):
        assert abs(r[k]["grad"] - r[k]["dloss"] * n * 2 / 8) < 1e-6


def _bucket_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from conv_tasnet_amd import parallel
    parallel.init_distributed(backend="gloo")

    class Opt:                                   # what GradientBuckets needs of FlatAdam: the flat gradient buffer
        pass
    sizes = [30, 1, 8, 8, 24, 1, 8, 8, 30]       # one "block": 9 tensors, each padded to a multiple of 4 in the flat buffer
    pad4 = lambda n: (n + 3) // 4 * 4            # noqa: E731
    per_block = sum(pad4(n) for n in sizes)
    front, nblocks, back = 52, 4, 20
    total = front + nblocks * per_block + back
    g = torch.Generator().manual_seed(7 + rank)
    init = torch.randn(total, generator=g)
    res = {}
    for mode in ("single", "buckets"):
        opt = Opt()
        opt.flat_grads = init.clone()
        if mode == "single":
            opt._ctn_buckets = None
        else:
            gb = parallel.GradientBuckets(opt, blocks_per_bucket=2)
            opt._ctn_buckets = gb
            for lo in (2, 0):                    # backward order: last blocks first
                sinks, o = [], front + lo * per_block
                for _ in range(2):
                    for n in sizes:
                        sinks.append(opt.flat_grads[o:o + n])
                        o += pad4(n)
                gb.bucket_ready(sinks)
            assert len(gb.covered) == 2
        scale = parallel.allreduce_gradients(opt)
        assert scale == 0.5
        res[mode] = opt.flat_grads.clone()
    torch.save(res, os.path.join(out_dir, "b%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_allreduce_equals_the_single_collective(tmp_path):
    """parallel.GradientBuckets (the overlapped, per-repeat all-reduce of the flat gradient): buckets started out of order plus
    the remainder reduce every element exactly once -- bitwise the result of one all-reduce of the whole buffer, on both ranks."""
    world, port = 2, _free_port()
    mp.spawn(_bucket_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "b0.pt"), torch.load(tmp_path / "b1.pt")
    assert torch.equal(r0["single"], r1["single"]) and torch.equal(r0["buckets"], r1["buckets"])
    assert torch.equal(r0["single"], r0["buckets"])
