"""Child process of tests/test_dp_gpu.py: one data-parallel rank (or the single-process reference run) of the PRODUCT
training path -- ConvTasNet on the HIP kernels, FlatAdam, Solver -- on cuda:0, gloo rendezvous on 127.0.0.1.

    python tests/dp_worker.py <rank> <world> <port> <out.pt>

Phase A: one epoch of 3 training steps at lr 1e-3 on ragged shards (rank 0 takes 5 of each global minibatch's 8
utterances, rank 1 the other 3; the single-process run takes all 8).  Phase B: epochs at lr 1e-12 (weights frozen to
the ulp, so the validation loss repeats): the schedule must halve at the 3rd..6th repeat and stop at the 7th, on every
rank in the same epoch.  Validation minibatches are dealt 2 / 1 to the ranks (unequal counts are legal there)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                  CTN_DIST_BACKEND="gloo")

import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import parallel  # noqa: E402
from conv_tasnet_amd.optim import FlatAdam  # noqa: E402
from conv_tasnet_amd.solver import Solver  # noqa: E402
from conv_tasnet_amd.train import SyntheticLoader  # noqa: E402

w, r, dev = parallel.init_distributed()
assert (w, r) == (world, rank) and dev.type == "cuda"
gen = SyntheticLoader(1, 1, samples=4000)


def batch(utts):
    src = torch.stack([gen._utt(u) for u in utts])
    return src.sum(1), torch.full((len(utts),), 4000, dtype=torch.long), src


split = {(2, 0): slice(0, 5), (2, 1): slice(5, 8), (1, 0): slice(0, 8)}[(world, rank)]
train = [batch(list(range(8 * b, 8 * b + 8))[split]) for b in range(3)]
cv_all = [batch([500 + 2 * i, 501 + 2 * i]) for i in range(3)]
cv = cv_all[rank::world]

save = out + ".dir"


def phase_a(overlap):
    """3 training steps from the broadcast weights; overlap: the gradient all-reduce in per-repeat buckets issued behind the
    weight-gradient stream during the backward pass (parallel.GradientBuckets) or as one collective after it."""
    os.environ["CTN_DP_OVERLAP"] = "1" if overlap else "0"
    torch.manual_seed(100 + rank)                  # different initial weights per rank, until the broadcast
    model = ctn.ConvTasNet(32, 20, 16, 32, 3, 2, 2, 2).to(dev)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    parallel.broadcast_parameters(opt.flat_params)
    arg = (1, 1, 1, 1, 5, save, 0, "", "a.pth.tar", 1000, 0, 0, "x")
    sa = Solver({"tr_loader": train, "cv_loader": cv}, model, opt, arg)
    assert (opt._ctn_buckets is not None) == (overlap and world > 1)
    sa.train()
    return model, opt, sa


_, opt_plain, _ = phase_a(False)
plain = opt_plain.flat_params.detach().clone()
model, opt, sa = phase_a(True)
res = {"losses": list(sa.iter_losses[:3]), "cv_a": float(sa.cv_loss[0]), "params": opt.flat_params.detach().cpu().clone(),
       "buckets_equal_single_collective": bool(torch.equal(opt.flat_params, plain))}



def phase_paper(overlap):
    """Two steps at the PAPER's widths and depth (B = 256, H = 512, 32 blocks: the default h3 arithmetic, four buckets of eight
    blocks = 8.4 MB each + the front / back-end remainder), one 1-s utterance per rank: buckets against the single collective."""
    os.environ["CTN_DP_OVERLAP"] = "1" if overlap else "0"
    torch.manual_seed(7)
    m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2).to(dev)
    o = FlatAdam(m.parameters(), lr=1e-3)
    parallel.broadcast_parameters(o.flat_params)
    gb = parallel.enable_overlap(o, 8)
    assert (gb is not None) == (overlap and world > 1)
    g = SyntheticLoader(1, 1, samples=8000)
    for step in range(2):
        src = torch.stack([g._utt(900 + 2 * step + rank)]).to(dev)
        o.zero_grad()
        ctn.cal_loss(src, m(src.sum(1)), torch.full((1,), 8000, dtype=torch.long, device=dev))[0].backward()
        o.step(max_grad_norm=5.0, grad_scale=parallel.allreduce_gradients(o))
    ctn.ops.set_grad_buckets(None)
    return o.flat_params.detach().clone()


pp_plain = phase_paper(False)
pp_buckets = phase_paper(True)
res["paper_buckets_equal_single_collective"] = bool(torch.equal(pp_plain, pp_buckets))
res["paper_params_finite"] = bool(torch.isfinite(pp_buckets).all())
res["gemm_arith"] = ctn.gemm_arith()
del pp_plain, pp_buckets

opt.param_groups[0]["lr"] = 1e-12
arg = (1, 12, 1, 1, 5, save, 0, "", "b.pth.tar", 1000, 0, 0, "x")
sb = Solver({"tr_loader": train, "cv_loader": cv}, model, opt, arg)
sb.train()
res.update(lr=float(opt.param_groups[0]["lr"]), misses=int(sb.val_no_impv), epochs_run=len(sb.iter_losses) // (3 + len(cv)),
           saved=sorted(os.listdir(save)) if os.path.isdir(save) else [])
torch.save(res, out)
if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
print("dp_worker rank %d/%d done: lr %.3e epochs_run %d" % (rank, world, res["lr"], res["epochs_run"]), flush=True)
