"""GPU: the product data-parallel path with two ranks (HIP kernels + FlatAdam's flat all-reduce + Solver control plane).

The two ranks and a single-process reference run are separate child processes on cuda:0 with a gloo rendezvous
(tests/dp_worker.py).  They are spawned by conftest.py BEFORE any test touches the GPU (a process that has initialised
the GPU must not start other programs on this pool) and only collected here."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_two_rank_product_training_matches_single_process(dp_children):
    procs, outs, logs = dp_children
    for name, p in procs.items():
        rc = p.wait(timeout=800)
        assert rc == 0, "%s exited with %d:\n%s" % (name, rc, open(logs[name]).read()[-3000:])
    r0, r1, one = (torch.load(outs[k], weights_only=True) for k in ("rank0", "rank1", "single"))
    # replicas stay bitwise identical: same reduced gradient, same optimiser kernel, same LR on both ranks
    assert torch.equal(r0["params"], r1["params"])
    # the bucketed all-reduce issued during the backward pass (one bucket per repeat behind the weight-gradient stream + the
    # remainder) gives bitwise the parameters of the single collective after it (two ranks: a + b == b + a)
    assert r0["buckets_equal_single_collective"] and r1["buckets_equal_single_collective"] and one["buckets_equal_single_collective"]
    # the same at the paper's widths and depth under the default arithmetic (four 8.4 MB buckets + remainder, h3 GEMMs)
    for r in (r0, r1, one):
        assert r["paper_buckets_equal_single_collective"] and r["paper_params_finite"], r["gemm_arith"]
    # ragged shards (5 + 3 utterances): the weighted global-minibatch loss equals the single-process loss on all 8
    for a, b, c in zip(r0["losses"], r1["losses"], one["losses"]):
        assert a == b, "ranks report different global losses"
        assert abs(a - c) < 1e-3, (a, c)
    rel = float((r0["params"] - one["params"]).abs().max() / one["params"].abs().max())
    assert rel < 5e-4, "parameters after 3 DP steps drifted from the single-process run: %.2e" % rel
    # validation minibatches dealt 2 + 1: both ranks hold the all-reduced value, equal to the single-process one
    assert r0["cv_a"] == r1["cv_a"] and abs(r0["cv_a"] - one["cv_a"]) < 1e-3
    # forced plateau: halved at the 3rd..6th repeat, stopped at the 7th -- every process in the same epoch
    for r in (r0, r1, one):
        assert r["misses"] == 7 and r["epochs_run"] == 8, r
        assert abs(r["lr"] - 1e-12 / 16) < 1e-20, r["lr"]
    assert "b.pth.tar" in r0["saved"] and r1["saved"] == [], "only rank 0 saves"
