"""GPU: the callers around the hot path -- Solver trajectory vs the reference Solver, FlatAdam, checkpoints,
separate(), evaluate()."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ctn_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_arith")]     # every test under both GEMM arithmetics (conftest.py)

import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd.optim import FlatAdam  # noqa: E402
from conv_tasnet_amd.solver import Solver  # noqa: E402

DEV = "cuda:0"


def _traj_setup(name="solver_traj"):
    g = load_golden(name)
    N, L, B, H, P, X, R, C = [int(v) for v in g["cfg"]]
    m = ctn.ConvTasNet(N, L, B, H, P, X, R, C)
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p0:")})
    T = int(g["T"])
    batches = [O.synth_batch(900 + 2 * i, 2, T) for i in range(3)]
    return g, m.to(DEV), batches


@pytest.mark.parametrize("fixture", ["solver_traj", "solver_traj_wide"])
@pytest.mark.parametrize("flat", [True, False])
def test_solver_trajectory_matches_reference_solver(tmp_path, flat, fixture):
    """Same weights, batches, Adam(1e-3), clip 5, 4 epochs as the run recorded from src/solver.py (oracle/make_golden.py).
    solver_traj: the tiny config (one layer with >= 64 rows); solver_traj_wide: B = 64, H = 128, so that every GEMM of the stacks
    runs the arithmetic under test (h3 / b6) against the REFERENCE's recorded 12-step trajectory."""
    g, m, batches = _traj_setup(fixture)
    opt = FlatAdam(m.parameters(), lr=1e-3) if flat else torch.optim.Adam(m.parameters(), lr=1e-3)
    arg = (1, int(g["epochs"]), 1, 0, 5, str(tmp_path), 0, "", "final.pth.tar", 1000, 0, 0, "x")
    s = Solver({"tr_loader": batches, "cv_loader": batches[:1]}, m, opt, arg)
    s.train()
    np.testing.assert_allclose(np.array(s.iter_losses), g["iter_losses"], atol=2e-3)
    np.testing.assert_allclose(s.tr_loss.numpy(), g["tr_loss"], atol=2e-3)       # includes the (n+1) divisor
    np.testing.assert_allclose(s.cv_loss.numpy(), g["cv_loss"], atol=2e-3)
    assert abs(opt.param_groups[0]["lr"] - float(g["final_lr"])) < 1e-12
    for k, v in m.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g["p1:" + k], atol=3e-4, err_msg=k)
    pkg = torch.load(tmp_path / "final.pth.tar", weights_only=False)
    assert sorted(pkg.keys()) == sorted(str(k) for k in g["pkg_keys"]) and pkg["epoch"] == int(g["pkg_epoch"])


@pytest.mark.parametrize("side", [True, False])
@pytest.mark.parametrize("norm,causal", [("gLN", False), ("cLN", True)])
def test_bucketed_backward_equals_the_single_call(norm, causal, side):
    """The per-bucket backward of the stacks (parallel.enable_overlap: one composite call per repeat, last repeat first, the
    weight-gradient stream left un-joined between calls, its own workspace per un-joined call, a fresh operand maximum at every
    range start) at widths where the arithmetic under test runs (B = 64, H = 128: h3 / b6 pieces; the two-rank test of
    tests/dp_worker.py uses B = 16, H = 32 = the fp32 kernels): with a stand-in GradientBuckets whose bucket_ready() does
    nothing, every gradient must be BITWISE the single-call backward's -- with and without the second stream."""
    from conv_tasnet_amd import ops

    class Stub:
        blocks_per_bucket = 2
        works, covered = [], []
        calls = 0

        def bucket_ready(self, sinks):
            Stub.calls += 1

    torch.manual_seed(4)
    m = ctn.ConvTasNet(64, 20, 64, 128, 3, 2, 3, 2, norm_type=norm, causal=causal).to(DEV)
    opt = FlatAdam(m.parameters(), lr=1e-3)
    mix, lens, src = O.synth_batch(40, 3, 4005)
    mix, lens, src = mix.to(DEV), lens.to(DEV), src.to(DEV)
    saved_side = ops._SIDE_ENABLED
    ops._SIDE_ENABLED = side
    try:
        grads = []
        for gb in (None, Stub()):
            ops.set_grad_buckets(gb)
            opt.zero_grad()
            ctn.cal_loss(src, m(mix), lens)[0].backward()
            ops.join_side_stream(opt.flat_grads.device)
            torch.cuda.synchronize()
            grads.append(opt.flat_grads.clone())
        assert Stub.calls == 3                              # X = 2 blocks per bucket, R = 3 repeats
        assert float(grads[0].abs().max()) > 0
        assert torch.equal(grads[0], grads[1]), "%d gradient elements differ" % int((grads[0] != grads[1]).sum())
    finally:
        ops.set_grad_buckets(None)
        ops._SIDE_ENABLED = saved_side


@pytest.mark.parametrize("side", [True, False])
@pytest.mark.parametrize("norm,causal", [("gLN", False), ("cLN", True)])
def test_chained_weight_gradients_equal_the_unchained(norm, causal, side):
    """Inside the composite stacks the split-K slabs of a weight gradient are summed by the next weight-gradient launch of the
    stream (ctn_tune("wgrad_chain", 1); opt-in) instead of a slab_reduce launch of their own: the same additions in the same
    order, so every gradient must be BITWISE the un-chained one's -- at widths where the split kernels run (B = 64, H = 128), an
    odd number of blocks per call, with the weight gradients on the second stream and on the main one."""
    from conv_tasnet_amd import ops
    torch.manual_seed(5)
    m = ctn.ConvTasNet(64, 20, 64, 128, 3, 3, 1, 2, norm_type=norm, causal=causal).to(DEV)
    opt = FlatAdam(m.parameters(), lr=1e-3)
    mix, lens, src = O.synth_batch(41, 3, 4005)
    mix, lens, src = mix.to(DEV), lens.to(DEV), src.to(DEV)
    saved_side = ops._SIDE_ENABLED
    ops._SIDE_ENABLED = side
    try:
        grads = []
        for chain in (0, 1, 1):
            ctn.lib.call("ctn_tune", b"wgrad_chain", chain)
            opt.zero_grad()
            ctn.cal_loss(src, m(mix), lens)[0].backward()
            ops.join_side_stream(opt.flat_grads.device)
            torch.cuda.synchronize()
            grads.append(opt.flat_grads.clone())
        assert float(grads[0].abs().max()) > 0
        assert torch.equal(grads[0], grads[1]), "%d gradient elements differ" % int((grads[0] != grads[1]).sum())
        assert torch.equal(grads[1], grads[2])
    finally:
        ctn.lib.call("ctn_tune", b"wgrad_chain", 0)
        ops._SIDE_ENABLED = saved_side


def test_flatadam_state_interchanges_with_torch_adam():
    g, m, batches = _traj_setup()
    opt = FlatAdam(m.parameters(), lr=1e-3)
    for mix, lens, src in batches[:2]:
        opt.zero_grad()
        loss = ctn.cal_loss(src.to(DEV), m(mix.to(DEV)), lens.to(DEV))[0]
        loss.backward()
        opt.step(max_grad_norm=5.0)
    sd = opt.state_dict()
    # a torch Adam over a copy of the model picks the state up and continues identically
    m2 = ctn.ConvTasNet(m.N, m.L, m.B, m.H, m.P, m.X, m.R, m.C).to(DEV)
    m2.load_state_dict(m.state_dict())
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3)
    opt2.load_state_dict(sd)
    mix, lens, src = batches[2]
    for mm, oo in ((m, opt), (m2, opt2)):
        oo.zero_grad()
        ctn.cal_loss(src.to(DEV), mm(mix.to(DEV)), lens.to(DEV))[0].backward()
        if oo is opt:
            oo.step(max_grad_norm=5.0)
        else:
            torch.nn.utils.clip_grad_norm_(mm.parameters(), 5.0)
            oo.step()
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert float((a - b).abs().max()) < 2e-6, k
    # and the other way round
    opt3 = FlatAdam(m2.parameters(), lr=1e-3)
    opt3.load_state_dict(opt2.state_dict())
    assert opt3._step == 3
    # grads live in one flat buffer (the all-reduce payload)
    p = next(m.parameters())
    assert p.grad.data_ptr() == opt.flat_grads.data_ptr()


def test_checkpoint_resume(tmp_path):
    g, m, batches = _traj_setup()
    opt = FlatAdam(m.parameters(), lr=1e-3)
    arg = (1, 2, 1, 0, 5, str(tmp_path), 1, "", "best.pth.tar", 1000, 0, 0, "x")
    Solver({"tr_loader": batches, "cv_loader": batches[:1]}, m, opt, arg).train()
    ck = tmp_path / "checkpoint_models" / "epoch2.pth.tar"
    assert ck.exists()
    m2 = ctn.ConvTasNet.load_model(str(ck)).to(DEV)
    opt2 = FlatAdam(m2.parameters(), lr=1e-3)
    arg2 = (1, 1, 1, 0, 5, str(tmp_path), 0, str(ck), "best2.pth.tar", 1000, 0, 0, "x")
    s2 = Solver({"tr_loader": batches, "cv_loader": batches[:1]}, m2, opt2, arg2)
    assert s2.start_epoch == 2 and s2.epochs == 1 + 2 + 1          # reference rule: epochs + start_epoch + 1
    assert opt2._step == 6
    s2.train()
    assert len(s2.iter_losses) == 2 * 4


def test_separate_writes_reference_named_files(tmp_path):
    from scipy.io import wavfile
    from conv_tasnet_amd.separate import separate
    torch.manual_seed(0)
    m = ctn.ConvTasNet(32, 20, 16, 32, 3, 2, 1, 2)
    path = str(tmp_path / "m.pth.tar")
    torch.save(ctn.ConvTasNet.serialize(m, torch.optim.Adam(m.parameters()), 1), path)
    mixdir = tmp_path / "mix"
    mixdir.mkdir()
    mix, _, _ = O.synth_batch(0, 2, 4000)
    wavfile.write(str(mixdir / "utt_a.wav"), 8000, mix[0].numpy())
    wavfile.write(str(mixdir / "utt_b.wav"), 8000, mix[1, :3000].numpy())
    out = tmp_path / "out"
    separate(path, str(mixdir), None, str(out), 1, 8000, 2)
    names = sorted(os.listdir(out))
    # 'utt_a.wav'.strip('.wav') == 'utt_' : the reference strips characters, not the suffix
    assert names == sorted(["utt_.wav", "utt__s1.wav", "utt__s2.wav", "utt_b.wav", "utt_b_s1.wav", "utt_b_s2.wav"])
    sr, s1 = wavfile.read(str(out / "utt_b_s1.wav"))
    assert sr == 8000 and len(s1) == 3000
    with torch.no_grad():
        padded = mix.clone()
        padded[1, 3000:] = 0          # the batch is zero-padded to its longest utterance, as in the reference loader
        ref = m.to(DEV)(padded.to(DEV))[1, 0, :3000].cpu().numpy()
    np.testing.assert_allclose(s1, ref, atol=1e-5)


def test_evaluate_sisnri_matches_oracle(capsys):
    from conv_tasnet_amd.evaluate import evaluate_loader as evaluate, cal_SISNR
    torch.manual_seed(1)
    m = ctn.ConvTasNet(32, 20, 16, 32, 3, 2, 1, 2).to(DEV)
    mix, lens, src = O.synth_batch(7, 2, 3000)
    lens = torch.tensor([3000, 2500])
    mix[1, 2500:] = 0
    src[1, :, 2500:] = 0
    got = evaluate(m, [(mix, lens, src)], use_cuda=True, verbose=False)
    cfg = O.Config(32, 20, 16, 32, 3, 2, 1, 2)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    est = O.forward(cfg, sd, mix)
    _, _, _, reord = O.cal_loss(src, est, lens)
    want = np.mean([O.cal_sisnri_np(src[b, :, :n].double().numpy(), reord[b, :, :n].double().numpy(),
                                    mix[b, :n].double().numpy()) for b, n in enumerate([3000, 2500])])
    assert abs(got - want) < 1e-3
    x = np.random.RandomState(0).randn(1000)
    assert abs(cal_SISNR(x, x) - O.cal_sisnr_np(x, x)) < 1e-9


def test_streaming_graph_replay_equals_eager_chunks():
    """StreamingSeparator(graph=True): from the third chunk of a length on, the whole chunk step is one HIP-graph replay (same kernels,
    same order, state in fixed buffers) -- every output chunk must be BITWISE the eager separator's, across a chunk of another
    length in mid-stream, the flush and a reset()."""
    from conv_tasnet_amd.streaming import StreamingSeparator
    torch.manual_seed(3)
    m = ctn.ConvTasNet(32, 20, 16, 32, 3, 4, 2, 2, norm_type="cLN", causal=True).to(DEV).eval()
    S = 10
    plan = [40, 40, 40, 40, 7, 40, 40, 7, 7, 7, 40]            # hops per chunk: 40 and 7 both reach their replay
    T = S * sum(plan)
    mix, _, _ = O.synth_batch(6, 2, T)
    eager, graphed = StreamingSeparator(m, batch=2), StreamingSeparator(m, batch=2, graph=True)
    for rnd in range(2):
        pos = 0
        for n in plan:
            a, b = eager.push(mix[:, pos:pos + n * S]), graphed.push(mix[:, pos:pos + n * S])
            assert torch.equal(a, b), (rnd, pos)
            pos += n * S
        assert torch.equal(eager.flush(), graphed.flush())
        assert sorted(graphed._graphs) == [7 * S, 40 * S]
        eager.reset()
        graphed.reset()                                        # the captured graphs stay valid: the state buffers are the same


def test_streaming_causal_inference_equals_full_forward():
    """chunk-by-chunk separation with carried state == one forward over the whole signal."""
    from conv_tasnet_amd.streaming import StreamingSeparator
    torch.manual_seed(2)
    m = ctn.ConvTasNet(32, 20, 16, 32, 3, 4, 2, 2, norm_type="cLN", causal=True).to(DEV).eval()
    S, T = 10, 10 * 537
    mix, _, _ = O.synth_batch(3, 2, T)
    with torch.no_grad():
        full = m(mix.to(DEV))                                  # [2, 2, T]
    s = StreamingSeparator(m, batch=2)
    outs, pos = [], 0
    for n in (40, 7, 133, 2, 64, 291):                         # ragged chunk sizes, in hops
        outs.append(s.push(mix[:, pos:pos + n * S]))
        pos += n * S
    assert pos == T
    outs.append(s.flush())
    got = torch.cat(outs, dim=2)
    assert got.shape == full.shape
    assert float((got - full).abs().max()) <= 2e-6 * float(full.abs().max())
    with pytest.raises(ValueError):
        StreamingSeparator(ctn.ConvTasNet(32, 20, 16, 32, 3, 2, 1, 2).to(DEV))
    # ... and against the REFERENCE's own output: the causal cLN fixture recorded from src/conv_tasnet.py (oracle/make_golden.py).
    # Its 3001 samples give 299 frames over samples 0 .. 2999 (the 3001st output sample is the reference's zero padding), exactly
    # the frames of a 3000-sample stream.
    gd = load_golden("model_tiny_cln_causal")
    N, L, B, H, P, X, R, C = [int(v) for v in gd["cfg"]]
    mg = ctn.ConvTasNet(N, L, B, H, P, X, R, C, norm_type="cLN", causal=True, mask_nonlinear=str(gd["mask_nonlinear"]))
    mg.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in gd.items() if k.startswith("p:")})
    mg = mg.to(DEV).eval()
    mixg, refg = torch.from_numpy(gd["mixture"])[:, :3000], torch.from_numpy(gd["est_source_raw"])[..., :3000]
    sg = StreamingSeparator(mg, batch=2)
    outs, pos = [], 0
    for n in (17, 100, 3, 180):
        outs.append(sg.push(mixg[:, pos:pos + n * (L // 2)]))
        pos += n * (L // 2)
    assert pos == 3000
    outs.append(sg.flush())
    gotg = torch.cat(outs, dim=2).cpu()
    assert gotg.shape == refg.shape
    assert float((gotg - refg).abs().max()) <= 2e-5 * float(refg.abs().max())


def test_input_dtype_and_layout_are_normalised():
    torch.manual_seed(0)
    m = ctn.ConvTasNet(32, 20, 16, 32, 3, 2, 1, 2).to(DEV).eval()
    mix, _, _ = O.synth_batch(0, 2, 1500)
    with torch.no_grad():
        ref = m(mix.to(DEV))
        assert torch.equal(m(mix.double().to(DEV)), ref)                      # float64 input is cast, not reinterpreted
        wide = torch.zeros(2, 3000)
        wide[:, ::2] = mix
        assert torch.equal(m(wide.to(DEV)[:, ::2]), ref)                      # strided view
    with pytest.raises(ValueError):
        m(torch.zeros(1, 10, device=DEV))


def test_double_precision_model_is_rejected_not_reinterpreted():
    m = ctn.ConvTasNet(32, 20, 16, 32, 3, 2, 1, 2).to(DEV).double()
    with pytest.raises(ctn.CtnError, match="fp32"):
        m(torch.zeros(1, 400, device=DEV))


def test_graphed_backprop_replays_the_eager_step():
    """HIP-graph replay of zero_grad+fwd+loss+bwd gives bit-identical gradients and parameter trajectories."""
    from conv_tasnet_amd.graphed import GraphedBackprop
    from conv_tasnet_amd.train import SyntheticLoader
    cfg = dict(N=64, L=20, B=32, H=64, P=3, X=3, R=2, C=2)
    batches = list(SyntheticLoader(3, 2, samples=8000))

    class Buckets:                  # what parallel.enable_overlap installs for N > 1: must not reach the capture
        blocks_per_bucket = 3
        works, covered = [], []

        def bucket_ready(self, sinks):
            raise AssertionError("a gradient bucket was issued inside the warm-up or the capture of GraphedBackprop")

    def run(graph):
        torch.manual_seed(3)
        model = ctn.ConvTasNet(**cfg).to(DEV)
        opt = FlatAdam(model.parameters(), lr=1e-3)
        if graph:
            from conv_tasnet_amd import ops
            opt._ctn_buckets = Buckets()
            ops.set_grad_buckets(opt._ctn_buckets)
        stepper = GraphedBackprop(model, opt, batches[0]) if graph else None
        if graph:
            assert ops._GRAD_BUCKETS is None and opt._ctn_buckets is None      # one collective outside the graph instead
        losses, grads = [], []
        for mix, lens, src in batches:
            mix, lens, src = mix.to(DEV), lens.to(DEV), src.to(DEV)
            if graph:
                loss = stepper(mix, lens, src)
            else:
                opt.zero_grad()
                loss = ctn.cal_loss(src, model(mix), lens)[0]
                loss.backward()
            losses.append(float(loss.detach()))
            opt.step(max_grad_norm=5.0)
            grads.append(opt.flat_grads.clone())
        return losses, grads, opt.flat_params.clone()

    l0, g0, p0 = run(False)
    l1, g1, p1 = run(True)
    assert l0 == l1
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    assert torch.equal(p0, p1)
    with pytest.raises(ValueError):
        torch.manual_seed(3)
        model = ctn.ConvTasNet(**cfg).to(DEV)
        opt = FlatAdam(model.parameters(), lr=1e-3)
        st = GraphedBackprop(model, opt, batches[0])
        st(batches[0][0][:1].to(DEV), batches[0][1][:1].to(DEV), batches[0][2][:1].to(DEV))


@pytest.mark.parametrize("flat", [True, False])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("norm_type", ["gLN", "cLN"])
@pytest.mark.parametrize("wide", [False, True])
def test_composite_stack_is_bitwise_the_per_kernel_path(flat, causal, norm_type, wide, monkeypatch):
    """ctn_tcn_{gln,cln}_fwd / _bwd (one C call per direction for the whole TemporalBlock stack, weight gradients on
    the second stream) against the per-kernel entry points driven block by block from Python: outputs, loss and every
    gradient must be bitwise equal -- the composite only moves the host side of the launches into C++.  wide: channel
    counts the split-bf16 kernels take (>= 64 rows, pre-split weight pieces)."""
    from conv_tasnet_amd import ops
    mix, lens, src = O.synth_batch(5, 3, 4000 + 7)
    res = []
    for composite in (True, False):
        monkeypatch.setattr(ops, "_COMPOSITE", composite)
        torch.manual_seed(3)
        dims = (64, 20, 64, 128, 3, 3, 2, 2) if wide else (32, 20, 16, 32, 3, 4, 2, 2)
        m = ctn.ConvTasNet(*dims, norm_type=norm_type, causal=causal).to(DEV)
        opt = FlatAdam(m.parameters(), lr=1e-3) if flat else None
        if opt is not None:
            opt.zero_grad()
        est = m(mix.to(DEV))
        loss = ctn.cal_loss(src.to(DEV), est, lens.to(DEV))[0]
        loss.backward()
        ops.join_side_stream()
        torch.cuda.synchronize()
        res.append((est.detach().clone(), loss.detach().clone(), [p.grad.detach().clone() for p in m.parameters()]))
        with torch.no_grad():                      # the inference form (ping-pong slots) gives the same output too
            assert torch.equal(m(mix.to(DEV)), res[-1][0])
    (e1, l1, g1), (e2, l2, g2) = res
    assert torch.equal(e1, e2) and torch.equal(l1, l2)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("level", [0, 1])
def test_cln_composite_is_bitwise_the_per_kernel_path_at_every_fusion_level(level, monkeypatch):
    """The same for the cLN stack under ctn_tune("cln_fuse", 0 | 1) (2, the default, is covered above): the un-fused norm passes
    and the backward-only fusion stay selectable, and the composite and ops.ClnBlock follow the same rule."""
    from conv_tasnet_amd import ops
    mix, lens, src = O.synth_batch(5, 3, 4000 + 7)
    ctn.lib.call("ctn_tune", b"cln_fuse", level)
    try:
        res = []
        for composite in (True, False):
            monkeypatch.setattr(ops, "_COMPOSITE", composite)
            torch.manual_seed(3)
            m = ctn.ConvTasNet(64, 20, 64, 128, 3, 3, 2, 2, norm_type="cLN", causal=True).to(DEV)
            opt = FlatAdam(m.parameters(), lr=1e-3)
            opt.zero_grad()
            est = m(mix.to(DEV))
            loss = ctn.cal_loss(src.to(DEV), est, lens.to(DEV))[0]
            loss.backward()
            ops.join_side_stream()
            torch.cuda.synchronize()
            res.append((est.detach().clone(), loss.detach().clone(), [p.grad.detach().clone() for p in m.parameters()]))
        (e1, l1, g1), (e2, l2, g2) = res
        assert torch.equal(e1, e2) and torch.equal(l1, l2)
        for a, b in zip(g1, g2):
            assert torch.equal(a, b)
    finally:
        ctn.lib.call("ctn_tune", b"cln_fuse", 2)
        ops._ws_cache.clear()


@pytest.mark.parametrize("norm_type,causal", [("gLN", False), ("cLN", True)])
def test_one_fork_per_block_backward_is_bitwise_the_two_fork_schedule(norm_type, causal):
    """ctn_tune("bwd_events", 1): the composite backward forks the weight-gradient stream once per block (behind B5: dW1 and the sums
    of that block, then dW2 of the NEXT block, which needs only that B5's output) instead of twice.  Only the queueing changes:
    every gradient is bitwise the two-fork schedule's."""
    from conv_tasnet_amd import ops
    mix, lens, src = O.synth_batch(5, 3, 4000 + 7)
    res = []
    try:
        for ev in (2, 1, 1):
            ctn.lib.call("ctn_tune", b"bwd_events", ev)
            torch.manual_seed(3)
            m = ctn.ConvTasNet(64, 20, 64, 128, 3, 3, 2, 2, norm_type=norm_type, causal=causal).to(DEV)
            opt = FlatAdam(m.parameters(), lr=1e-3)
            opt.zero_grad()
            loss = ctn.cal_loss(src.to(DEV), m(mix.to(DEV)), lens.to(DEV))[0]
            loss.backward()
            ops.join_side_stream()
            torch.cuda.synchronize()
            res.append([p.grad.detach().clone() for p in m.parameters()])
    finally:
        ctn.lib.call("ctn_tune", b"bwd_events", 0)           # the default: per-stack choice
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)


def test_cln_fusion_level_must_not_change_between_forward_and_backward():
    """Level 2 of ctn_tune("cln_fuse") never stores the first norm's output: a backward pass under a lower level would read a
    buffer that was never written -- the composite stack refuses it."""
    if ctn.lib.ctn_gemm_arith() != 3:
        pytest.skip("one arithmetic is enough")
    mix, lens, src = O.synth_batch(5, 2, 3000)
    torch.manual_seed(3)
    m = ctn.ConvTasNet(32, 20, 16, 32, 3, 2, 1, 2, norm_type="cLN", causal=True).to(DEV)
    loss = ctn.cal_loss(src.to(DEV), m(mix.to(DEV)), lens.to(DEV))[0]
    ctn.lib.call("ctn_tune", b"cln_fuse", 0)
    try:
        with pytest.raises(ctn.CtnError, match="cln_fuse"):
            loss.backward()
    finally:
        ctn.lib.call("ctn_tune", b"cln_fuse", 2)


def test_evaluate_with_the_reference_signature(tmp_path, capsys):
    """evaluate(model_path, data_dir, calc_sdr, use_cuda, sample_rate, batch_size), src/evaluate.py:21: checkpoint file +
    {mix,s1,s2}.json manifests of wav files in, average SI-SNRi out; calc_sdr (mir_eval) is an explicit error."""
    import json
    from scipy.io import wavfile
    from conv_tasnet_amd.evaluate import evaluate, evaluate_loader
    torch.manual_seed(3)
    m = ctn.ConvTasNet(32, 20, 16, 32, 3, 2, 1, 2).to(DEV)
    path = str(tmp_path / "m.pth.tar")
    torch.save(m.serialize(m, torch.optim.Adam(m.parameters()), 1), path)
    mix, lens, src = O.synth_batch(11, 3, 2400)
    src = src / src.abs().max() * 0.4
    manifests = {"mix": [], "s1": [], "s2": []}
    for u in range(3):
        n = 2400 - 300 * u
        sig = {"s1": src[u, 0, :n], "s2": src[u, 1, :n]}
        sig["mix"] = sig["s1"] + sig["s2"]
        for k, v in sig.items():
            p = str(tmp_path / ("%s_%d.wav" % (k, u)))
            wavfile.write(p, 8000, (v.numpy() * 32767).astype(np.int16))
            manifests[k].append([p, n])
    for k, v in manifests.items():
        (tmp_path / (k + ".json")).write_text(json.dumps(v))
    got = evaluate(path, str(tmp_path), 0, 1, 8000, 2)
    from conv_tasnet_amd.data import AudioDataLoader, AudioDataset
    want = evaluate_loader(m, AudioDataLoader(AudioDataset(str(tmp_path), 2, sample_rate=8000, segment=-1)), verbose=False)
    assert abs(got - want) < 1e-6 and "Average SISNR improvement" in capsys.readouterr().out
    with pytest.raises(NotImplementedError):
        evaluate(path, str(tmp_path), 1, 1, 8000, 2)


def test_direct_gradients_refuse_silent_overwrite_and_are_readable_after_backward():
    """FlatAdam(direct_grads=True): the backward stages overwrite the flat gradient buffer.  A second backward pass without
    zero_grad() must raise (the reference's autograd would accumulate), and right after loss.backward() the gradients are
    complete in stream order (the weight-gradient stream is joined by the last backward node), so reading them is safe."""
    from conv_tasnet_amd import CtnError
    torch.manual_seed(5)
    m = ctn.ConvTasNet(32, 20, 16, 32, 3, 2, 2, 2).to(DEV)
    opt = FlatAdam(m.parameters(), lr=1e-3)
    mix, lens, src = O.synth_batch(21, 2, 3000)

    def backprop():
        ctn.cal_loss(src.to(DEV), m(mix.to(DEV)), lens.to(DEV))[0].backward()

    opt.zero_grad()
    backprop()
    g1 = opt.flat_grads.clone()                 # same stream, no explicit join: must already be final
    torch.cuda.synchronize()
    assert torch.equal(g1, opt.flat_grads) and float(g1.abs().max()) > 0
    with pytest.raises(CtnError):
        backprop()
    opt.zero_grad()
    backprop()                                  # fine again after zero_grad()
    torch.cuda.synchronize()
    assert torch.equal(g1, opt.flat_grads)


def test_library_probe_brackets_every_launch_group_of_the_stacks():
    """ctn_probe_enable / ctn_probe_read (bench.py's roofline leg): one record per launch group of the composite stacks in
    issue order, family ids in range, positive durations; reading ends the recording."""
    import ctypes
    from conv_tasnet_amd import ops
    if not ops._COMPOSITE:
        pytest.skip("per-kernel sequencing selected")
    torch.manual_seed(0)
    m = ctn.ConvTasNet(32, 20, 16, 32, 3, 3, 2, 2).to(DEV)          # 6 gLN blocks
    mix, lens, src = O.synth_batch(5, 2, 4000)
    opt = FlatAdam(m.parameters(), lr=1e-3)
    opt.zero_grad()
    ctn.lib.call("ctn_probe_enable", 1)
    ctn.cal_loss(src.to(DEV), m(mix.to(DEV)), lens.to(DEV))[0].backward()
    torch.cuda.synchronize()
    fam, us = (ctypes.c_int * 512)(), (ctypes.c_float * 512)()
    n = ctn.lib.load().ctn_probe_read(fam, us, 512)
    # per block: K1 K2 K3 forward (each twice when the forward pass runs as two half-batch chains); B1 B2 B3 B4 B5 B6 finalize
    # backward; plus the weight preparation launches
    ids = [fam[i] for i in range(n)]
    assert n >= 6 * 10 and all(0 <= f <= 14 for f in ids) and all(us[i] > 0 for i in range(n))
    chains = 2 if (ops._SIDE_ENABLED and ops._FWD_DUAL) else 1
    for f in (0, 1, 2):
        assert ids.count(f) == 6 * chains, (f, ids.count(f))
    for f in (3, 4, 5, 6, 7, 8, 9):
        assert ids.count(f) == 6, (f, ids.count(f))
    assert ids.count(14) == 0           # (14: the slab_reduce that ends a chain of weight gradients, ctn_tune("wgrad_chain", 1))
    assert ctn.lib.load().ctn_probe_read(fam, us, 512) == 0         # recording ended
    m(mix.to(DEV))
    assert ctn.lib.load().ctn_probe_read(fam, us, 512) == 0         # and off
