#!/usr/bin/env python
"""Conv-TasNet training-step throughput on MI355X (BASELINE.json metric: 4 s / 8 kHz / 2-spk utterances per second, fwd+bwd).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one synthetic minibatch resident in HBM: forward, PIT SI-SNR loss, backward,
(N>1: one RCCL all-reduce of the flat gradient), clip_grad_norm(5) + Adam -- the step of src/solver.py:188-196.
Workload: BASELINE configs[1], paper config N256 L20 B256 H512 P3 X8 R4 gLN C2, 8 utterances of 4 s @ 8 kHz per GPU
(weak scaling: global batch = 8 * N).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PAPER = dict(N=256, L=20, B=256, H=512, P=3, X=8, R=4, C=2)
PER_GPU_BATCH = 8
T_SAMPLES = 32000


def flops_fwd(c, T):
    """F_fwd per utterance, BASELINE.md section 3."""
    K = (T - c["L"]) // (c["L"] // 2) + 1
    return 2 * K * (c["N"] * c["L"] + c["N"] * c["B"] + c["X"] * c["R"] * (2 * c["B"] * c["H"] + c["H"] * c["P"])
                    + c["B"] * c["C"] * c["N"] + c["C"] * c["N"] * c["L"]), K


def cpu_baseline(max_seconds=30.0):
    """The oracle (torch CPU restatement of the reference step) on this host's cores: bounded sample."""
    from oracle import ctn_oracle as O
    cfg = O.Config(**PAPER)
    torch.set_num_threads(min(16, torch.get_num_threads()))     # the 1-GPU box's CPU share
    threads = torch.get_num_threads()
    sd = O.init_params(cfg, seed=0)
    state = {}
    mix, lens, src = O.synth_batch(0, 1, T_SAMPLES)
    O.train_step(cfg, sd, state, mix, src, lens)            # warm-up (allocator, thread pool)
    n, t0 = 0, time.perf_counter()
    while True:
        O.train_step(cfg, sd, state, mix, src, lens)
        n += 1
        dt = time.perf_counter() - t0
        if n >= 3 or dt > max_seconds:
            break
    return {"value": round(n / dt, 4), "unit": "utterances/sec", "cores": threads, "kind": "port",
            "sample": "%d full training steps (fwd+loss+bwd+clip+Adam) of the paper config on 1 utterance of 4 s, "
                      "oracle/ctn_oracle.py with torch CPU ops, %d threads, after 1 warm-up step" % (n, threads)}


def dominant_kernel_roofline(ctn, device, K, iters=30, in_step_us=None, in_step_n=0, in_step_steps=0):
    """Time the dominant kernel (the 1x1-conv fp32-MFMA GEMM, B->H with the fused PReLU/gLN-statistics epilogue)
    at the workload's shape with HIP events on the stream it is launched on."""
    from conv_tasnet_amd import ops
    M, B, H = PER_GPU_BATCH, PAPER["B"], PAPER["H"]
    Kp = ops.padded_frames(K)
    x = torch.randn(M, B, Kp, device=device)
    x[..., K:] = 0
    W = torch.randn(H, B, device=device) * 0.05
    a = torch.full((1,), 0.25, device=device)
    for _ in range(3):
        ops.pw_gemm(W, x, H, B, K, epi_alpha=a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()                      # torch's current stream == the stream ops._stream() hands to the C ABI
    for _ in range(iters):
        ops.pw_gemm(W, x, H, B, K, epi_alpha=a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flop = 2.0 * H * B * M * K       # algorithmic: 2*H*B per frame, K frames per utterance, M utterances per launch
    ach = flop / (ms * 1e-3) / 1e12
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {"bound": "mfma", "kernel": "pw_gemm_kernel<0,PRO_NONE,EPI_PRELU_STATS> (1x1 conv B->H, fp32 MFMA)",
           "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
           "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "us_per_launch": round(ms * 1e3, 2),
           "flop_per_launch": flop, "traffic": traffic, "how": "%d back-to-back launches after the timed steps" % iters}
    if in_step_us is not None:       # the same kernel measured inside real training steps (one event pair per launch)
        ach2 = flop / (in_step_us * 1e-6) / 1e12
        out.update({"achieved": round(ach2, 2), "frac": round(ach2 / PEAK_F32_MFMA_TFLOPS, 4),
                    "us_per_launch": round(in_step_us, 2), "how": "HIP-event pair around each of the %d launches of this "
                    "kernel in %d extra training steps after the timed region" % (in_step_n, in_step_steps),
                    "back_to_back": {"us_per_launch": round(ms * 1e3, 2), "achieved": round(ach, 2),
                                     "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "launches": iters}})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("CTN_BENCH_GRAPH", "0")),
                    help="1: replay zero_grad+fwd+loss+bwd from one captured HIP graph (conv_tasnet_amd.graphed)")
    ap.add_argument("--gemm", choices=["fp32", "x6"], default=None,
                    help="1x1-conv arithmetic: fp32 MFMA (default, bit-exact fp32 chains) or split-bf16 emulation (experimental)")
    args = ap.parse_args()

    import torch.distributed as dist
    import conv_tasnet_amd as ctn
    from conv_tasnet_amd import parallel
    from conv_tasnet_amd.optim import FlatAdam
    from conv_tasnet_amd.train import SyntheticLoader   # synthetic workload of SURVEY 8d (product code, not the oracle)

    if args.gemm:
        from conv_tasnet_amd import ops as _ops
        _ops.set_gemm_mode(args.gemm)
    world, rank, device = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, world))
    if device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")

    torch.manual_seed(0)
    model = ctn.ConvTasNet(**PAPER, norm_type="gLN", causal=False, mask_nonlinear="relu").to(device)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    parallel.broadcast_parameters(opt.flat_params)
    # this rank's shard: utterances [rank*8, rank*8+8) of the deterministic harmonic-mixture workload
    mix, lens, src = next(iter(SyntheticLoader(1, PER_GPU_BATCH, samples=T_SAMPLES, rank=rank, world=world)))
    mix, lens, src = mix.to(device), lens.to(device), src.to(device)
    loss_acc = torch.zeros((), device=device)

    graphed = None
    if args.graph:
        from conv_tasnet_amd.graphed import GraphedBackprop
        graphed = GraphedBackprop(model, opt, (mix, lens, src))

    def step():
        if graphed is not None:
            loss = graphed(mix, lens, src)
        else:
            opt.zero_grad()
            est = model(mix)
            loss, _, _, _ = ctn.cal_loss(src, est, lens)
            loss.backward()
        scale = parallel.allreduce_gradients(opt)
        opt.step(max_grad_norm=5.0, grad_scale=scale)
        loss_acc.add_(loss.detach())

    for _ in range(args.warmup):
        step()
    loss_acc.zero_()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_issue = time.perf_counter() - t0          # host time to enqueue the timed steps (diagnostic only)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    mean_loss = float(loss_acc) / max(args.steps, 1)

    if rank == 0:
        ffwd, K = flops_fwd(PAPER, T_SAMPLES)
        ftrain = 3 * ffwd - 4 * K * PAPER["N"] * PAPER["L"]
        utt = PER_GPU_BATCH * world * args.steps
        value = utt / dt
        out = {
            "metric": "4s 8kHz 2-spk utterances/sec (fwd+bwd)", "value": round(value, 2), "unit": "utterances/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: paper config N256 L20 B256 H512 P3 X8 R4 gLN non-causal C2, "
                                   "%d x 4s@8kHz utterances per GPU, fwd+PIT-loss+bwd+clip(5)+Adam" % PER_GPU_BATCH,
                       "global_batch": PER_GPU_BATCH * world, "samples_per_utterance": T_SAMPLES,
                       "parallelism": "dp%d" % world},
            "gemm_mode": __import__("conv_tasnet_amd").ops.gemm_mode(), "hip_graph": bool(args.graph),
            "mean_loss": round(mean_loss, 4), "host_issue_ms_per_step": round(1e3 * t_issue / args.steps, 3),
            "model_tflops": round(value * ftrain / 1e12, 2),
            "model_frac_of_f32_mfma_peak": round(value * ftrain / 1e12 / (PEAK_F32_MFMA_TFLOPS * world), 4),
        }
        if world == 1:
            # the dominant kernel inside real steps: a few extra steps with an event pair around each of its launches
            from conv_tasnet_amd import ops as _o
            probe, probe_steps = [], 3
            _o.set_stats_gemm_probe(probe)
            for _ in range(probe_steps):
                step()
            _o.set_stats_gemm_probe(None)
            torch.cuda.synchronize()
            M_, B_, H_ = PER_GPU_BATCH, PAPER["B"], PAPER["H"]
            us = [1e3 * e0.elapsed_time(e1) for e0, e1, m, r, cn, k in probe if (m, r, cn) == (M_, H_, B_)]
            in_us = sum(us) / len(us) if us else None
            out["roofline"] = dominant_kernel_roofline(ctn, device, K, in_step_us=in_us, in_step_n=len(us), in_step_steps=probe_steps)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
