#!/usr/bin/env python
"""Conv-TasNet training-step throughput on MI355X (BASELINE.json metric: 4 s / 8 kHz / 2-spk utterances per second, fwd+bwd).

    python bench.py [--gpus N --steps K --warmup W] [--config paper|causal|c3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one synthetic minibatch resident in HBM: forward, PIT SI-SNR loss, backward,
(N>1: one RCCL all-reduce of the flat gradient), clip_grad_norm(5) + Adam -- the step of src/solver.py:188-196.
Default workload: BASELINE configs[1], paper config N256 L20 B256 H512 P3 X8 R4 gLN C2, 8 utterances of 4 s @ 8 kHz per
GPU (weak scaling: global batch = 8 * N).  --config causal = configs[3] (cLN, causal), --config c3 = configs[4]
(3 speakers, L=16, 16 kHz).  Rank 0 prints ONE JSON line.

`roofline` (rank 0): after the timed region a few more training steps run with a HIP-event pair around EVERY launch
group of the block kernels (the per-kernel entry points driven from Python: the same kernels in the same order as the
composite C calls of the timed region, each bracketed on the stream it is launched on).  The line reports the family
with the largest total time (`roofline.kernel`) and the whole table (`roofline.families`).  `cpu_baseline` (rank 0):
the oracle's training step on the host cores, bounded in wall time.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

_T_PROC = time.time()         # the driver's clock runs from process start: the CPU leg is bounded against it (WALL_LIMIT_S)
WALL_LIMIT_S = 540.0          # whole default run, import torch on a cold box included (the driver allows 600 s)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)" (154.5 measured on the box:
                               # benchmarks/mfma_probe.hip, profiles/r02_a_mfma_probe.txt)
PEAK_BF16_MFMA_TFLOPS = 2516.6  # same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense" (256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz)
PEAK_HBM_GBS = 8000.0          # same guide, HBM3E peak
ARITH = {"name": "h3"}         # GEMM arithmetic of this run (--arith): "h3" (library default) = the composite stacks on two fp16 pieces per
                               # fp32 operand under tracked power-of-two scales, 3 f16 MFMAs per product step (fp32-faithful products; every
                               # other GEMM as b6), "b6" = three bf16 pieces, 6 bf16 MFMAs (fp32-faithful), "fp32" = fp32 MFMA
ARITH_IDS = {"fp32": 0, "b6": 2, "h3": 3}
# `dtype` of the JSON line: the arithmetic type the path computes in (not a precision claim: tests/test_gpu_h3.py holds those)
DTYPE_TEXT = {"h3": "f32 (1x1-conv GEMMs of the stacks as h3: two fp16 pieces per f32 operand under tracked power-of-two scales, three "
                    "f16 MFMAs, f32 accumulate; other GEMMs b6: three bf16 pieces, six bf16 MFMAs; everything else f32, statistics f64)",
              "b6": "f32 (1x1-conv GEMMs as b6: three bf16 pieces per f32 operand, six bf16 MFMAs, f32 accumulate; everything else f32, statistics f64)",
              "fp32": "f32 (f32 MFMA, bit-exact f32 FMA chains; statistics f64)"}
MFMA_PER_STEP = {"b6": 6, "h3": 3}
PER_GPU_BATCH = 8
CONFIGS = {
    "paper": dict(model=dict(N=256, L=20, B=256, H=512, P=3, X=8, R=4, C=2), norm_type="gLN", causal=False, T=32000, sr=8000,
                  name="BASELINE configs[1]: paper config N256 L20 B256 H512 P3 X8 R4 gLN non-causal C2, %d x 4s@8kHz"),
    "causal": dict(model=dict(N=256, L=20, B=256, H=512, P=3, X=8, R=4, C=2), norm_type="cLN", causal=True, T=32000, sr=8000,
                   name="BASELINE configs[3]: causal cLN variant N256 L20 B256 H512 P3 X8 R4 C2, %d x 4s@8kHz"),
    "c3": dict(model=dict(N=256, L=16, B=256, H=512, P=3, X=8, R=4, C=3), norm_type="gLN", causal=False, T=64000, sr=16000,
               name="BASELINE configs[4]: 3-speaker C3 L16 N256 B256 H512 P3 X8 R4 gLN, %d x 4s@16kHz"),
    # the reference's own CPU-runnable case (a parity-test config, and the N > 1 rehearsal of tests/test_bench_dp_gpu.py): not a bench line
    "tiny": dict(model=dict(N=64, L=20, B=32, H=64, P=3, X=2, R=2, C=2), norm_type="gLN", causal=False, T=8000, sr=8000, batch=2,
                 name="BASELINE configs[0]: tiny N64 L20 B32 H64 P3 X2 R2 gLN C2, %d x 1s@8kHz"),
}


def flops_fwd(c, T):
    """F_fwd per utterance, BASELINE.md section 3."""
    K = (T - c["L"]) // (c["L"] // 2) + 1
    return 2 * K * (c["N"] * c["L"] + c["N"] * c["B"] + c["X"] * c["R"] * (2 * c["B"] * c["H"] + c["H"] * c["P"])
                    + c["B"] * c["C"] * c["N"] + c["C"] * c["N"] * c["L"]), K


# ---------------------------------------------------------------------------------------------------------------------
# cpu_baseline: BASELINE.md section 4
# ---------------------------------------------------------------------------------------------------------------------
def _host_cpu():
    """(model string, cores this process may really use, how that was found): physical cores of /proc/cpuinfo, capped by
    the affinity mask and by the cgroup CPU quota (a 1-GPU box of the pool gets a share of its host's cores: running one
    thread per physical core of the HOST on that share is slower than the share -- measured 77 s vs 25 s per step)."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    limits = {"physical_cores": len(cores) or (os.cpu_count() or 1)}
    if hasattr(os, "sched_getaffinity"):
        limits["affinity"] = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    limits["cgroup_quota"] = max(1, int(int(txt[0]) / int(txt[1])))
            elif int(txt[0]) > 0:
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                limits["cgroup_quota"] = max(1, int(int(txt[0]) / period))
            break
        except (OSError, ValueError, IndexError):
            continue
    env = os.environ.get("CTN_CPU_BASELINE_THREADS")
    if env:
        limits["CTN_CPU_BASELINE_THREADS"] = int(env)
    n = min(limits.values())
    return model, max(n, 1), ", ".join("%s=%d" % kv for kv in limits.items())


def cpu_baseline(cfg, budget_s=240.0):
    # never past the whole-run limit: what is left of it, minus the 8-thread leg (~3 steps) -- at least 3 timed steps are taken
    left = WALL_LIMIT_S - (time.time() - _T_PROC)
    budget_s = max(60.0, min(budget_s, left - 150.0))
    """The oracle (torch CPU restatement of the reference step: fwd + loss + bwd + clip(5) + Adam) on this host, batch 8,
    all physical cores this process may use; then the same on 8 threads (the survey container's count).  BASELINE.md section 4:
    3 warm-up + >= 10 timed steps, median and min.  A paper-config step takes ~20 s on a 1-GPU box's 16-core share, so the
    sample is bounded in wall time: 3 warm-up steps (2 when a step exceeds 25 s), then timed steps until 10 are in, or until at
    least 5 are in and the wall budget is spent."""
    from oracle import ctn_oracle as O
    ocfg = O.Config(norm_type=cfg["norm_type"], causal=cfg["causal"], **cfg["model"])
    model, cores, how = _host_cpu()
    mix, lens, src = O.synth_batch(0, PER_GPU_BATCH, cfg["T"], C=cfg["model"]["C"], sr=cfg["sr"])

    def run(threads, budget, max_warm, min_steps, max_steps):
        torch.set_num_threads(threads)
        sd, state = O.init_params(ocfg, seed=0), {}
        t_start, times, warm = time.perf_counter(), [], 0
        while True:
            t0 = time.perf_counter()
            O.train_step(ocfg, sd, state, mix, src, lens)
            dt = time.perf_counter() - t0
            elapsed = time.perf_counter() - t_start
            if warm < (max_warm if dt <= 25.0 else min(max_warm, 2)):
                warm += 1                      # allocator, thread pool, first-touch of 18 GB of saved activations
                continue
            times.append(dt)
            if len(times) >= max_steps or (len(times) >= min_steps and elapsed + dt > budget) or \
                    (len(times) >= 3 and time.time() - _T_PROC + dt > WALL_LIMIT_S - 60.0):
                return warm, times

    warm, times = run(cores, budget_s, 3, 5, 10)
    med, best = statistics.median(times), min(times)
    out = {"value": round(PER_GPU_BATCH / med, 4), "unit": "utterances/sec", "cores": cores, "kind": "port",
           "cpu_model": model, "cores_from": how, "batch": PER_GPU_BATCH, "warmup_steps": warm, "timed_steps": len(times),
           "median_s_per_step": round(med, 3), "min_s_per_step": round(best, 3),
           "value_at_min": round(PER_GPU_BATCH / best, 4),
           "sample": "%d timed full training steps (fwd+PIT loss+bwd+clip(5)+Adam) of the workload's config on its batch of "
                     "%d utterances after %d warm-up step(s), oracle/ctn_oracle.py with torch CPU ops on %d threads (every "
                     "core usable by the process: min of physical cores, affinity mask, cgroup quota), wall budget %.0f s; "
                     "value = batch / median step time"
                     % (len(times), PER_GPU_BATCH, warm, cores, budget_s)}
    if cores != 8 and time.time() - _T_PROC + 3.5 * med < WALL_LIMIT_S:
        w8, t8 = run(min(8, cores), budget_s * 0.2, 1, 2, 3)
        out["threads8"] = {"value": round(PER_GPU_BATCH / statistics.median(t8), 4), "timed_steps": len(t8), "warmup_steps": w8,
                           "threads": min(8, cores), "min_s_per_step": round(min(t8), 3)}
    return out


# ---------------------------------------------------------------------------------------------------------------------
# roofline: per-family in-step timing of the block kernels
# ---------------------------------------------------------------------------------------------------------------------
def _family(name, a):
    """Classify one probed lib.call (entry point + arguments, include/ctn_hip.h) into a kernel family of DESIGN.md."""
    if name == "ctn_pw_gemm":
        M, R, Cn, trans_w, pro, residual, epi_part, relu = a[3], a[4], a[5], a[8], a[9], a[15], a[17], a[18]
        shape = (M, R, Cn)
        if pro:
            return "K3 pw_gemm<PRO_PRELU_NORM,EPI_RESIDUAL> (1x1 H->B, gLN prologue + residual)", "mfma", shape
        if epi_part:
            return "K1 pw_gemm<EPI_PRELU_STATS> (1x1 B->H + PReLU/gLN statistics)", "mfma", shape
        if trans_w and residual:
            return "B5 pw_gemm<T,EPI_RESIDUAL> (input gradient W1^T.dh1 + dout)", "mfma", shape
        return "pw_gemm plain (encoder / bottleneck / mask / decoder, fwd or dgrad)", "mfma", shape
    if name in ("ctn_pw_dgrad_gln", "ctn_pw_dgrad_gln_planes"):
        return "B1 pw_gemm<T,EPI_GLN_BWD> (input gradient W2^T.dout + gLN backward sums)", "mfma", (a[3], a[4], a[5])
    if name == "ctn_split_b3_batch":
        return ("weight split into bf16 pieces, one launch per GEMM -- per-kernel probe path only (the timed composite path "
                "issues 4 batched launches per step)"), "hbm", None
    if name == "ctn_pw_wgrad":
        if a[11]:
            fam = "B2 pw_wgrad<PRO> + slab_reduce (dW2 = dout . gLN2(prelu(d))^T)"
        elif a[4] * a[5] >= 256 * 256:
            fam = "B6 pw_wgrad + slab_reduce (dW1 = dh1 . x^T)"
        else:
            fam = "pw_wgrad + slab_reduce, small layers (encoder / decoder bases)"
        return fam, "mfma", (a[3], a[4], a[5])
    return {"ctn_dw_fwd": "K2 dw_fwd (gLN1+PReLU prologue, depthwise, statistics)",
            "ctn_dw_bwd": "B3 dw_bwd fused (gLN2'.PReLU2'.dw^T)",
            "ctn_gln_prelu_bwd": "B4 gln_prelu_bwd",
            "ctn_dw_bwd_finalize": "dw_bwd_finalize (fixed-order parameter-gradient sums)"}.get(name, name), "hbm", None


def _gemm_bytes(name, a, K):
    """Algorithmic HBM bytes of one GEMM launch: every operand / result tensor once (weights are negligible)."""
    M, R, Cn = a[3], a[4], a[5]
    t = 4.0 * M * K
    if name == "ctn_pw_gemm":
        return t * (Cn + R + (R if a[15] else 0))                 # X, Out, residual
    if name in ("ctn_pw_dgrad_gln", "ctn_pw_dgrad_gln_planes"):
        return t * (Cn + 2 * R)                                   # dOut, dN, y
    return t * (R + Cn)                                           # weight gradient: both activations


def family_table(probe, stack, cfg, K, steps):
    """Per kernel family: launches, in-step time, and the roofline that binds it.  A GEMM family is priced against BOTH
    roofs -- executed MFMA FLOPs (3 f16 MFMAs per algorithmic product step under h3, 6 bf16 under b6, 1 fp32 MFMA under fp32) over the
    dense peak of that MFMA type, and algorithmic bytes over 8 TB/s -- and reports the binding (larger) bound."""
    c = cfg["model"]
    M, H = PER_GPU_BATCH, c["H"]
    hbm_bytes = {"ctn_dw_fwd": 2, "ctn_dw_bwd": 4, "ctn_gln_prelu_bwd": 3}      # tensors of M*H*K*4 bytes read + written
    b3 = ARITH["name"] in MFMA_PER_STEP          # split arithmetics: the 1x1 GEMMs run on the bf16 / f16 matrix cores
    nprod = MFMA_PER_STEP.get(ARITH["name"], 1)  # of the composite stacks' GEMMs
    nprod_other = 6 if ARITH["name"] == "h3" else nprod      # GEMMs outside the stacks run as b6 under h3
    fams = {}
    B = c["B"]
    gln = cfg["norm_type"] == "gLN"
    # launch groups of the composite stacks (ctn_probe_read): family id -> (name, GEMM shape (M, R, Cn) or None, tensors of
    # M*ch*K*4 bytes moved as (channels, count) pairs)
    t4 = 4.0 * M * K
    import conv_tasnet_amd as ctn
    cf = (not gln) and ctn.lib.load().ctn_cln_fuse() != 0       # cLN stacks: the second norm's backward rides in B1's epilogue + dw_bwd
    cf1 = (not gln) and ctn.lib.load().ctn_cln_fuse() >= 2      # and the first norm's forward in K1's epilogue + K2's prologue
    STACK = {
        0: ("K1 1x1 B->H (+ PReLU/gLN statistics)" if gln else ("K1 1x1 B->H (+ per-frame PReLU/cLN statistics)" if cf1 else "K1 1x1 B->H"), (M, H, B), t4 * (B + H)),
        1: ("K2 dw_fwd (gLN1+PReLU prologue, depthwise, statistics)" if gln else ("K2 dw_fwd (cLN1+PReLU prologue, depthwise)" if cf1 else "K2 dw_fwd (depthwise)"),
            None, t4 * 2 * H),
        2: ("K3 1x1 H->B (gLN prologue + residual)" if gln else "K3 1x1 H->B + residual", (M, B, H), t4 * (H + 2 * B)),
        3: ("B1 input gradient W2^T.dout (+ gLN backward sums)" if gln else ("B1 input gradient W2^T.dout (+ per-frame cLN backward sums)" if cf else "B1 input gradient W2^T.dout"),
            (M, H, B), t4 * (B + (2 if (gln or cf) else 1) * H)),
        4: ("B2 weight gradient dW2 (gLN prologue) + slab_reduce" if gln else "B2 weight gradient dW2 + slab_reduce", (M, B, H), t4 * (B + H)),
        5: ("B3 dw_bwd fused (gLN2'.PReLU2'.dw^T)" if gln else ("B3 dw_bwd fused (cLN2'.PReLU2'.dw^T)" if cf else "B3 dw_bwd (depthwise^T)"), None,
            t4 * (4 if (gln or cf) else 3) * H),
        6: ("B4 gln_prelu_bwd", None, t4 * 3 * H),
        7: ("B5 input gradient W1^T.dh1 + dout", (M, B, H), t4 * (H + 2 * B)),
        8: ("B6 weight gradient dW1 + slab_reduce", (M, H, B), t4 * (B + H)),
        9: ("fixed-order parameter-gradient sums (finalize)", None, 0.0),
        10: ("weight operands of the stack (bf16 pieces / transposes), 2 launches per direction", None, 0.0),
        11: ("cln_fwd (channel-wise LayerNorm of PReLU(.))", None, t4 * 2 * H),
        12: ("cln_bwd (input gradient + parameter-gradient partials)", None, t4 * 3 * H),
        13: ("dw_bwd_taps (depthwise weight gradient sums)", None, 0.0),
        14: ("slab_reduce that ends a chain of weight gradients (ctn_tune wgrad_chain = 1 only)", None, 0.0),
        15: ("cln_stats_frame / cln_bwd_frame (per-frame statistics / backward constants of the fused cLN)", None, 0.0),
    }
    # the forward families run as `chains` half-batch launches per block (two streams): each launch does 1 / chains of the work
    nblk = c["X"] * c["R"]
    per_fid = {}
    for fid, _ in stack:
        per_fid[fid] = per_fid.get(fid, 0) + 1
    for fid, us in stack:
        name, shape, nbytes = STACK[fid]
        per_block = per_fid[fid] / float(steps * nblk)
        chains = float(round(per_fid.get(0, 0) / float(steps * nblk))) if fid in (0, 1, 2, 11) else 1.0      # forward: K1's launches per block = chains
        chains = max(chains, 1.0)
        if chains > 1.0 and "half-batch" not in name:
            name += " [two half-batch launches per block]"
        f = fams.setdefault(name, {"bound": "mfma" if shape else "hbm", "us": [], "flops": 0.0, "bytes": 0.0, "b3": 0, "entry": "stack", "nprod": nprod})
        f["us"].append(us)
        f["bytes"] += nbytes / chains
        if shape:
            f["flops"] += 2.0 * shape[1] * shape[2] * K * shape[0] / chains
            small = (shape[1] < 32 or shape[2] < 32) if fid in (4, 8) else shape[1] < 64
            f["b3"] += int(b3 and not small)
    for name, a, e0, e1 in probe:
        if name.startswith("ctn_tcn_") or name.startswith("ctn_probe"):
            continue                                # the stacks as wholes: their launch groups are in `stack`
        fam, bound, shape = _family(name, a)
        f = fams.setdefault(fam, {"bound": bound, "us": [], "flops": 0.0, "bytes": 0.0, "b3": 0, "entry": name, "nprod": nprod_other})
        f["us"].append(1e3 * e0.elapsed_time(e1))
        if bound == "mfma":
            f["flops"] += 2.0 * shape[1] * shape[2] * K * shape[0]           # algorithmic FLOPs: 2*R*Cn per frame
            f["bytes"] += _gemm_bytes(name, a, K)
            small = (shape[1] < 32 or shape[2] < 32) if name == "ctn_pw_wgrad" else shape[1] < 64
            f["b3"] += int(b3 and not small)                                 # the library's rule (ctn_gemm.hip: b3_fwd / b3_wgrad)
        elif name in hbm_bytes:
            f["bytes"] += hbm_bytes[name] * 4.0 * M * H * K                  # algorithmic bytes
    rows = []
    for fam, f in fams.items():
        tot = sum(f["us"])
        n = len(f["us"])
        row = {"family": fam, "bound": f["bound"], "launches_per_step": round(n / steps, 1),
               "us_per_launch": round(tot / n, 2), "ms_per_step": round(tot / steps / 1e3, 3),
               "algorithmic_bytes_per_step": f["bytes"] / steps}
        if f["flops"] > 0:
            row["executed_mfma_flops_per_step"] = f["flops"] / steps * (f["nprod"] if f["b3"] * 2 > n else 1)
            row["on_bf16_mfma"] = f["b3"] * 2 > n
        if f["flops"] > 0:
            on_b3 = f["b3"] * 2 > n
            peak = PEAK_BF16_MFMA_TFLOPS if on_b3 else PEAK_F32_MFMA_TFLOPS
            mfma_flops = f["flops"] * (f["nprod"] if on_b3 else 1)           # executed MFMA FLOPs
            t_mfma, t_hbm = mfma_flops / (peak * 1e12), f["bytes"] / (PEAK_HBM_GBS * 1e9)
            h3_row = ARITH["name"] == "h3" and f["entry"] == "stack"
            row["arith"] = ("%s (%d x v_mfma_f32_32x32x16_%s per 16-deep step)" % ("h3" if h3_row else ("b6" if ARITH["name"] == "h3" else ARITH["name"]), f["nprod"],
                                                                                   "f16" if h3_row else "bf16")) if on_b3 else "fp32 (v_mfma_f32_32x32x2_f32)"
            row["algorithmic_tflops"] = round(f["flops"] / (tot * 1e-6) / 1e12, 2)
            row["mfma_floor_us"] = round(t_mfma / n * 1e6, 2)
            row["hbm_floor_us"] = round(t_hbm / n * 1e6, 2)
            if t_hbm >= t_mfma:
                rate = f["bytes"] / (tot * 1e-6)
                row.update(bound="hbm", achieved=round(rate / 1e9, 1), unit="GB/s", peak=PEAK_HBM_GBS, frac=round(rate / 1e9 / PEAK_HBM_GBS, 4),
                           work_per_launch=f["bytes"] / n)
            else:
                rate = mfma_flops / (tot * 1e-6)
                row.update(bound="mfma", achieved=round(rate / 1e12, 2), unit="TFLOP/s", peak=peak, frac=round(rate / 1e12 / peak, 4),
                           work_per_launch=mfma_flops / n)
        elif f["bytes"] > 0:
            rate = f["bytes"] / (tot * 1e-6)
            row.update(achieved=round(rate / 1e9, 1), unit="GB/s", peak=PEAK_HBM_GBS, frac=round(rate / 1e9 / PEAK_HBM_GBS, 4),
                       work_per_launch=f["bytes"] / n)
        rows.append(row)
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows


def roofline(rows, probe_steps, ms_per_step, fused_min_bytes):
    dom = rows[0]
    # concurrency-neutral roll-up of the whole step: every family's algorithmic bytes (each operand / result tensor of each
    # launch once -- this build's blocking) and executed MFMA FLOPs, over the TIMED step (not the sum of in-step durations,
    # which double-counts the overlap of the two streams)
    step_bytes = sum(r["algorithmic_bytes_per_step"] for r in rows)
    bf16_flops = sum(r.get("executed_mfma_flops_per_step", 0.0) for r in rows if r.get("on_bf16_mfma"))
    f32_flops = sum(r.get("executed_mfma_flops_per_step", 0.0) for r in rows if r.get("on_bf16_mfma") is False)
    t = ms_per_step * 1e-3
    step = {"ms_per_step": ms_per_step,
            "algorithmic_bytes": step_bytes, "hbm_frac": round(step_bytes / t / (PEAK_HBM_GBS * 1e9), 4),
            "fused_minimum_bytes": fused_min_bytes, "hbm_frac_of_fused_minimum": round(fused_min_bytes / t / (PEAK_HBM_GBS * 1e9), 4),
            "executed_bf16_mfma_flops": bf16_flops, "bf16_mfma_frac": round(bf16_flops / t / (PEAK_BF16_MFMA_TFLOPS * 1e12), 4),
            "executed_f32_mfma_flops": f32_flops, "f32_mfma_frac": round(f32_flops / t / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
            "note": "whole-step fractions: bytes (this build's blocking / SURVEY 8d's fused minimum 4K(3B+4H)+4K(3B+9H) per block "
                    "and utterance) and executed MFMA FLOPs of one step over the timed ms_per_step and the peaks"}
    traffic, src = None, None
    for pmc in sorted(os.listdir(os.path.join(ROOT, "profiles"))):      # PMC passes (benchmarks/pmc_traffic.sh): family + arithmetic must match
        if not (len(pmc) > 8 and pmc[0] == "r" and pmc[1:3].isdigit() and pmc[3:8] == "_pmc_" and pmc.endswith(".json")):
            continue                                    # r<round>_pmc_*.json, sorted: the latest round's pass wins
        try:
            j = json.load(open(os.path.join(ROOT, "profiles", pmc)))
            if j.get("family", "")[:2] == dom["family"][:2] and j.get("arith", "fp32") in (ARITH["name"], "any"):
                traffic, src = j.get("hbm_bytes_per_launch"), "profiles/" + pmc
        except Exception:
            pass
    return {"step": step, "bound": dom["bound"], "kernel": dom["family"], "achieved": dom.get("achieved"),
            "peak": dom.get("peak"), "unit": dom.get("unit"),
            "frac": dom.get("frac"), "us_per_launch": dom["us_per_launch"], "launches_per_step": dom["launches_per_step"],
            "work_per_launch": dom.get("work_per_launch"), "traffic": traffic, "traffic_source": src,
            "how": "largest total time among the kernel families of %d extra training steps after the timed region, run "
                   "exactly like the timed steps (composite stacks, second stream); every launch group is bracketed by a "
                   "HIP-event pair on the stream it is launched to -- inside the library for the stacks (ctn_probe_enable), "
                   "in the Python wrapper for the other calls (weight-gradient groups run on the second stream and overlap "
                   "the chain, so in-step durations include that sharing)" % probe_steps,
            "families": rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="paper")
    ap.add_argument("--arith", choices=["h3", "b6", "fp32"], default=os.environ.get("CTN_GEMM_ARITH", "h3"),
                    help="GEMM arithmetic: h3 = the composite stacks on two fp16 pieces per fp32 operand under tracked power-of-two scales, "
                         "three f16 MFMAs per product step (fp32-faithful products; library default), b6 = three bf16 pieces per fp32 "
                         "operand, six bf16 MFMAs (fp32-faithful), fp32 = fp32-MFMA kernels")
    ap.add_argument("--no-side-arith", action="store_true", help="skip the short runs on the other arithmetics after the timed region")
    ap.add_argument("--no-side-configs", action="store_true", help="skip the short side records (causal, c3, inference, streaming) after the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("CTN_BENCH_GRAPH", "0")),
                    help="1: replay zero_grad+fwd+loss+bwd from one captured HIP graph (conv_tasnet_amd.graphed)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    global PER_GPU_BATCH
    PER_GPU_BATCH = cfg.get("batch", PER_GPU_BATCH)

    import torch.distributed as dist
    import conv_tasnet_amd as ctn
    from conv_tasnet_amd import ops, parallel
    from conv_tasnet_amd.optim import FlatAdam
    from conv_tasnet_amd.train import SyntheticLoader   # synthetic workload of SURVEY 8d (product code, not the oracle)

    world, rank, device = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, world))
    if device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")

    ARITH["name"] = args.arith
    ctn.set_gemm_arith(args.arith)
    torch.manual_seed(0)
    model = ctn.ConvTasNet(**cfg["model"], norm_type=cfg["norm_type"], causal=cfg["causal"], mask_nonlinear="relu").to(device)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    parallel.broadcast_parameters(opt.flat_params)
    parallel.enable_overlap(opt, cfg["model"]["X"])      # N > 1: one all-reduce bucket per repeat, issued during the backward pass
    # this rank's shard: utterances [rank*8, rank*8+8) of the deterministic harmonic-mixture workload
    mix, lens, src = next(iter(SyntheticLoader(1, PER_GPU_BATCH, samples=cfg["T"], C=cfg["model"]["C"], sample_rate=cfg["sr"],
                                               rank=rank, world=world)))
    mix, lens, src = mix.to(device), lens.to(device), src.to(device)
    loss_acc = torch.zeros((), device=device)

    graphed = None
    if args.graph:
        from conv_tasnet_amd.graphed import GraphedBackprop
        graphed = GraphedBackprop(model, opt, (mix, lens, src))

    AR_PROBE = None

    def step():
        if graphed is not None:
            loss = graphed(mix, lens, src)
        else:
            opt.zero_grad()
            est = model(mix)
            loss, _, _, _ = ctn.cal_loss(src, est, lens)
            loss.backward()
        if AR_PROBE is not None and world > 1:      # roofline leg: the exposed part of the gradient all-reduce (what the step waits for)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            scale = parallel.allreduce_gradients(opt)
            e1.record()
            AR_PROBE.append((e0, e1))
        else:
            scale = parallel.allreduce_gradients(opt)
        opt.step(max_grad_norm=5.0, grad_scale=scale)
        loss_acc.add_(loss.detach())

    for _ in range(args.warmup):
        step()
    loss_acc.zero_()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    issue = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ti = time.perf_counter()
        step()
        issue.append(time.perf_counter() - ti)
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
        ffwd, K = flops_fwd(cfg["model"], cfg["T"])
        ftrain = 3 * ffwd - 4 * K * cfg["model"]["N"] * cfg["model"]["L"]
        utt = PER_GPU_BATCH * world * args.steps
        value = utt / dt
        out = {
            "metric": "4s 8kHz 2-spk utterances/sec (fwd+bwd)", "value": round(value, 2), "unit": "utterances/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_TEXT[args.arith],
            "gemm_arith": args.arith, "data": "synthetic",
            "config": {"workload": (cfg["name"] % PER_GPU_BATCH) + " utterances per GPU, fwd+PIT-loss+bwd+clip(5)+Adam",
                       "global_batch": PER_GPU_BATCH * world, "samples_per_utterance": cfg["T"],
                       "parallelism": "dp%d" % world},
            "hip_graph": bool(args.graph), "mean_loss": round(mean_loss, 4),
            # host time to enqueue one step, measured while the queues have room (the median: late steps of a long run
            # block on queue back-pressure, which is GPU time, not host work)
            "host_issue_ms_per_step": round(1e3 * statistics.median(issue[: max(3, len(issue) // 2)]), 3),
            "model_tflops": round(value * ftrain / 1e12, 2),
        }
        if args.arith == "fp32":       # only a run ON the fp32 MFMA is priced against its peak
            out["model_frac_of_f32_mfma_peak"] = round(value * ftrain / 1e12 / (PEAK_F32_MFMA_TFLOPS * world), 4)
        else:                          # executed bf16 / f16 MFMA FLOPs of the whole step (nprod per algorithmic product) over the dense peak
            out["executed_mfma_frac_of_bf16_peak"] = round(MFMA_PER_STEP[args.arith] * value * ftrain / 1e12 / (PEAK_BF16_MFMA_TFLOPS * world), 4)
    if graphed is None and not args.no_side_arith:
        # the same workload on the other arithmetics, timed the same way right after the main measurement (every rank takes
        # part): `value` above is the run's own arithmetic; these side records show, on the same box in the same process, what
        # the other reference-precision arithmetics (bit-exact fp32 MFMA, b6) cost
        def side_run(name):
            ctn.set_gemm_arith(name)
            nref = max(3, min(args.steps, 10))
            for _ in range(2):
                step()
            loss_acc.zero_()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nref):
                step()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            dref = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dref], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dref = float(t)
            ctn.set_gemm_arith(args.arith)
            return {"value": round(PER_GPU_BATCH * world * nref / dref, 2), "unit": "utterances/sec",
                    "ms_per_step": round(1e3 * dref / nref, 3), "steps": nref, "warmup": 2}
        side = {}
        for name, note in (("fp32", "same step with CTN_GEMM_ARITH=fp32 (v_mfma_f32_32x32x2_f32, bit-exact fp32 products)"),
                           ("b6", "same step with CTN_GEMM_ARITH=b6 (three bf16 pieces per operand, six bf16 MFMAs: fp32-faithful products)"),
                           ("h3", "same step with CTN_GEMM_ARITH=h3 (the library default)")):
            if name != args.arith:
                r = side_run(name)
                r["note"] = note
                side[name + "_arithmetic"] = r
        if rank == 0:
            out.update(side)
    if not args.no_roofline and graphed is None:
        # every rank takes part (the steps contain the gradient all-reduce); rank 0 keeps the table.  The steps run exactly as
        # in the timed region (composite stacks, second stream): the library brackets every launch group of the stacks with a
        # HIP-event pair on its stream (ctn_probe_enable), lib.probe does the same for the calls made from Python (front end,
        # back end, loss, optimiser)
        import ctypes
        probe_steps = 3
        step()
        torch.cuda.synchronize()
        ctn.lib.probe = []
        ctn.lib.call("ctn_probe_enable", 1)
        AR_PROBE = []
        for _ in range(probe_steps):
            step()
        torch.cuda.synchronize()
        ar_us = [1e3 * a.elapsed_time(b) for a, b in AR_PROBE]
        AR_PROBE = None
        cap = 4096 * probe_steps
        fam_ids, fam_us = (ctypes.c_int * cap)(), (ctypes.c_float * cap)()
        n = ctn.lib.load().ctn_probe_read(fam_ids, fam_us, cap)
        probe, ctn.lib.probe = ctn.lib.probe, None
        if rank == 0:
            stack = [(int(fam_ids[i]), float(fam_us[i])) for i in range(min(n, cap))]
            c = cfg["model"]
            fused_min = PER_GPU_BATCH * c["X"] * c["R"] * 4.0 * K * ((3 * c["B"] + 4 * c["H"]) + (3 * c["B"] + 9 * c["H"]))
            out["roofline"] = roofline(family_table(probe, stack, cfg, K, probe_steps), probe_steps, out["ms_per_step"], fused_min)
            if ar_us:
                out["roofline"]["gradient_allreduce"] = {
                    "bytes_per_step": int(opt.flat_grads.numel()) * 4, "exposed_us_per_step": round(sum(ar_us) / len(ar_us), 1),
                    "note": "time the main stream spends in allreduce_gradients() behind the backward pass: the buckets issued "
                            "during it (parallel.GradientBuckets) are waited for here, the front / back-end remainder is reduced here"}
    if rank == 0 and "roofline" in out:
        # executed f16 / bf16 MFMA FLOPs from the per-family table (stack GEMMs at 3 MFMAs per product under h3, the front / back
        # end at b6's 6) instead of 3 x all model FLOPs
        st = out["roofline"]["step"]
        if args.arith != "fp32":
            out["executed_mfma_frac_of_bf16_peak"] = st["bf16_mfma_frac"]          # rank 0's FLOPs over the step and one GPU's peak
            out["executed_mfma_frac_source"] = "roofline.families (per-family executed MFMA FLOPs over the timed step)"

    # ---- side records: the other BASELINE configs and the inference callers, short runs in the same process (every rank takes
    # part: the training records contain the gradient all-reduce) -------------------------------------------------------------
    if not args.no_side_configs and graphed is None and args.config == "paper":
        def sync_time(fn, n):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            d = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([d], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                d = float(t)
            return d

        def side_config(name, warm=5, nsteps=10):
            c2 = CONFIGS[name]
            torch.manual_seed(0)
            m2 = ctn.ConvTasNet(**c2["model"], norm_type=c2["norm_type"], causal=c2["causal"], mask_nonlinear="relu").to(device)
            o2 = FlatAdam(m2.parameters(), lr=1e-3)
            parallel.broadcast_parameters(o2.flat_params)
            parallel.enable_overlap(o2, c2["model"]["X"])
            mx, ln, sr = next(iter(SyntheticLoader(1, 8, samples=c2["T"], C=c2["model"]["C"], sample_rate=c2["sr"], rank=rank, world=world)))
            mx, ln, sr = mx.to(device), ln.to(device), sr.to(device)

            def st2():
                o2.zero_grad()
                ctn.cal_loss(sr, m2(mx), ln)[0].backward()
                o2.step(max_grad_norm=5.0, grad_scale=parallel.allreduce_gradients(o2))
            for _ in range(warm):
                st2()
            d = sync_time(st2, nsteps)
            f2, K2 = flops_fwd(c2["model"], c2["T"])
            rec = {"workload": (c2["name"] % 8) + " utterances per GPU, fwd+PIT-loss+bwd+clip(5)+Adam", "value": round(8 * world * nsteps / d, 2),
                   "unit": "utterances/sec", "ms_per_step": round(1e3 * d / nsteps, 3), "steps": nsteps, "warmup": warm, "gemm_arith": args.arith,
                   "hbm_frac_of_fused_minimum": round(8 * c2["model"]["X"] * c2["model"]["R"] * 4.0 * K2 * ((3 * c2["model"]["B"] + 4 * c2["model"]["H"]) + (3 * c2["model"]["B"] + 9 * c2["model"]["H"]))
                                                      / (d / nsteps) / (PEAK_HBM_GBS * 1e9), 4)}
            return rec, m2, o2

        side_cfg = {}
        rec, m_c, o_c = side_config("causal")
        # streaming causal inference (src/conv_tasnet.py:182,257-266 -- the reason the causal variant exists; conv_tasnet_amd/streaming.py):
        # one utterance, 100-ms chunks, carried depthwise / encoder / overlap-add state; real-time factor = processing time / audio time
        from conv_tasnet_amd.streaming import StreamingSeparator
        chunk = 800
        audio = torch.randn(1, 40 * chunk, device=device) * 0.1
        stream_ms = {}
        for mode in ("graph", "eager"):
            sep = StreamingSeparator(m_c.eval(), batch=1, graph=(mode == "graph"))
            with torch.no_grad():
                for i in range(5):
                    sep.push(audio[:, i * chunk:(i + 1) * chunk])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(40):
                    sep.push(audio[:, i * chunk:(i + 1) * chunk])
                torch.cuda.synchronize()
                stream_ms[mode] = 1e3 * (time.perf_counter() - t0) / 40
        rec["streaming_inference"] = {"chunk_samples": chunk, "chunk_ms": 100.0, "batch": 1, "chunks": 40, "ms_per_chunk": round(stream_ms["graph"], 3),
                                      "real_time_factor": round(stream_ms["graph"] / (1e3 * chunk / 8000.0), 5),
                                      "ms_per_chunk_eager": round(stream_ms["eager"], 3),
                                      "note": "StreamingSeparator(graph=True).push on the causal cLN model: exact chunk-wise inference with carried state, "
                                              "the chunk step (~300 launches) replayed as one HIP graph; ms_per_chunk_eager = the same step issued "
                                              "launch by launch from Python (host-bound)"}
        side_cfg["causal"] = rec
        del sep, m_c, o_c
        rec, m_3, o_3 = side_config("c3")
        side_cfg["c3"] = rec
        del m_3, o_3
        torch.cuda.empty_cache()
        parallel.enable_overlap(opt, cfg["model"]["X"])
        # forward-only throughput of configs[1] (separate.py / evaluate.py: model(mixture) under no_grad -> the stack's forward
        # composite without saved activations), and its dominant kernel family
        model.eval()
        with torch.no_grad():
            def fwd():
                model(mix)
            for _ in range(5):
                fwd()
            ninf = 20
            dinf = sync_time(fwd, ninf)
            inf = {"workload": (cfg["name"] % PER_GPU_BATCH) + " utterances per GPU, forward only (torch.no_grad)", "value": round(PER_GPU_BATCH * world * ninf / dinf, 2),
                   "unit": "utterances/sec", "ms_per_forward": round(1e3 * dinf / ninf, 3), "passes": ninf, "warmup": 5, "gemm_arith": args.arith}
            import ctypes
            fwd()
            torch.cuda.synchronize()
            ctn.lib.probe = []
            ctn.lib.call("ctn_probe_enable", 1)
            for _ in range(3):
                fwd()
            torch.cuda.synchronize()
            cap = 4096 * 3
            fam_ids, fam_us = (ctypes.c_int * cap)(), (ctypes.c_float * cap)()
            n = ctn.lib.load().ctn_probe_read(fam_ids, fam_us, cap)
            probe, ctn.lib.probe = ctn.lib.probe, None
        model.train()
        if rank == 0:
            ffwd, K = flops_fwd(cfg["model"], cfg["T"])
            c = cfg["model"]
            rows = family_table(probe, [(int(fam_ids[i]), float(fam_us[i])) for i in range(min(n, cap))], cfg, K, 3)
            dom = rows[0]
            fwd_min = PER_GPU_BATCH * c["X"] * c["R"] * 4.0 * K * (3 * c["B"] + 4 * c["H"])
            tfw = dinf / ninf
            inf["roofline"] = {"kernel": dom["family"], "bound": dom["bound"], "achieved": dom.get("achieved"), "peak": dom.get("peak"), "unit": dom.get("unit"),
                               "frac": dom.get("frac"), "us_per_launch": dom["us_per_launch"], "launches_per_pass": dom["launches_per_step"],
                               "step": {"fused_minimum_bytes": fwd_min, "hbm_frac_of_fused_minimum": round(fwd_min / tfw / (PEAK_HBM_GBS * 1e9), 4),
                                        "executed_f16_mfma_frac": round(sum(r.get("executed_mfma_flops_per_step", 0.0) for r in rows if r.get("on_bf16_mfma")) / tfw / (PEAK_BF16_MFMA_TFLOPS * 1e12), 4)},
                               "families": [{k: r[k] for k in ("family", "bound", "launches_per_step", "us_per_launch", "ms_per_step", "frac") if k in r} for r in rows[:6]]}
            out["inference"] = inf
            out["other_configs"] = side_cfg

    if world > 1:
        # what the collective library saw: backend, ranks, one device per rank (RCCL == torch's "nccl" backend on ROCm)
        info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "device": torch.cuda.current_device(),
                "device_name": torch.cuda.get_device_name(), "pid": os.getpid()}
        try:
            info["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:      # gloo rehearsal builds
            info["nccl_version"] = "n/a (%s)" % type(e).__name__
        gathered = [None] * world
        dist.all_gather_object(gathered, info)
        if rank == 0:
            out["distributed"] = {"backend": info["backend"], "rccl_ranks": info["world_size"], "nccl_version": info["nccl_version"],
                                  "ranks": [{k: g[k] for k in ("device", "device_name", "pid")} for g in gathered],
                                  "gradient_allreduce_bytes_per_step": int(opt.flat_grads.numel()) * 4,
                                  "buckets": "one per repeat (X blocks) issued during the backward pass + remainder" if getattr(opt, "_ctn_buckets", None) is not None else "one collective after the backward pass"}
    # the process group is finished BEFORE the CPU leg: the other ranks must not sit in a collective (or hold the GPUs) while
    # rank 0 spends minutes on the host cores
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
            out["wall_s_since_process_start"] = round(time.time() - _T_PROC, 1)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
