"""torch.autograd glue over the C ABI (include/ctn_hip.h).

PyTorch is plumbing here: it owns device memory and streams and records which stage's
backward runs when.  Every FLOP of the hot path is a hand-written gfx950 kernel behind
``lib.call``; there is no eager fallback (a missing library or a CPU tensor raises).

Internal activation format: fp32 ``[M, Ch, Kp]`` with ``Kp = padded_frames(K)`` and exact
zeros in columns ``k >= K`` (see the header).  The stage functions below each cover one
autograd node:

  Frontend : encoder conv + ReLU -> input cLN -> bottleneck 1x1   (src/conv_tasnet.py:108-121,172-174)
  GlnBlock : TemporalBlock with gLN, fully fused                  (src/conv_tasnet.py:218-278)
  ClnBlock : TemporalBlock with cLN (causal variant)              (same, norm_type='cLN'; the first norm's forward and the second
             norm's backward ride in the neighbouring GEMM epilogues / depthwise kernels: ctn_tune("cln_fuse"), include/ctn_hip.h)
  Backend  : mask 1x1 -> relu|softmax -> mask*w -> basis -> OLA   (src/conv_tasnet.py:191,206-215,131-146)
  SiSnrPit : PIT SI-SNR loss                                      (src/pit_criterion.py:12-77)
"""
import ctypes
import itertools
import os

import torch

from ._lib import lib, CtnError

F32 = torch.float32
F64 = torch.float64
BF16 = torch.bfloat16

AMAX_SLOTS = 64        # CTN_AMAX_SLOTS of include/ctn_hip.h
_ARITH_NAMES = ("fp32", None, "b6", "h3")        # ids of ctn_gemm_arith / ctn_tune("arith", id)  (1 was round 2's b3: removed)


def gemm_arith():
    """'h3' (default: the composite stacks multiply two fp16 pieces per fp32 operand under tracked power-of-two scales, three f16
    MFMAs, fp32 accumulation -- fp32-faithful products; every other GEMM as b6), 'b6' (three bf16 pieces per operand, six bf16
    MFMAs) or 'fp32' (fp32-MFMA kernels, bit-exact fp32 FMA chains)."""
    return _ARITH_NAMES[lib.ctn_gemm_arith()]


def set_gemm_arith(name):
    """Select the arithmetic of every 1x1-convolution GEMM (include/ctn_hip.h: ctn_gemm_arith).  Change it between steps
    only: statistics layouts and workspace sizes depend on it (the cached workspaces are dropped here)."""
    if name is None or name not in _ARITH_NAMES:
        raise ValueError("gemm arithmetic must be one of %s" % ([n for n in _ARITH_NAMES if n],))
    lib.call("ctn_tune", b"arith", _ARITH_NAMES.index(name))
    _ws_cache.clear()


class gemm_arithmetic:
    """with ops.gemm_arithmetic('fp32'): ...   -- forward AND backward of the enclosed steps on that arithmetic."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prev = gemm_arith()
        set_gemm_arith(self.name)
        return self

    def __exit__(self, *exc):
        set_gemm_arith(self.prev)
        return False


def _p(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts, dtypes=(F32,)):
    """Every tensor handed to the C ABI: on the GPU, contiguous, fp32 (the kernels reinterpret nothing)."""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise CtnError("the HIP path needs tensors on the GPU (got a CPU tensor); there is no CPU fallback")
        if t.dtype not in dtypes:
            raise CtnError("the HIP path computes in fp32: got a %s tensor (call .float() on the model / inputs)" % t.dtype)
        if not t.is_contiguous():
            raise CtnError("internal error: non-contiguous tensor reached the C ABI")


def _chk_aux(*ts):
    """fp64 statistics partials, bf16 planes, integer index tensors."""
    _chk(*ts, dtypes=(F64, BF16, torch.int64, torch.int32))


def padded_frames(K):
    return lib.ctn_padded_frames(int(K))


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _sink(p):
    """Destination view registered by FlatAdam: parameter gradients are then written straight into the flat
    gradient buffer (one use of the parameter per step: overwrite, not accumulate) and autograd gets None."""
    return getattr(p, "_ctn_grad_sink", None)


def _claim_sinks(p):
    """A backward stage is about to OVERWRITE the gradient sinks of its parameters (direct_grads: no accumulation).  A second
    backward pass into the same sinks before zero_grad() would silently drop the first contribution, where the reference's
    autograd accumulates: refuse it."""
    owner = getattr(p, "_ctn_sink_owner", None)
    if owner is not None:
        if id(p) in owner._written:
            raise CtnError("a second backward pass wrote into FlatAdam's gradient buffer without zero_grad() in between: "
                           "direct gradients overwrite -- use FlatAdam(..., direct_grads=False) to accumulate over passes")
        owner._written.add(id(p))


def _emit(grad, sink):
    """Return `grad` to autograd, or copy it into the sink and return None."""
    if sink is None:
        return grad
    sink.copy_(grad.reshape(sink.shape))
    return None


# ---------------------------------------------------------------------------------------
# thin typed wrappers (one per entry point actually used below)
# ---------------------------------------------------------------------------------------
def pw_gemm(W, X, R, Cn, K, trans_w=False, pro=None, residual=None, epi_alpha=None, relu_out=False, ms_out=None):
    """Out[M,R,Kp] = op(W) . f(X).  pro = (part[M,np,2] f64, gamma, beta, alpha).  Returns (Out, epi_part|None)."""
    M, _, Kp = X.shape
    out = torch.empty((M, R, Kp), dtype=F32, device=X.device)
    epi_part = None
    if epi_alpha is not None:
        epi_part = torch.empty((M, lib.ctn_pw_stats_parts(M, R, Kp), 2), dtype=F64, device=X.device)
    pp, npart, pg, pb, pa = (None, 0, None, None, None) if pro is None else (pro[0], pro[0].shape[1], pro[1], pro[2], pro[3])
    _chk(W, X, pg, pb, pa, residual, epi_alpha, ms_out)
    _chk_aux(pp)
    tw = int(trans_w)
    if _b3_planes_ok(R) and not relu_out:
        W, tw = _b3_pieces(W, R, Cn, bool(trans_w)), 2           # the product kernel of the split-bf16 arithmetics (pre-split weights)
    lib.call("ctn_pw_gemm", _p(W), _p(X), _p(out), M, R, Cn, K, Kp, tw,
             _p(pp), npart, _p(pg), _p(pb), _p(pa), _p(ms_out), _p(residual), _p(epi_alpha), _p(epi_part),
             int(relu_out), _stream())
    return out, epi_part


def _b3_planes_ok(R):
    return R >= 64 and lib.ctn_gemm_arith() != 0


def _h3_block(B, H):
    """The rule of the composite stacks (csrc/ctn_block.hip: use_h3): h3 arithmetic selected and both layer widths >= 64."""
    return lib.ctn_gemm_arith() == 3 and B >= 64 and H >= 64


def _b3_pieces(W, R, Cn, k_major):
    """bf16 piece fragments of one GEMM weight operand [R, Cn] (ctn_split_b3_batch); W stored [Cn, R] when k_major."""
    dst = torch.empty(lib.ctn_split_b3_bytes(R, Cn), dtype=torch.uint8, device=W.device)
    src_t = (ctypes.c_void_p * 1)(W.data_ptr())
    dst_t = (ctypes.c_void_p * 1)(dst.data_ptr())
    lib.call("ctn_split_b3_batch", src_t, dst_t, 1, R, Cn, int(k_major), _stream())
    return dst


# ---- h3 arithmetic (include/ctn_hip.h, "h3" section): the same GEMM forms on two fp16 pieces per operand, with the operands'
# per-utterance maxima as explicit inputs.  The composite stacks drive these entry points from C++; the wrappers are the
# unit-tested surface.
def h3_pieces(W, R, Cn, k_major):
    """fp16 piece fragments (+ the weight's maximum) of one GEMM weight operand [R, Cn]; W stored [Cn, R] when k_major."""
    _chk(W)
    dst = torch.empty(lib.ctn_split_h3_bytes(R, Cn), dtype=torch.uint8, device=W.device)
    lib.call("ctn_split_h3_batch", (ctypes.c_void_p * 1)(W.data_ptr()), (ctypes.c_void_p * 1)(dst.data_ptr()), 1, R, Cn,
             int(k_major), _stream())
    return dst


def absmax_rows(x, out=None):
    """int32 [M, AMAX_SLOTS]: tracked maximum of |x[m]| (bit patterns; the maximum over the slots), merged into `out` when given."""
    _chk(x)
    M = x.shape[0]
    if out is None:
        out = torch.zeros((M, AMAX_SLOTS), dtype=torch.int32, device=x.device)
    lib.call("ctn_absmax_rows", _p(x), M, x[0].numel(), _p(out), _stream())
    return out


def absmax_of(*vectors):
    """float32 [len(vectors)]: max |v| of each (equally long) parameter vector -- pro_gbmax = absmax_of(gamma, beta)."""
    _chk(*vectors)
    out = torch.empty((len(vectors),), dtype=F32, device=vectors[0].device)
    n = len(vectors)
    lib.call("ctn_absmax_batch", (ctypes.c_void_p * n)(*[v.data_ptr() for v in vectors]),
             (ctypes.c_void_p * n)(*[out.data_ptr() + 4 * i for i in range(n)]), n, vectors[0].numel(), _stream())
    return out


def pw_gemm_h3(Wp, X, R, Cn, K, x_amax, pro=None, gbmax=None, residual=None, epi_alpha=None, ms_out=None, out_amax=None):
    """ctn_pw_gemm_h3: Out = W . f(X) on h3 pieces Wp (h3_pieces).  Returns (Out, epi_part|None)."""
    M, _, Kp = X.shape
    out = torch.empty((M, R, Kp), dtype=F32, device=X.device)
    epi_part = None
    if epi_alpha is not None:
        epi_part = torch.empty((M, lib.ctn_pw_stats_parts(M, R, Kp), 2), dtype=F64, device=X.device)
    pp, npart, pg, pb, pa = (None, 0, None, None, None) if pro is None else (pro[0], pro[0].shape[1], pro[1], pro[2], pro[3])
    _chk(X, pg, pb, pa, residual, epi_alpha, ms_out, gbmax)
    _chk_aux(pp, x_amax, out_amax)
    lib.call("ctn_pw_gemm_h3", _p(Wp), _p(X), _p(out), M, R, Cn, K, Kp, _p(pp), npart, _p(pg), _p(pb), _p(pa), _p(ms_out),
             _p(residual), _p(epi_alpha), _p(epi_part), _p(x_amax), _p(gbmax), _p(out_amax), _stream())
    return out, epi_part


def pw_dgrad_gln_h3(Wp, dOut, R, Cn, K, y, gamma, alpha, ms, g_amax):
    M, _, Kp = dOut.shape
    dn = torch.empty((M, R, Kp), dtype=F32, device=dOut.device)
    part = torch.empty((M, lib.ctn_pw_stats_parts(M, R, Kp), 2), dtype=F64, device=dOut.device)
    _chk(dOut, y, gamma, alpha, ms)
    _chk_aux(g_amax)
    lib.call("ctn_pw_dgrad_gln_h3", _p(Wp), _p(dOut), _p(dn), M, R, Cn, K, Kp, _p(y), _p(gamma), _p(alpha), _p(ms), _p(part),
             _p(g_amax), _stream())
    return dn, part


def pw_wgrad_h3(dOut, X, R, Cn, K, g_amax, x_amax, pro=None, gbmax=None, out=None, ws_tag="wgrad_h3"):
    """ctn_pw_wgrad_h3: dW[R,Cn] = sum_{m,k} dOut[m,r,k] * f(X[m,c,k]).  pro = (gamma, beta, alpha, ms[M,2])."""
    M, _, Kp = X.shape
    dW = torch.empty((R, Cn), dtype=F32, device=X.device) if out is None else out
    nbytes = lib.ctn_pw_wgrad_h3_workspace(M, R, Cn, Kp)
    ws = _workspace(nbytes, X.device, ws_tag)
    pg, pb, pa, pms = (None, None, None, None) if pro is None else pro
    _chk(dOut, X, pg, pb, pa, pms, gbmax)
    _chk_aux(g_amax, x_amax)
    lib.call("ctn_pw_wgrad_h3", _p(dOut), _p(X), _p(dW), M, R, Cn, K, Kp, _p(pg), _p(pb), _p(pa), _p(pms), _p(g_amax), _p(x_amax),
             _p(gbmax), _p(ws), nbytes, _stream())
    return dW


def pw_dgrad_gln(W, dOut, R, Cn, K, y, gamma, alpha, ms):
    """dN = W^T . dOut (W stored [Cn, R]) + the gLN-backward sums partials.  Returns (dN, sums_part)."""
    M, _, Kp = dOut.shape
    dn = torch.empty((M, R, Kp), dtype=F32, device=dOut.device)
    part = torch.empty((M, lib.ctn_pw_stats_parts(M, R, Kp), 2), dtype=F64, device=dOut.device)
    _chk(W, dOut, y, gamma, alpha, ms)
    if _b3_planes_ok(R):
        lib.call("ctn_pw_dgrad_gln_planes", _p(_b3_pieces(W, R, Cn, True)), _p(dOut), _p(dn), M, R, Cn, K, Kp, _p(y), _p(gamma),
                 _p(alpha), _p(ms), _p(part), _stream())
    else:
        lib.call("ctn_pw_dgrad_gln", _p(W), _p(dOut), _p(dn), M, R, Cn, K, Kp, _p(y), _p(gamma), _p(alpha), _p(ms),
                 _p(part), _stream())
    return dn, part


def pw_dgrad_cln(W, dOut, R, Cn, K, y, gamma, alpha, mean, rstd, g_amax=None):
    """ctn_pw_dgrad_cln: dN = W^T . dOut (W stored [Cn, R]) + the per-frame column partials of the cLN backward sums, on the weight
    form the composite stack uses (h3 pieces when g_amax is given, b6 pieces under the split arithmetics, else the stored matrix).
    Returns (dN, col_part [M, nparts, Kp, 2] f64)."""
    M, _, Kp = dOut.shape
    if g_amax is not None:
        Wp, form = h3_pieces(W, R, Cn, True), 3
    elif _b3_planes_ok(R):
        Wp, form = _b3_pieces(W, R, Cn, True), 2
    else:
        Wp, form = W, 1
    dn = torch.empty((M, R, Kp), dtype=F32, device=dOut.device)
    part = torch.empty((M, lib.ctn_pw_col_parts(M, R, Kp, form), Kp, 2), dtype=F64, device=dOut.device)
    _chk(dOut, y, gamma, alpha, mean, rstd)
    _chk_aux(g_amax)
    lib.call("ctn_pw_dgrad_cln", _p(Wp), form, _p(dOut), _p(dn), M, R, Cn, K, Kp, _p(y), _p(gamma), _p(alpha), _p(mean), _p(rstd),
             _p(part), _p(g_amax), _stream())
    return dn, part


def pw_dgrad_gln2(W, dOut, R, Cn, K, y, gamma, alpha, ms, gamma1, beta1, D, dilation, causal, g_amax=None):
    """ctn_pw_dgrad_gln2: dN = W^T . dOut (W stored [Cn, R]) + the EIGHT per-utterance sums partials from which both norms' backward
    sums follow (include/ctn_hip.h), on the weight form the composite stack uses.  Returns (dN, sums_part [M, parts, 8] f64)."""
    M, _, Kp = dOut.shape
    if g_amax is not None:
        Wp, form = h3_pieces(W, R, Cn, True), 3
    elif _b3_planes_ok(R):
        Wp, form = _b3_pieces(W, R, Cn, True), 2
    else:
        Wp, form = W, 1
    dn = torch.empty((M, R, Kp), dtype=F32, device=dOut.device)
    part = torch.empty((M, lib.ctn_pw_stats_parts(M, R, Kp), 8), dtype=F64, device=dOut.device)
    _chk(dOut, y, gamma, alpha, ms, gamma1, beta1, D)
    _chk_aux(g_amax)
    lib.call("ctn_pw_dgrad_gln2", _p(Wp), form, _p(dOut), _p(dn), M, R, Cn, K, Kp, _p(y), _p(gamma), _p(alpha), _p(ms), _p(gamma1),
             _p(beta1), _p(D), D.shape[-1], dilation, int(causal), _p(part), _p(g_amax), _stream())
    return dn, part


def pw_gemm_cln(W, X, R, Cn, K, alpha, x_amax=None):
    """ctn_pw_gemm_cln: Out = W . X (W stored [R, Cn]) + the per-frame column partials of (sum p, sum p^2), p = prelu(Out, alpha), on
    the weight form the composite stack uses.  Returns (Out, col_part [M, nparts, Kp, 2] f64)."""
    M, _, Kp = X.shape
    if x_amax is not None:
        Wp, form = h3_pieces(W, R, Cn, False), 3
    elif _b3_planes_ok(R):
        Wp, form = _b3_pieces(W, R, Cn, False), 2
    else:
        Wp, form = W.reshape(R, Cn).t().contiguous(), 1          # the composite's [I, O] copy (ctn_transpose_batch)
    out = torch.empty((M, R, Kp), dtype=F32, device=X.device)
    part = torch.empty((M, lib.ctn_pw_col_parts(M, R, Kp, form), Kp, 2), dtype=F64, device=X.device)
    _chk(X, alpha)
    _chk_aux(x_amax)
    lib.call("ctn_pw_gemm_cln", _p(Wp), form, _p(X), _p(out), M, R, Cn, K, Kp, _p(alpha), _p(part), _p(x_amax), _stream())
    return out, part


def cln_stats_frame(col_part, Ch):
    """ctn_cln_stats_frame: (mean, rstd) [M, Kp] of a channel-wise LayerNorm from ctn_pw_gemm_cln's column partials."""
    M, nparts, Kp, _ = col_part.shape
    mean = torch.empty((M, Kp), dtype=F32, device=col_part.device)
    rstd = torch.empty((M, Kp), dtype=F32, device=col_part.device)
    _chk_aux(col_part)
    lib.call("ctn_cln_stats_frame", _p(col_part), nparts, _p(mean), _p(rstd), M, Ch, Kp, _stream())
    return mean, rstd


def dw_fwd_cln(Y, D, K, dilation, causal, mean, rstd, gamma, beta, alpha):
    """ctn_dw_fwd_cln: depthwise conv of cLN(prelu(Y)) with the norm applied in the prologue (per-frame statistics)."""
    M, H, Kp = Y.shape
    Z = torch.empty_like(Y)
    _chk(Y, D, mean, rstd, gamma, beta, alpha)
    lib.call("ctn_dw_fwd_cln", _p(Y), _p(Z), _p(D), M, H, K, Kp, D.shape[-1], dilation, int(causal), _p(mean), _p(rstd), _p(gamma),
             _p(beta), _p(alpha), _stream())
    return Z


def cln_bwd_frame(col_part, mean, rstd, Ch):
    """ctn_cln_bwd_frame: fc [M, 4, Kp] = (rstd, mean rstd, rstd S1/Ch, rstd S2/Ch) per frame from ctn_pw_dgrad_cln's partials."""
    M, nparts, Kp, _ = col_part.shape
    fc = torch.empty((M, 4, Kp), dtype=F32, device=mean.device)
    _chk(mean, rstd)
    _chk_aux(col_part)
    lib.call("ctn_cln_bwd_frame", _p(col_part), nparts, _p(mean), _p(rstd), _p(fc), M, Ch, Kp, _stream())
    return fc


def dw_bwd_cln(dn2, d, n1, D, K, dilation, causal, g2, a2, fc, sinks=None, norm1=None):
    """ctn_dw_bwd_cln + ctn_dw_bwd_cln_finalize: cLN2 <- PReLU2 <- depthwise backward in one pass.
    -> dn1, dD [H,1,P], dgamma2 [H], dbeta2 [H], dalpha2 [1]; sinks = (dD, dgamma2, dbeta2, dalpha2) destinations or None.
    norm1 = (g1, b1, a1, mean1, rstd1): `n1` is then the first norm's INPUT h1 and its output is recomputed in the kernel."""
    M, H, Kp = d.shape
    P = D.shape[-1]
    dev = d.device
    pc = torch.empty((P + 3, M, H), dtype=F32, device=dev)
    dn1 = torch.empty((M, H, Kp), dtype=F32, device=dev)
    n1a = (None,) * 5 if norm1 is None else norm1
    _chk(dn2, d, n1, D, g2, a2, fc, *n1a)
    lib.call("ctn_dw_bwd_cln", _p(dn2), _p(d), _p(n1), _p(dn1), _p(D), M, H, K, Kp, P, dilation, int(causal), _p(g2), _p(a2), _p(fc),
             _p(n1a[0]), _p(n1a[1]), _p(n1a[2]), _p(n1a[3]), _p(n1a[4]), _p(pc), _stream())
    if sinks is None:
        dD = torch.empty((H, 1, P), dtype=F32, device=dev)
        dg, db, da = (torch.empty((H,), dtype=F32, device=dev), torch.empty((H,), dtype=F32, device=dev),
                      torch.empty((1,), dtype=F32, device=dev))
    else:
        dD, dg, db, da = sinks
    lib.call("ctn_dw_bwd_cln_finalize", _p(pc), P, M, H, _p(dD), _p(dg), _p(db), _p(da), _stream())
    return dn1, dD, dg, db, da


_ws_cache = {}

# Weight-gradient GEMMs do not feed the backward chain (their results are only read by the optimiser), so in
# direct-gradient mode they run on a second HIP stream and overlap the latency-bound tails of the chain kernels.
# FlatAdam.step()/the gradient all-reduce join the stream again (join_side_stream()).
_side = {}
_SIDE_ENABLED = os.environ.get("CTN_SIDE_STREAM", "1") != "0"
_CLN_SIDE = os.environ.get("CTN_CLN_SIDE", "1") != "0"     # round 2 (w4 weight-gradient kernel, 5 us slab reduce): 20.07 vs 21.7 ms/step
_SIDE_FIN = os.environ.get("CTN_SIDE_FIN", "0") != "0"   # finishing reductions on the side stream too: measured slower (464 vs 490)


def _side_stream(device):
    st = _side.get(device)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _side[device] = st
    return st


# CTN_LIGHT_EVENTS=1: cross-stream ordering through the library's device-scope event (ctn_stream_order) instead of
# torch's default event.  Measured +0.4 % (500-502 vs 498.5 utt/s): the ~6 us queue bubble per event is not the
# system-scope fence.  Off by default.
_LIGHT_EVENTS = os.environ.get("CTN_LIGHT_EVENTS", "0") != "0"


def _order(src, dst):
    """dst waits for everything enqueued on src so far (device-scope event of the library, or torch's default event)."""
    if _LIGHT_EVENTS:
        with torch.cuda.device(src.device):
            lib.call("ctn_stream_order", src.cuda_stream, dst.cuda_stream)
    else:
        dst.wait_stream(src)


def join_side_stream(device=None):
    """Make the current stream wait for every weight-gradient kernel issued on the side stream."""
    for dev, st in _side.items():
        if device is None or dev == device:
            _order(st, torch.cuda.current_stream(dev))


def _wgrad_async(dOut, X, R, Cn, K, out, pro=None, first=None, first_inputs=(), h3=None):
    """pw_wgrad on the side stream, ordered after everything issued so far on the current stream.
    h3 = (g_amax, x_amax, gbmax): the same through ctn_pw_wgrad_h3.

    first(): other gradient-finishing launches that ride behind the same cross-stream event (each event costs the
    issuing queue a bubble, so they never get one of their own); first_inputs: the tensors they read."""
    dev = X.device
    side = _side_stream(dev)
    _order(torch.cuda.current_stream(dev), side)
    with torch.cuda.stream(side):
        if first is not None:
            first()
        if h3 is not None:
            pw_wgrad_h3(dOut, X, R, Cn, K, h3[0], h3[1], pro=pro, gbmax=h3[2], out=out, ws_tag="wgrad_side")
        else:
            pw_wgrad(dOut, X, R, Cn, K, pro=pro, out=out, ws_tag="wgrad_side")
    for t in (dOut, X) + (tuple(pro) if pro is not None else ()) + tuple(first_inputs) + tuple(t for t in (h3 or ()) if t is not None):
        t.record_stream(side)        # the caching allocator must not hand these out before the side stream is done


_SMALL_SIDE = os.environ.get("CTN_SMALL_WGRAD_SIDE", "1") != "0"


def _wgrad_small(dOut, X, R, Cn, K, sink):
    """Weight gradient of a front-end / back-end 1x1 conv: on the side stream when it lands in a gradient sink (nothing on
    the current stream reads it before join_side_stream()), on the current stream otherwise (autograd consumes the result)."""
    if sink is not None and _SIDE_ENABLED and _SMALL_SIDE and X.is_cuda:
        _wgrad_async(dOut, X, R, Cn, K, sink)
        return None
    return pw_wgrad(dOut, X, R, Cn, K, out=sink)


def _workspace(nbytes, device, tag):
    """Scratch that is fully consumed inside one C call (stream-ordered), so one buffer per tag is enough."""
    key = (device, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def pw_wgrad(dOut, X, R, Cn, K, pro=None, out=None, ws_tag="wgrad"):
    """dW[R,Cn] = sum_{m,k} dOut[m,r,k] * f(X[m,c,k]).  pro = (gamma, beta, alpha, ms[M,2]).  out: optional destination."""
    M, _, Kp = X.shape
    dW = torch.empty((R, Cn), dtype=F32, device=X.device) if out is None else out
    nbytes = lib.ctn_pw_wgrad_workspace(M, R, Cn, Kp)
    ws = _workspace(nbytes, X.device, ws_tag)
    pg, pb, pa, pms = (None, None, None, None) if pro is None else pro
    _chk(dOut, X, pg, pb, pa, pms)
    lib.call("ctn_pw_wgrad", _p(dOut), _p(X), _p(dW), M, R, Cn, K, Kp, _p(pg), _p(pb),
             _p(pa), _p(pms), _p(ws), nbytes, _stream())
    return dW


def reduce_mid(x, F, Mid, Inner, out=None):
    if out is None:
        out = torch.empty((F, Inner), dtype=F32, device=x.device)
    _chk(x, out)
    lib.call("ctn_reduce_mid", _p(x), _p(out), F, Mid, Inner, _stream())
    return out


def cln_fwd(Y, gamma, beta, alpha, K):
    M, Ch, Kp = Y.shape
    out = torch.empty_like(Y)
    mean = torch.empty((M, Kp), dtype=F32, device=Y.device)
    rstd = torch.empty((M, Kp), dtype=F32, device=Y.device)
    _chk(Y, gamma, beta, alpha)
    lib.call("ctn_cln_fwd", _p(Y), _p(out), _p(mean), _p(rstd), M, Ch, K, Kp, _p(gamma), _p(beta), _p(alpha), 0, _stream())
    return out, mean, rstd


def cln_bwd(dOut, Y, mean, rstd, gamma, alpha, K, add=None, relu_ref=None, sinks=None):
    """-> dY, dgamma[Ch], dbeta[Ch], dalpha[1]|None.  sinks = (dgamma, dbeta, dalpha) destinations (FlatAdam's flat
    gradient views): the fixed-order finishing reductions then write there directly and None is returned for them."""
    M, Ch, Kp = Y.shape
    dY = torch.empty_like(Y)
    pc = torch.empty((lib.ctn_cln_bwd_pc_floats(M, Ch, Kp),), dtype=F32, device=Y.device)
    dap = None
    if alpha is not None:
        dap = torch.empty((lib.ctn_cln_bwd_blocks(M, Kp),), dtype=F32, device=Y.device)
    _chk(dOut, Y, mean, rstd, gamma, alpha, add, relu_ref)
    lib.call("ctn_cln_bwd", _p(dOut), _p(Y), _p(dY), _p(mean), _p(rstd), M, Ch, K, Kp, _p(gamma), _p(alpha),
             _p(add), _p(relu_ref), _p(dap), _p(pc), 0, _stream())
    if sinks is not None:
        dg, db, da = sinks
    else:
        dg = torch.empty((Ch,), dtype=F32, device=Y.device)
        db = torch.empty((Ch,), dtype=F32, device=Y.device)
        da = None if dap is None else torch.empty((1,), dtype=F32, device=Y.device)
    _chk(dg, db, da)
    lib.call("ctn_cln_bwd_finalize", _p(pc), _p(dap), M, Ch, Kp, _p(dg), _p(db), _p(da if dap is not None else None),
             _stream())
    if sinks is not None:
        return dY, None, None, None
    return dY, dg, db, da


def dw_fwd(Y, D, K, dilation, causal, pro=None, epi_alpha=None, ms_out=None):
    M, H, Kp = Y.shape
    P = D.shape[-1]
    Z = torch.empty_like(Y)
    epi_part = None if epi_alpha is None else torch.empty((M, H, 2), dtype=F64, device=Y.device)
    pp, npart, pg, pb, pa = (None, 0, None, None, None) if pro is None else (pro[0], pro[0].shape[1], pro[1], pro[2], pro[3])
    _chk(Y, D, pg, pb, pa, epi_alpha, ms_out)
    _chk_aux(pp)
    lib.call("ctn_dw_fwd", _p(Y), _p(Z), _p(D), M, H, K, Kp, P, dilation, int(causal),
             _p(pp), npart, _p(pg), _p(pb), _p(pa), _p(ms_out), _p(epi_alpha), _p(epi_part), 0, _stream())
    return Z, epi_part


# ---------------------------------------------------------------------------------------
# Frontend: mixture -> (mixture_w, bottleneck output)
# ---------------------------------------------------------------------------------------
class Frontend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mix, U, g0, b0, Wb):
        M, T = mix.shape
        N, _, L = U.shape
        B = Wb.shape[0]
        if L % 4 or N % 4 or B % 4:
            raise ValueError("HIP path needs L, N, B to be multiples of 4 (got L=%d N=%d B=%d)" % (L, N, B))
        S = L // 2
        K = (T - L) // S + 1
        Kp = padded_frames(K)
        if K < 1:
            raise ValueError("mixture of %d samples is shorter than one encoder frame (L=%d)" % (T, L))
        mix = _c(mix.to(F32))
        _chk(mix, U, g0, b0, Wb)
        if lib.ctn_encoder_supported(L):        # sliding windows staged in LDS, no im2col buffer (the backward pass unfolds)
            xcol = None
            w = torch.empty((M, N, Kp), dtype=F32, device=mix.device)
            lib.call("ctn_encoder_fwd", _p(mix), _p(_c(U)), _p(w), M, T, N, L, K, Kp, _stream())
        else:
            xcol = torch.empty((M, L, Kp), dtype=F32, device=mix.device)
            lib.call("ctn_im2col", _p(mix), _p(xcol), M, T, L, L, K, Kp, _stream())
            w, _ = pw_gemm(U, xcol, N, L, K, relu_out=True)
        y0, mean0, rstd0 = cln_fwd(w, g0, b0, None, K)
        x0, _ = pw_gemm(Wb, y0, B, N, K)
        ctx.save_for_backward(mix if xcol is None else xcol, w, y0, mean0, rstd0, U, g0, Wb)
        ctx.K, ctx.T, ctx.unfolded = K, T, xcol is not None
        ctx.sinks = (_sink(U), _sink(g0), _sink(b0), _sink(Wb))
        ctx.set_materialize_grads(False)
        return w, x0

    @staticmethod
    def backward(ctx, dw_dec, dx0):
        xcol, w, y0, mean0, rstd0, U, g0, Wb = ctx.saved_tensors
        K = ctx.K
        M, N, Kp = w.shape
        B = Wb.shape[0]
        L = U.shape[-1]
        if not ctx.unfolded:                    # saved the mixture itself: unfold it now for the basis gradient (2 MB)
            mix, xcol = xcol, torch.empty((M, L, Kp), dtype=F32, device=w.device)
            lib.call("ctn_im2col", _p(mix), _p(xcol), M, ctx.T, L, L, K, Kp, _stream())
        if dx0 is None:
            dx0 = torch.zeros((M, B, Kp), dtype=F32, device=w.device)
        dx0 = _c(dx0)
        sU, sg0, sb0, sWb = ctx.sinks
        if sU is not None:
            _claim_sinks(U)
        dy0, _ = pw_gemm(Wb, dx0, N, B, K, trans_w=True)
        dWb = _wgrad_small(dx0, y0, B, N, K, sWb)
        add = None if dw_dec is None else _c(dw_dec)
        g, dg0, db0, _ = cln_bwd(dy0, w, mean0, rstd0, g0, None, K, add=add, relu_ref=w)
        dU = pw_wgrad(g, xcol, N, L, K, out=sU)
        # the first stage of the forward pass is the last node of the backward pass: every weight-gradient kernel of the
        # second stream has been issued by now -- order the current stream after them, so that p.grad / flat_grads may be
        # read right after loss.backward() (not only by FlatAdam.step / the all-reduce, which join as well)
        join_side_stream(w.device)
        return (None, None if sU is not None else dU.view(N, 1, L), _emit(dg0.view(1, N, 1), sg0),
                _emit(db0.view(1, N, 1), sb0), None if sWb is not None else dWb.view(B, N, 1))


# ---------------------------------------------------------------------------------------
# TemporalBlock, gLN: 3 kernels forward, 6 (+3 tiny reductions) backward
# ---------------------------------------------------------------------------------------
class GlnBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, a1, g1, b1, D, a2, g2, b2, w2, K, dilation, causal):
        x = _c(x)
        M, B, Kp = x.shape
        H = w1.shape[0]
        if H % 4 or B % 4:
            raise ValueError("HIP path needs B and H to be multiples of 4")
        dev = x.device
        ms1 = torch.empty((M, 2), dtype=F32, device=dev)
        ms2 = torch.empty((M, 2), dtype=F32, device=dev)
        h3 = _h3_block(B, H)
        if h3:      # the composite stack's arithmetic, kernel by kernel: the tracked maxima are exact, so measuring them here
            #         (absmax_rows) gives the scales -- and the bits -- of the composite, whose producers track them on the fly
            ax = absmax_rows(x)
            h1, st1 = pw_gemm_h3(h3_pieces(w1, H, B, False), x, H, B, K, ax, epi_alpha=a1)
            d, st2 = dw_fwd(h1, D, K, dilation, causal, pro=(st1, g1, b1, a1), epi_alpha=a2, ms_out=ms1)
            ad = absmax_rows(d)
            out, _ = pw_gemm_h3(h3_pieces(w2, B, H, False), d, B, H, K, ad, pro=(st2, g2, b2, a2), gbmax=absmax_of(g2, b2), residual=x, ms_out=ms2)
            ctx.h3 = (ax, ad)
        else:
            h1, st1 = pw_gemm(w1, x, H, B, K, epi_alpha=a1)
            d, st2 = dw_fwd(h1, D, K, dilation, causal, pro=(st1, g1, b1, a1), epi_alpha=a2, ms_out=ms1)
            out, _ = pw_gemm(w2, d, B, H, K, pro=(st2, g2, b2, a2), residual=x, ms_out=ms2)
            ctx.h3 = None
        ctx.save_for_backward(x, h1, d, ms1, ms2, w1, a1, g1, b1, D, a2, g2, b2, w2)
        ctx.cfg = (K, dilation, causal)
        ctx.sinks = tuple(_sink(p) for p in (w1, a1, g1, b1, D, a2, g2, b2, w2))
        return out

    @staticmethod
    def backward(ctx, dout):
        x, h1, d, ms1, ms2, w1, a1, g1, b1, D, a2, g2, b2, w2 = ctx.saved_tensors
        sinks = ctx.sinks
        direct = all(t is not None for t in sinks)
        if direct:
            _claim_sinks(w1)
        K, dilation, causal = ctx.cfg
        dout = _c(dout)
        M, B, Kp = x.shape
        H = w1.shape[0]
        P = D.shape[-1]
        dev = x.device
        st = _stream()
        # -- second 1x1: input gradient (+ gLN2 backward sums) and weight gradient
        _chk(dout, x, h1, d)
        h3 = ctx.h3
        fuse4 = lib.ctn_gln_fuse() != 0         # no gLN-1' / PReLU-1' pass (the composite's rule; include/ctn_hip.h, "gln_fuse")
        ady = None
        if h3 is not None:
            ax, ad = h3
            ady, gbm = absmax_rows(dout), absmax_of(g2, b2)
        if fuse4:
            dn2, s2p = pw_dgrad_gln2(w2, dout, H, B, K, d, g2, a2, ms2, g1, b1, D, dilation, causal, g_amax=ady)
        elif h3 is not None:
            dn2, s2p = pw_dgrad_gln_h3(h3_pieces(w2, H, B, True), dout, H, B, K, d, g2, a2, ms2, ady)
        else:
            dn2, s2p = pw_dgrad_gln(w2, dout, H, B, K, d, g2, a2, ms2)
        np2 = s2p.shape[1]
        side = direct and _SIDE_ENABLED
        if side:
            _wgrad_async(dout, d, B, H, K, sinks[8], pro=(g2, b2, a2, ms2), h3=None if h3 is None else (ady, ad, gbm))
            dW2 = None
        elif h3 is not None:
            dW2 = pw_wgrad_h3(dout, d, B, H, K, ady, ad, pro=(g2, b2, a2, ms2), gbmax=gbm, out=sinks[8] if direct else None)
        else:
            dW2 = pw_wgrad(dout, d, B, H, K, pro=(g2, b2, a2, ms2), out=sinks[8] if direct else None)
        # -- gLN2 <- PReLU2 <- depthwise <- gLN1 output, one pass
        Fr = lib.ctn_dw_bwd_rows(P, 3 if fuse4 else 1)
        pc = torch.empty((Fr, M, H), dtype=F32, device=dev)
        dn1 = torch.empty((M, H, Kp), dtype=F32, device=dev)
        if fuse4:       # the kernel applies the first norm's backward itself: dn1 receives dh1, pc row P + 5 the dalpha1 partials
            lib.call("ctn_dw_bwd_gln2", _p(dn2), _p(d), _p(h1), _p(dn1), _p(D), M, H, K, Kp, P, dilation, int(causal),
                     _p(g1), _p(b1), _p(a1), _p(ms1), _p(g2), _p(a2), _p(ms2), _p(s2p), np2, _p(pc), 0, st)
        else:
            s1p = torch.empty((M, H, 2), dtype=F64, device=dev)
            lib.call("ctn_dw_bwd", _p(dn2), _p(d), _p(h1), _p(dn1), _p(D), M, H, K, Kp, P, dilation, int(causal), 1,
                     _p(g1), _p(b1), _p(a1), _p(ms1), _p(g2), _p(a2), _p(ms2), _p(s2p), np2, _p(pc), _p(s1p), st)
        if direct:
            _, s_a1, s_g1, s_b1, s_D, s_a2, s_g2, s_b2, _ = sinks
            dD, dg2, db2, dg1, db1, da2, da1 = s_D, s_g2, s_b2, s_g1, s_b1, s_a2, s_a1
        else:
            dD = torch.empty((H, 1, P), dtype=F32, device=dev)
            dg2, db2, dg1, db1 = (torch.empty((1, H, 1), dtype=F32, device=dev) for _ in range(4))
            da2 = torch.empty((1,), dtype=F32, device=dev)
            da1 = torch.empty((1,), dtype=F32, device=dev)
        # -- gLN1 + PReLU1 backward, in place on dn1
        if fuse4:
            da1p = pc[P + 5].reshape(-1)
        else:
            da1p = torch.empty((M * H,), dtype=F32, device=dev)
            lib.call("ctn_gln_prelu_bwd", _p(dn1), _p(h1), _p(dn1), M, H, K, Kp, _p(g1), _p(a1), _p(ms1), _p(s1p), H, _p(da1p), 0, st)

        # The fixed-order finishing reductions (one launch: depthwise-weight / gamma / beta / both alpha gradients) feed
        # only parameter gradients.  CTN_SIDE_FIN=1 issues them with the first layer's weight gradient on the second
        # stream; that stream is not the shorter one, so it loses (464 vs 490 utt/s) and they stay on the chain.
        def finish():
            lib.call("ctn_dw_bwd_finalize", _p(pc), P, M, H, _p(dD), _p(dg2), _p(db2), _p(dg1), _p(db1), _p(da2),
                     _p(da1p), M * H, _p(da1), _stream())
        side_fin = side and _SIDE_FIN
        if not side_fin:
            finish()
        # -- first 1x1
        h3w = None if h3 is None else (absmax_rows(dn1), ax, None)
        if side:
            if side_fin:
                _wgrad_async(dn1, x, H, B, K, sinks[0], first=finish, first_inputs=(pc, da1p), h3=h3w)
            else:
                _wgrad_async(dn1, x, H, B, K, sinks[0], h3=h3w)
        if h3 is not None:
            dx, _ = pw_gemm_h3(h3_pieces(w1, B, H, True), dn1, B, H, K, h3w[0], residual=dout)
        else:
            dx, _ = pw_gemm(w1, dn1, B, H, K, trans_w=True, residual=dout)
        if not side:
            if h3 is not None:
                dW1 = pw_wgrad_h3(dn1, x, H, B, K, h3w[0], ax, out=sinks[0] if direct else None)
            else:
                dW1 = pw_wgrad(dn1, x, H, B, K, out=sinks[0] if direct else None)
        if direct:
            return (dx,) + (None,) * 12
        return (dx, dW1.view(H, B, 1), da1, dg1, db1, dD, da2, dg2, db2, dW2.view(B, H, 1), None, None, None)


# ---------------------------------------------------------------------------------------
# The whole stack of gLN TemporalBlocks as ONE autograd node over the composite entry points
# (ctn_tcn_gln_fwd / ctn_tcn_gln_bwd): the host side of 32 blocks is two C calls instead of ~400.
# ---------------------------------------------------------------------------------------
import ctypes  # noqa: E402

_COMPOSITE = os.environ.get("CTN_COMPOSITE", "1") != "0"
_GRAD_BUCKETS = None     # parallel.GradientBuckets: the stack's backward is then issued bucket by bucket (enable_overlap)


def set_grad_buckets(gb):
    global _GRAD_BUCKETS
    _GRAD_BUCKETS = gb


def _bucket_ranges(nb, direct):
    """Block ranges [lo, hi) of the stack's backward calls, last blocks first: one call for the whole stack, or one per
    gradient bucket when an overlapped all-reduce is installed and the gradients go straight into the flat buffer."""
    gb = _GRAD_BUCKETS
    if gb is None or not direct or gb.blocks_per_bucket <= 0 or gb.blocks_per_bucket >= nb:
        return [(0, nb)], None
    if torch.cuda.is_current_stream_capturing():
        # inside a HIP-graph capture (graphed.GraphedBackprop) a collective would either break the capture or be baked into
        # the graph and replayed every step beside finish()'s reduction of the same slice: one range, no bucket; the whole
        # flat gradient is reduced by allreduce_gradients() outside the graph
        return [(0, nb)], None
    per = gb.blocks_per_bucket
    return [(max(hi - per, 0), hi) for hi in range(nb, 0, -per)], gb

NPARAM = 9       # per block: w1, a1, g1, b1, D, a2, g2, b2, w2  (order of include/ctn_hip.h)


_FWD_DUAL = os.environ.get("CTN_FWD_DUAL", "1") != "0"      # forward of the gLN stack as two half-batch chains on two streams


def _fwd_side(dev):
    """Second stream for the forward pass of the composite stack (0 = one chain)."""
    return _side_stream(dev).cuda_stream if (_SIDE_ENABLED and _FWD_DUAL) else 0


def composite_enabled():
    return _COMPOSITE


def _ptr_table(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def tcn_gln_infer(x0, K, dilations, causal, params):
    """Forward of the stack without saving activations (torch.no_grad paths): two x slots, one h1 / d slot."""
    nb = len(dilations)
    M, B, Kp = x0.shape
    H, P = params[0].shape[0], params[4].shape[-1]
    dev = x0.device
    _chk(x0, *params)
    xs = torch.empty((2, M, B, Kp), dtype=F32, device=dev)
    h1 = torch.empty((M, H, Kp), dtype=F32, device=dev)
    d = torch.empty((M, H, Kp), dtype=F32, device=dev)
    ms = torch.empty((2, M, 2), dtype=F32, device=dev)
    amax = torch.empty((nb, 2, M, AMAX_SLOTS), dtype=torch.int32, device=dev)     # h3 arithmetic: tracked operand maxima (zeroed by the call)
    nbytes = lib.ctn_tcn_gln_fwd_workspace(M, B, H, Kp, nb)
    ws = _workspace(nbytes, dev, "tcn_fwd")
    dil = (ctypes.c_int * nb)(*dilations)
    lib.call("ctn_tcn_gln_fwd", _ptr_table(params), dil, nb, _p(x0), _p(xs), _p(h1), _p(d), _p(ms), _p(amax), 0,
             M, B, H, K, Kp, P, int(causal), _p(ws), nbytes, _stream(), _fwd_side(dev))
    return xs[(nb - 1) & 1]


class TcnGln(torch.autograd.Function):
    """x0 [M,B,Kp] -> output of the last TemporalBlock; `params` = 9 tensors per block in NPARAM order."""

    @staticmethod
    def forward(ctx, x0, K, dilations, causal, *params):
        nb = len(dilations)
        if len(params) != nb * NPARAM:
            raise ValueError("TcnGln: expected %d parameter tensors, got %d" % (nb * NPARAM, len(params)))
        x0 = _c(x0)
        M, B, Kp = x0.shape
        H, P = params[0].shape[0], params[4].shape[-1]
        if H % 4 or B % 4:
            raise ValueError("HIP path needs B and H to be multiples of 4")
        dev = x0.device
        _chk(x0, *params)
        xs = torch.empty((nb, M, B, Kp), dtype=F32, device=dev)
        h1s = torch.empty((nb, M, H, Kp), dtype=F32, device=dev)
        ds = torch.empty((nb, M, H, Kp), dtype=F32, device=dev)
        ms = torch.empty((nb, 2, M, 2), dtype=F32, device=dev)
        amax = torch.empty((nb, 2, M, AMAX_SLOTS), dtype=torch.int32, device=dev)     # h3 arithmetic: tracked maxima of every block's input / depthwise output
        nbytes = lib.ctn_tcn_gln_fwd_workspace(M, B, H, Kp, nb)
        ws = _workspace(nbytes, dev, "tcn_fwd")
        dil = (ctypes.c_int * nb)(*dilations)
        lib.call("ctn_tcn_gln_fwd", _ptr_table(params), dil, nb, _p(x0), _p(xs), _p(h1s), _p(ds), _p(ms), _p(amax), 1,
                 M, B, H, K, Kp, P, int(causal), _p(ws), nbytes, _stream(), _fwd_side(dev))
        # our own buffers, written once and read once by backward: plain attributes (released as soon as they are consumed)
        ctx.acts = (x0, xs, h1s, ds, ms, amax)
        ctx.save_for_backward(*params)      # (autograd's version check: an in-place parameter update before backward is an error)
        ctx.cfg = (K, dil, nb, causal, P)
        ctx.sinks = tuple(_sink(p) for p in params)
        return xs[nb - 1]

    @staticmethod
    def backward(ctx, dout):
        if ctx.acts is None:
            raise CtnError("composite TemporalBlock stack: backward called twice on one forward pass (its saved activations are "
                           "released after the first); set CTN_COMPOSITE=0 for retain_graph=True")
        x0, xs, h1s, ds, ms, amax = ctx.acts
        params = ctx.saved_tensors
        K, dil, nb, causal, P = ctx.cfg
        dout = _c(dout)
        _, M, B, Kp = xs.shape
        H = h1s.shape[2]
        dev = x0.device
        _chk(dout)
        direct = all(s is not None for s in ctx.sinks)
        if direct:
            _claim_sinks(params[0])
            gdst, flat = ctx.sinks, None
        else:                       # plain autograd parameters: gradients land in one scratch buffer, returned as views
            sizes = [(p.numel() + 3) // 4 * 4 for p in params]
            flat = torch.empty((sum(sizes),), dtype=F32, device=dev)
            gdst, o = [], 0
            for p, n in zip(params, sizes):
                gdst.append(flat[o:o + p.numel()].view(p.shape))
                o += n
        dxs = torch.empty((nb, M, B, Kp), dtype=F32, device=dev)
        dn1s = torch.empty((nb, M, H, Kp), dtype=F32, device=dev)
        nbytes = lib.ctn_tcn_gln_bwd_workspace(M, B, H, Kp, P, nb)
        ws = _workspace(nbytes, dev, "tcn_bwd")
        side = _side_stream(dev) if (direct and _SIDE_ENABLED) else None
        ranges, gb = _bucket_ranges(nb, direct)
        for i, (lo, hi) in enumerate(ranges):
            # gradient buckets: every call but the last leaves the weight-gradient stream un-joined (its own workspace), and the
            # bucket's all-reduce is issued BEHIND that stream -- all parameter gradients of the stack are produced there --
            # while the main stream already runs the next bucket's chain
            unjoined = gb is not None and side is not None and i + 1 < len(ranges)
            nbi = lib.ctn_tcn_gln_bwd_workspace(M, B, H, Kp, P, hi - lo) if gb is not None else nbytes
            wsi = _workspace(nbi, dev, "tcn_bwd_bucket%d" % i) if unjoined else ws
            lib.call("ctn_tcn_gln_bwd", _ptr_table(params[lo * NPARAM:hi * NPARAM]), _ptr_table(gdst[lo * NPARAM:hi * NPARAM]),
                     (ctypes.c_int * (hi - lo))(*dil[lo:hi]), hi - lo, _p(x0 if lo == 0 else xs[lo - 1]), _p(xs[lo]), _p(h1s[lo]),
                     _p(ds[lo]), _p(ms[lo]), _p(amax[lo]), _p(dout if hi == nb else dxs[hi]), _p(dxs[lo]), _p(dn1s[lo]), M, B, H, K, Kp, P,
                     int(causal), _p(wsi), nbi, _stream(), 0 if side is None else side.cuda_stream, int(unjoined))
            if gb is not None:
                with torch.cuda.stream(side if side is not None else torch.cuda.current_stream(dev)):
                    gb.bucket_ready(gdst[lo * NPARAM:hi * NPARAM])
        ctx.acts = None             # release 4 GB of saved activations as soon as they are consumed
        # the call joined the side stream into the current one, so stream-ordered reuse of these buffers is safe
        if direct:
            return (dxs[0], None, None, None) + (None,) * len(params)
        return (dxs[0], None, None, None) + tuple(gdst)


def tcn_cln_infer(x0, K, dilations, causal, params):
    """cLN stack without saving activations (torch.no_grad paths)."""
    nb = len(dilations)
    M, B, Kp = x0.shape
    H, P = params[0].shape[0], params[4].shape[-1]
    dev = x0.device
    _chk(x0, *params)
    xs = torch.empty((2, M, B, Kp), dtype=F32, device=dev)
    fuse1 = lib.ctn_cln_fuse() >= 2
    h = torch.empty((3 if fuse1 else 4, M, H, Kp), dtype=F32, device=dev)
    h = (h[0], h[0], h[1], h[2]) if fuse1 else (h[0], h[1], h[2], h[3])
    st = torch.empty((4, M, Kp), dtype=F32, device=dev)
    amax = torch.empty((nb, 2, M, AMAX_SLOTS), dtype=torch.int32, device=dev)     # h3 arithmetic: tracked operand maxima (zeroed by the call)
    nbytes = lib.ctn_tcn_cln_fwd_workspace(M, B, H, Kp, nb)
    ws = _workspace(nbytes, dev, "tcn_cln_fwd")
    dil = (ctypes.c_int * nb)(*dilations)
    lib.call("ctn_tcn_cln_fwd", _ptr_table(params), dil, nb, _p(x0), _p(xs), _p(h[0]), _p(h[1]), _p(h[2]), _p(h[3]), _p(st), _p(amax), 0,
             M, B, H, K, Kp, P, int(causal), _p(ws), nbytes, _stream(), _fwd_side(dev))
    return xs[(nb - 1) & 1]


class TcnCln(torch.autograd.Function):
    """The stack of cLN TemporalBlocks (causal BASELINE config) as one autograd node over ctn_tcn_cln_fwd / _bwd: the
    kernels and their order are ClnBlock's, issued from C++ (bitwise the per-kernel path)."""

    @staticmethod
    def forward(ctx, x0, K, dilations, causal, *params):
        nb = len(dilations)
        if len(params) != nb * NPARAM:
            raise ValueError("TcnCln: expected %d parameter tensors, got %d" % (nb * NPARAM, len(params)))
        x0 = _c(x0)
        M, B, Kp = x0.shape
        H, P = params[0].shape[0], params[4].shape[-1]
        if H % 4 or B % 4:
            raise ValueError("HIP path needs B and H to be multiples of 4")
        dev = x0.device
        _chk(x0, *params)
        xs = torch.empty((nb, M, B, Kp), dtype=F32, device=dev)
        fuse1 = lib.ctn_cln_fuse() >= 2            # the first norm's output is neither written nor read (include/ctn_hip.h, "cln_fuse")
        hs = torch.empty((3 if fuse1 else 4, nb, M, H, Kp), dtype=F32, device=dev)          # h1, [n1,] d, n2
        hs = (hs[0], hs[0], hs[1], hs[2]) if fuse1 else (hs[0], hs[1], hs[2], hs[3])          # (n1s must be a valid pointer: never touched when fused)
        st = torch.empty((nb, 4, M, Kp), dtype=F32, device=dev)            # mean1, rstd1, mean2, rstd2
        amax = torch.empty((nb, 2, M, AMAX_SLOTS), dtype=torch.int32, device=dev)     # h3 arithmetic: tracked maxima of every block's input / second norm output
        nbytes = lib.ctn_tcn_cln_fwd_workspace(M, B, H, Kp, nb)
        ws = _workspace(nbytes, dev, "tcn_cln_fwd")
        dil = (ctypes.c_int * nb)(*dilations)
        lib.call("ctn_tcn_cln_fwd", _ptr_table(params), dil, nb, _p(x0), _p(xs), _p(hs[0]), _p(hs[1]), _p(hs[2]), _p(hs[3]), _p(st), _p(amax), 1,
                 M, B, H, K, Kp, P, int(causal), _p(ws), nbytes, _stream(), _fwd_side(dev))
        ctx.acts = (x0, xs, hs, st, amax)
        ctx.save_for_backward(*params)      # (autograd's version check: an in-place parameter update before backward is an error)
        ctx.cfg = (K, dil, nb, causal, P)
        ctx.fuse = lib.ctn_cln_fuse()       # what the forward pass stored (level 2: no n1) -- the backward pass must run under the same value
        ctx.sinks = tuple(_sink(p) for p in params)
        return xs[nb - 1]

    @staticmethod
    def backward(ctx, dout):
        if ctx.acts is None:
            raise CtnError("composite TemporalBlock stack: backward called twice on one forward pass (its saved activations are "
                           "released after the first); set CTN_COMPOSITE=0 for retain_graph=True")
        x0, xs, hs, st, amax = ctx.acts
        if (lib.ctn_cln_fuse() >= 2) != (ctx.fuse >= 2):
            raise CtnError("ctn_tune(\"cln_fuse\") changed between the forward and the backward pass of a cLN stack (%d -> %d): the first "
                           "norm's output is stored only below level 2" % (ctx.fuse, lib.ctn_cln_fuse()))
        params = ctx.saved_tensors
        K, dil, nb, causal, P = ctx.cfg
        dout = _c(dout)
        _, M, B, Kp = xs.shape
        H = hs[0].shape[2]
        dev = x0.device
        _chk(dout)
        direct = all(s is not None for s in ctx.sinks)
        if direct:
            _claim_sinks(params[0])
            gdst = ctx.sinks
        else:
            sizes = [(p.numel() + 3) // 4 * 4 for p in params]
            flat = torch.empty((sum(sizes),), dtype=F32, device=dev)
            gdst, o = [], 0
            for p, n in zip(params, sizes):
                gdst.append(flat[o:o + p.numel()].view(p.shape))
                o += n
        dxs = torch.empty((nb, M, B, Kp), dtype=F32, device=dev)
        dh1s = torch.empty((nb, M, H, Kp), dtype=F32, device=dev)
        nbytes = lib.ctn_tcn_cln_bwd_workspace(M, B, H, Kp, P, nb)
        ws = _workspace(nbytes, dev, "tcn_cln_bwd")
        side = _side_stream(dev) if (direct and _SIDE_ENABLED and _CLN_SIDE) else None
        ranges, gb = _bucket_ranges(nb, direct)
        for i, (lo, hi) in enumerate(ranges):            # (gradient buckets: as in TcnGln.backward)
            unjoined = gb is not None and side is not None and i + 1 < len(ranges)
            nbi = lib.ctn_tcn_cln_bwd_workspace(M, B, H, Kp, P, hi - lo) if gb is not None else nbytes
            wsi = _workspace(nbi, dev, "tcn_cln_bwd_bucket%d" % i) if unjoined else ws
            lib.call("ctn_tcn_cln_bwd", _ptr_table(params[lo * NPARAM:hi * NPARAM]), _ptr_table(gdst[lo * NPARAM:hi * NPARAM]),
                     (ctypes.c_int * (hi - lo))(*dil[lo:hi]), hi - lo, _p(x0 if lo == 0 else xs[lo - 1]), _p(xs[lo]), _p(hs[0][lo]),
                     _p(hs[1][lo]), _p(hs[2][lo]), _p(hs[3][lo]), _p(st[lo]), _p(amax[lo]), _p(dout if hi == nb else dxs[hi]), _p(dxs[lo]), _p(dh1s[lo]),
                     M, B, H, K, Kp, P, int(causal), _p(wsi), nbi, _stream(), 0 if side is None else side.cuda_stream, int(unjoined))
            if gb is not None:
                with torch.cuda.stream(side if side is not None else torch.cuda.current_stream(dev)):
                    gb.bucket_ready(gdst[lo * NPARAM:hi * NPARAM])
        ctx.acts = None
        if direct:
            return (dxs[0], None, None, None) + (None,) * len(params)
        return (dxs[0], None, None, None) + tuple(gdst)


# ---------------------------------------------------------------------------------------
# TemporalBlock, cLN (causal config): unfused norm kernels around the same GEMMs / depthwise
# ---------------------------------------------------------------------------------------
class ClnBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, a1, g1, b1, D, a2, g2, b2, w2, K, dilation, causal):
        x = _c(x)
        M, B, Kp = x.shape
        H = w1.shape[0]
        if H % 4 or B % 4:
            raise ValueError("HIP path needs B and H to be multiples of 4")
        h3 = _h3_block(B, H)       # the composite stack's arithmetic, kernel by kernel (maxima measured here: exact, so the same bits)
        fuse1 = lib.ctn_cln_fuse() >= 2     # the first norm without a pass (the composite's rule): statistics from K1's epilogue, applied in K2's prologue
        ax = absmax_rows(x) if h3 else None
        if fuse1:
            h1, colp = pw_gemm_cln(w1, x, H, B, K, a1, x_amax=ax)
            mean1, rstd1 = cln_stats_frame(colp, H)
            d = dw_fwd_cln(h1, D, K, dilation, causal, mean1, rstd1, g1, b1, a1)
            n1 = h1.new_empty(0)                    # never stored
        else:
            if h3:
                h1, _ = pw_gemm_h3(h3_pieces(w1, H, B, False), x, H, B, K, ax)
            else:
                h1, _ = pw_gemm(w1, x, H, B, K)
            n1, mean1, rstd1 = cln_fwd(h1, g1, b1, a1, K)
            d, _ = dw_fwd(n1, D, K, dilation, causal)
        n2, mean2, rstd2 = cln_fwd(d, g2, b2, a2, K)
        if h3:
            an = absmax_rows(n2)
            out, _ = pw_gemm_h3(h3_pieces(w2, B, H, False), n2, B, H, K, an, residual=x)
        else:
            out, _ = pw_gemm(w2, n2, B, H, K, residual=x)
        ctx.h3 = (ax, an) if h3 else None
        ctx.save_for_backward(x, h1, n1, d, n2, mean1, rstd1, mean2, rstd2, w1, a1, g1, D, a2, g2, w2, b1)
        ctx.cfg = (K, dilation, causal)
        ctx.fuse1 = fuse1
        ctx.sinks = tuple(_sink(p) for p in (w1, a1, g1, b1, D, a2, g2, b2, w2))
        return out

    @staticmethod
    def backward(ctx, dout):
        x, h1, n1, d, n2, mean1, rstd1, mean2, rstd2, w1, a1, g1, D, a2, g2, w2, b1 = ctx.saved_tensors
        K, dilation, causal = ctx.cfg
        dout = _c(dout)
        M, B, Kp = x.shape
        H = w1.shape[0]
        P = D.shape[-1]
        dev = x.device
        sk = ctx.sinks
        direct = all(t is not None for t in sk)       # FlatAdam: gradients go straight into the flat buffer
        if direct:
            _claim_sinks(w1)
        # Weight gradients on the second stream: with the round-1 weight-gradient kernel this lost (22.7 vs 22.3 ms/step: the
        # 1024-thread cLN kernels fill every wave slot); with the w4 kernel and the 5 us slab reduce it wins, 20.07 vs
        # 21.7 ms/step at paper size (CONFIG=causal benchmarks/ab_step.py).  CTN_CLN_SIDE=0 turns it off.
        side = direct and _SIDE_ENABLED and _CLN_SIDE
        h3 = ctx.h3
        fuse = lib.ctn_cln_fuse() != 0 or ctx.fuse1     # the second norm's backward inside the GEMM epilogue + the depthwise backward (the composite's rule)
        ady = None
        if h3 is not None:
            ax, an = h3
            ady = absmax_rows(dout)
        if fuse:
            dn2, colp = pw_dgrad_cln(w2, dout, H, B, K, d, g2, a2, mean2, rstd2, g_amax=ady)
        elif h3 is not None:
            dn2, _ = pw_gemm_h3(h3_pieces(w2, H, B, True), dout, H, B, K, ady)
        else:
            dn2, _ = pw_gemm(w2, dout, H, B, K, trans_w=True)
        if side:
            _wgrad_async(dout, n2, B, H, K, sk[8], h3=None if h3 is None else (ady, an, None))
            dW2 = None
        elif h3 is not None:
            dW2 = pw_wgrad_h3(dout, n2, B, H, K, ady, an, out=sk[8] if direct else None)
        else:
            dW2 = pw_wgrad(dout, n2, B, H, K, out=sk[8] if direct else None)
        if fuse:
            fc = cln_bwd_frame(colp, mean2, rstd2, H)
            dn1, dD, dg2, db2, da2 = dw_bwd_cln(dn2, d, h1 if ctx.fuse1 else n1, D, K, dilation, causal, g2, a2, fc,
                                                sinks=(sk[4], sk[6], sk[7], sk[5]) if direct else None,
                                                norm1=(g1, b1, a1, mean1, rstd1) if ctx.fuse1 else None)
        else:
            dd, dg2, db2, da2 = cln_bwd(dn2, d, mean2, rstd2, g2, a2, K, sinks=(sk[6], sk[7], sk[5]) if direct else None)
            pc = torch.empty((P, M, H), dtype=F32, device=dev)
            dn1 = torch.empty((M, H, Kp), dtype=F32, device=dev)
            _chk(dd, n1)
            lib.call("ctn_dw_bwd", _p(dd), 0, _p(n1), _p(dn1), _p(D), M, H, K, Kp, P, dilation, int(causal), 0,
                     0, 0, 0, 0, 0, 0, 0, 0, 0, _p(pc), 0, _stream())
            dD = reduce_mid(pc, P, M, H).t().contiguous().view(H, 1, P)
        dh1, dg1, db1, da1 = cln_bwd(dn1, h1, mean1, rstd1, g1, a1, K, sinks=(sk[2], sk[3], sk[1]) if direct else None)
        adh = None if h3 is None else absmax_rows(dh1)
        if side:
            _wgrad_async(dh1, x, H, B, K, sk[0], h3=None if h3 is None else (adh, ax, None))
            dW1 = None
        if h3 is not None:
            dx, _ = pw_gemm_h3(h3_pieces(w1, B, H, True), dh1, B, H, K, adh, residual=dout)
        else:
            dx, _ = pw_gemm(w1, dh1, B, H, K, trans_w=True, residual=dout)
        if not side:
            if h3 is not None:
                dW1 = pw_wgrad_h3(dh1, x, H, B, K, adh, ax, out=sk[0] if direct else None)
            else:
                dW1 = pw_wgrad(dh1, x, H, B, K, out=sk[0] if direct else None)
        if direct:
            if not fuse:
                sk[4].copy_(dD)
            return (dx,) + (None,) * 12
        return (dx, dW1.view(H, B, 1), da1, dg1.view(1, H, 1), db1.view(1, H, 1), dD, da2, dg2.view(1, H, 1),
                db2.view(1, H, 1), dW2.view(B, H, 1), None, None, None)


def bn_fwd(Y, alpha, weight, bias, running_mean, running_var, training, eps, momentum, K):
    """(PReLU +) BatchNorm1d over (m, k) per channel -> (out, mr[Ch,2]).  Updates the running statistics in place."""
    M, Ch, Kp = Y.shape
    out = torch.empty_like(Y)
    mr = torch.empty((Ch, 2), dtype=F32, device=Y.device)
    part = torch.empty((Ch * M * 2,), dtype=F64, device=Y.device) if training else None
    _chk(Y, alpha, weight, bias, running_mean, running_var)
    lib.call("ctn_bn_fwd", _p(Y), _p(out), _p(alpha), _p(weight), _p(bias), _p(running_mean), _p(running_var),
             int(training), float(eps), float(momentum), M, Ch, K, Kp, _p(part), _p(mr), _stream())
    return out, mr


def bn_bwd(dOut, Y, alpha, weight, mr, training, K):
    """-> dY, dweight[Ch], dbias[Ch], dalpha[1]|None"""
    M, Ch, Kp = Y.shape
    dev = Y.device
    dY = torch.empty_like(Y)
    part = torch.empty((Ch * M * 2,), dtype=F64, device=dev)
    coef = torch.empty((Ch, 2), dtype=F32, device=dev)
    dg = torch.empty((Ch,), dtype=F32, device=dev)
    db = torch.empty((Ch,), dtype=F32, device=dev)
    dap = None if alpha is None else torch.empty((M * Ch,), dtype=F32, device=dev)
    _chk(dOut, Y, alpha, weight, mr)
    lib.call("ctn_bn_bwd", _p(dOut), _p(Y), _p(dY), _p(alpha), _p(weight), _p(mr), int(training), M, Ch, K, Kp,
             _p(part), _p(coef), _p(dg), _p(db), _p(dap), _stream())
    dalpha = None if dap is None else reduce_mid(dap, 1, M * Ch, 1).view(1)
    return dY, dg, db, dalpha


class BnBlock(torch.autograd.Function):
    """TemporalBlock with norm_type="BN" (src/conv_tasnet.py:218-244,305-309): same chain as ClnBlock with the two
    norms replaced by (PReLU +) BatchNorm1d.  bn1 / bn2 = (running_mean, running_var, training, eps, momentum)."""

    @staticmethod
    def forward(ctx, *args):
        return BnBlock._forward(ctx, *args)

    @staticmethod
    def backward(ctx, dout):
        return BnBlock._backward(ctx, dout)

    @staticmethod
    def _forward(ctx, x, w1, a1, g1, b1, D, a2, g2, b2, w2, K, dilation, causal, bn1, bn2):
        x = _c(x)
        M, B, Kp = x.shape
        H = w1.shape[0]
        if H % 4 or B % 4:
            raise ValueError("HIP path needs B and H to be multiples of 4")
        h1, _ = pw_gemm(w1, x, H, B, K)
        n1, mr1 = bn_fwd(h1, a1, g1, b1, bn1[0], bn1[1], bn1[2], bn1[3], bn1[4], K)
        d, _ = dw_fwd(n1, D, K, dilation, causal)
        n2, mr2 = bn_fwd(d, a2, g2, b2, bn2[0], bn2[1], bn2[2], bn2[3], bn2[4], K)
        out, _ = pw_gemm(w2, n2, B, H, K, residual=x)
        ctx.save_for_backward(x, h1, n1, d, n2, mr1, mr2, w1, a1, g1, D, a2, g2, w2)
        ctx.cfg = (K, dilation, causal, bool(bn1[2]), bool(bn2[2]))
        ctx.sinks = tuple(_sink(p) for p in (w1, a1, g1, b1, D, a2, g2, b2, w2))
        return out

    @staticmethod
    def _backward(ctx, dout):
        x, h1, n1, d, n2, mr1, mr2, w1, a1, g1, D, a2, g2, w2 = ctx.saved_tensors
        K, dilation, causal, tr1, tr2 = ctx.cfg
        dout = _c(dout)
        M, B, Kp = x.shape
        H = w1.shape[0]
        P = D.shape[-1]
        dev = x.device
        dn2, _ = pw_gemm(w2, dout, H, B, K, trans_w=True)
        dW2 = pw_wgrad(dout, n2, B, H, K)
        dd, dg2, db2, da2 = bn_bwd(dn2, d, a2, g2, mr2, tr2, K)
        pc = torch.empty((P, M, H), dtype=F32, device=dev)
        dn1 = torch.empty((M, H, Kp), dtype=F32, device=dev)
        _chk(dd, n1)
        lib.call("ctn_dw_bwd", _p(dd), 0, _p(n1), _p(dn1), _p(D), M, H, K, Kp, P, dilation, int(causal), 0,
                 0, 0, 0, 0, 0, 0, 0, 0, 0, _p(pc), 0, _stream())
        dD = reduce_mid(pc, P, M, H).t().contiguous().view(H, 1, P)
        dh1, dg1, db1, da1 = bn_bwd(dn1, h1, a1, g1, mr1, tr1, K)
        dx, _ = pw_gemm(w1, dh1, B, H, K, trans_w=True, residual=dout)
        dW1 = pw_wgrad(dh1, x, H, B, K)
        sk = ctx.sinks
        return (dx, _emit(dW1.view(H, B, 1), sk[0]), _emit(da1, sk[1]), _emit(dg1, sk[2]), _emit(db1, sk[3]),
                _emit(dD, sk[4]), _emit(da2, sk[5]), _emit(dg2, sk[6]), _emit(db2, sk[7]),
                _emit(dW2.view(B, H, 1), sk[8]), None, None, None, None, None)


# ---------------------------------------------------------------------------------------
# Backend: (TCN output, mixture_w) -> estimated sources [M, C, T]
# ---------------------------------------------------------------------------------------
def _mask_scores(x, Wm, K):
    CN, B = Wm.shape[0], Wm.shape[1]
    score, _ = pw_gemm(Wm, x, CN, B, K)
    return score


def mask_apply(score, w, C, softmax):
    M, N, Kp = w.shape
    sw = torch.empty((M, C, N, Kp), dtype=F32, device=w.device)
    _chk(score, w)
    lib.call("ctn_mask_apply", _p(score), _p(w), _p(sw), M, C, N, Kp, int(softmax), _stream())
    return sw


class Backend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, Wm, V, K, T, C, softmax):
        x, w = _c(x), _c(w)
        M, N, Kp = w.shape
        L = V.shape[0]
        if L % 4 or N % 4:
            raise ValueError("HIP path needs L and N to be multiples of 4")
        score = _mask_scores(x, Wm, K)
        sw = mask_apply(score, w, C, softmax)
        fr, _ = pw_gemm(V, sw.view(M * C, N, Kp), L, N, K)
        est = torch.empty((M, C, T), dtype=F32, device=w.device)
        lib.call("ctn_ola", _p(fr), _p(est), M * C, T, L, L, K, Kp, _stream())
        ctx.save_for_backward(x, w, score, sw, Wm, V)
        ctx.cfg = (K, T, C, softmax)
        ctx.sinks = (_sink(Wm), _sink(V))
        return est

    @staticmethod
    def backward(ctx, dest):
        x, w, score, sw, Wm, V = ctx.saved_tensors
        K, T, C, softmax = ctx.cfg
        dest = _c(dest)
        M, N, Kp = w.shape
        L = V.shape[0]
        B = Wm.shape[1]
        CN = C * N
        dev = w.device
        dfr = torch.empty((M * C, L, Kp), dtype=F32, device=dev)
        _chk(dest)
        lib.call("ctn_unfold", _p(dest), _p(dfr), M * C, T, L, L, K, Kp, _stream())
        dsw, _ = pw_gemm(V, dfr, N, L, K, trans_w=True)                 # [M*C, N, Kp]
        sWm, sV = ctx.sinks
        if sWm is not None:
            _claim_sinks(Wm)
        dV = _wgrad_small(dfr, sw.view(M * C, N, Kp), L, N, K, sV)
        dw = torch.empty((M, N, Kp), dtype=F32, device=dev)
        lib.call("ctn_mask_apply_bwd", _p(dsw), _p(score), _p(w), _p(dsw), _p(dw), M, C, N, Kp, int(softmax), _stream())
        dscore = dsw.view(M, CN, Kp)
        dx, _ = pw_gemm(Wm, dscore, B, CN, K, trans_w=True)
        dWm = _wgrad_small(dscore, x, CN, B, K, sWm)
        return (dx, dw, None if sWm is not None else dWm.view(CN, B, 1), None if sV is not None else dV,
                None, None, None, None)


# ---------------------------------------------------------------------------------------
# PIT SI-SNR loss
# ---------------------------------------------------------------------------------------
_perm_cache = {}


def _perms(C, device):
    key = (C, device)
    if key not in _perm_cache:
        p = list(itertools.permutations(range(C)))   # same order as src/pit_criterion.py:67
        _perm_cache[key] = (torch.tensor(p, dtype=torch.int32, device=device),
                            torch.tensor(p, dtype=torch.int64, device=device))
    return _perm_cache[key]


class SiSnrPit(torch.autograd.Function):
    """(source, estimate, lengths) -> (loss[], max_snr[B,1], estimate masked in place, best perm index [B])."""

    @staticmethod
    def forward(ctx, source, estimate, lengths):
        if source.shape != estimate.shape:
            raise AssertionError("source and estimate_source must have the same size")   # src/pit_criterion.py:34
        Bn, C, T = source.shape
        source = _c(source.to(F32))
        if not estimate.is_contiguous() or estimate.dtype != F32:
            raise CtnError("estimate_source must be a contiguous fp32 tensor (it is masked in place)")
        lengths = _c(lengths.to(device=source.device, dtype=torch.int64))
        dev = source.device
        p32, _ = _perms(C, dev)
        max_snr = torch.empty((Bn, 1), dtype=F32, device=dev)
        idx = torch.empty((Bn,), dtype=torch.int64, device=dev)
        loss = torch.empty((), dtype=F32, device=dev)
        coef = torch.empty((Bn, C, 4), dtype=F32, device=dev)
        jsel = torch.empty((Bn, C), dtype=torch.int32, device=dev)
        nbytes = lib.ctn_sisnr_workspace(Bn, C, T)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        _chk(source, estimate)
        _chk_aux(lengths)
        lib.call("ctn_sisnr_pit_fwd", _p(source), _p(estimate), _p(lengths), _p(p32), p32.shape[0], Bn, C, T,
                 _p(max_snr), _p(idx), _p(loss), 0, _p(coef), _p(jsel), _p(ws), nbytes, _stream())
        ctx.mark_dirty(estimate)
        ctx.mark_non_differentiable(idx)
        ctx.save_for_backward(source, estimate, lengths, coef, jsel)
        ctx.set_materialize_grads(False)
        return loss, max_snr, estimate, idx

    @staticmethod
    def backward(ctx, g_loss, g_max, g_est, _g_idx):
        source, estimate, lengths, coef, jsel = ctx.saved_tensors
        Bn, C, T = source.shape
        d_est = torch.empty_like(source)
        g_loss = None if g_loss is None else _c(g_loss.to(F32))
        g_max = None if g_max is None else _c(g_max.to(F32))
        _chk(g_loss, g_max)
        lib.call("ctn_sisnr_pit_bwd", _p(source), _p(estimate), _p(lengths), _p(coef), _p(jsel), _p(g_loss), _p(g_max),
                 Bn, C, T, _p(d_est), _stream())
        if g_est is not None:
            t = torch.arange(T, device=source.device).view(1, 1, T)
            d_est = d_est + g_est * (t < lengths.view(-1, 1, 1)).to(F32)
        return None, d_est, None
