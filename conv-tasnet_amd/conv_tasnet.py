"""ConvTasNet(N, L, B, H, P, X, R, C) on MI355X -- drop-in for the reference module tree.

Same constructor, attributes, sub-module names and state_dict keys as src/conv_tasnet.py:13-361
(SURVEY Appendix A), so ``serialize`` packages and ``load_state_dict`` interchange with the
reference.  The sub-modules only *hold* parameters; the arithmetic runs in the fused HIP
stages of ``ops.py`` (one autograd node per TemporalBlock), never in torch.nn.functional.
"""
import torch
import torch.nn as nn

from . import ops
from .utils import overlap_and_add  # noqa: F401  (re-exported like the reference module does)

EPS = 1e-8


def _pad_frames(t, K):
    """[.., K] -> [.., Kp] zero padded, the internal activation format."""
    Kp = ops.padded_frames(K)
    if t.shape[-1] == Kp:
        return t.contiguous()
    out = t.new_zeros(t.shape[:-1] + (Kp,))
    out[..., :K] = t
    return out


class ConvTasNet(nn.Module):
    def __init__(self, N, L, B, H, P, X, R, C, norm_type="gLN", causal=False, mask_nonlinear='relu'):
        super().__init__()
        self.N, self.L, self.B, self.H, self.P, self.X, self.R, self.C = N, L, B, H, P, X, R, C
        self.norm_type = norm_type
        self.causal = causal
        self.mask_nonlinear = mask_nonlinear
        self.encoder = Encoder(L, N)
        self.separator = TemporalConvNet(N, B, H, P, X, R, C, norm_type, causal, mask_nonlinear)
        self.decoder = Decoder(N, L)
        # reference init rule (src/conv_tasnet.py:41-43): xavier-normal on every parameter with
        # dim() > 1 -- which includes the [1,Ch,1] gamma/beta of each norm.
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_normal_(p)

    def forward(self, mixture):
        """mixture [M, T] -> est_source [M, C, T]  (src/conv_tasnet.py:45-60)."""
        T = mixture.size(-1)
        w, x, K = self.separator.frontend(mixture, self.encoder.conv1d_U.weight)
        x = self.separator.blocks(x, K)
        return ops.Backend.apply(x, w, self.separator.network[3].weight, self.decoder.basis_signals.weight,
                                 K, T, self.C, self.separator.softmax_mask())

    @classmethod
    def load_model(cls, path):
        package = torch.load(path, map_location=lambda storage, loc: storage)   # CPU, like the reference
        return cls.load_model_from_package(package)

    @classmethod
    def load_model_from_package(cls, package):
        model = cls(package['N'], package['L'], package['B'], package['H'], package['P'], package['X'],
                    package['R'], package['C'], norm_type=package['norm_type'], causal=package['causal'],
                    mask_nonlinear=package['mask_nonlinear'])
        model.load_state_dict(package['state_dict'])
        return model

    @staticmethod
    def serialize(model, optimizer, epoch, tr_loss=None, cv_loss=None):
        package = {k: getattr(model, k) for k in ('N', 'L', 'B', 'H', 'P', 'X', 'R', 'C',
                                                  'norm_type', 'causal', 'mask_nonlinear')}
        package['state_dict'] = model.state_dict()
        package['optim_dict'] = optimizer.state_dict()
        package['epoch'] = epoch
        if tr_loss is not None:
            package['tr_loss'] = tr_loss
            package['cv_loss'] = cv_loss
        return package


class Encoder(nn.Module):
    """mixture [M,T] -> mixture_w [M,N,K] = relu(conv1d(stride L/2))  (src/conv_tasnet.py:97-121)."""

    def __init__(self, L, N):
        super().__init__()
        self.L, self.N = L, N
        self.conv1d_U = nn.Conv1d(1, N, kernel_size=L, stride=L // 2, bias=False)   # parameter holder

    def forward(self, mix_wave):
        U = self.conv1d_U.weight
        w = _EncoderOnly.apply(mix_wave, U)
        K = (mix_wave.size(-1) - self.L) // (self.L // 2) + 1
        return w[..., :K]


class _EncoderOnly(torch.autograd.Function):
    """Stand-alone encoder for API parity (ConvTasNet.forward uses the fused Frontend)."""

    @staticmethod
    def forward(ctx, mix, U):
        M, T = mix.shape
        N, _, L = U.shape
        K = (T - L) // (L // 2) + 1
        Kp = ops.padded_frames(K)
        mix = mix.to(torch.float32).contiguous()
        xcol = torch.empty((M, L, Kp), dtype=torch.float32, device=mix.device)
        ops._chk(mix, U)
        ops.lib.call("ctn_im2col", mix.data_ptr(), xcol.data_ptr(), M, T, L, L, K, Kp, ops._stream())
        w, _ = ops.pw_gemm(U, xcol, N, L, K, relu_out=True)
        ctx.save_for_backward(xcol, w)
        ctx.K = K
        return w

    @staticmethod
    def backward(ctx, dw):
        xcol, w = ctx.saved_tensors
        N, L = w.shape[1], xcol.shape[1]
        g = (dw * (w > 0)).contiguous()
        return None, ops.pw_wgrad(g, xcol, N, L, ctx.K).view(N, 1, L)


class Decoder(nn.Module):
    """(mixture_w [M,N,K], est_mask [M,C,N,K]) -> est_source [M,C,T_conv]  (src/conv_tasnet.py:123-146)."""

    def __init__(self, N, L):
        super().__init__()
        self.N, self.L = N, L
        self.basis_signals = nn.Linear(N, L, bias=False)   # parameter holder

    def forward(self, mixture_w, est_mask):
        M, C, N, K = est_mask.shape
        w = _pad_frames(mixture_w.to(torch.float32), K)
        mask = _pad_frames(est_mask.to(torch.float32), K).view(M, C * N, -1)
        sw = _MaskMul.apply(mask, w, C, 2).view(M * C, N, -1)                     # source_w = mixture_w * est_mask (:140)
        T = (K - 1) * (self.L // 2) + self.L
        return _BasisOla.apply(sw, self.basis_signals.weight, K, T).view(M, C, T)


class _MaskMul(torch.autograd.Function):
    """sw [M,C,N,Kp] = w [M,N,Kp] * act(score [M,C*N,Kp]) through ctn_mask_apply; mode 0 relu, 1 softmax over speakers,
    2 identity.  Columns k >= K of score and w are zeros (so are sw's, except the softmax of a zero score column times
    w = 0 -- also zero)."""

    @staticmethod
    def forward(ctx, score, w, C, mode):
        score, w = score.contiguous(), w.contiguous()
        M, N, Kp = w.shape
        sw = torch.empty((M, C, N, Kp), dtype=torch.float32, device=w.device)
        ops._chk(score, w)
        ops.lib.call("ctn_mask_apply", score.data_ptr(), w.data_ptr(), sw.data_ptr(), M, C, N, Kp, mode, ops._stream())
        ctx.save_for_backward(score, w)
        ctx.cfg = (C, mode)
        return sw

    @staticmethod
    def backward(ctx, dsw):
        score, w = ctx.saved_tensors
        C, mode = ctx.cfg
        M, N, Kp = w.shape
        dsw = dsw.contiguous()
        dscore = torch.empty_like(score)
        dw = torch.empty_like(w)
        ops._chk(dsw)
        ops.lib.call("ctn_mask_apply_bwd", dsw.data_ptr(), score.data_ptr(), w.data_ptr(), dscore.data_ptr(), dw.data_ptr(),
                     M, C, N, Kp, mode, ops._stream())
        return dscore, dw, None, None


class _BasisOla(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sw, V, K, T):
        Bn, N, Kp = sw.shape
        L = V.shape[0]
        fr, _ = ops.pw_gemm(V, sw, L, N, K)
        est = torch.empty((Bn, T), dtype=torch.float32, device=sw.device)
        ops.lib.call("ctn_ola", fr.data_ptr(), est.data_ptr(), Bn, T, L, L, K, Kp, ops._stream())
        ctx.save_for_backward(sw, V)
        ctx.cfg = (K, T)
        return est

    @staticmethod
    def backward(ctx, dest):
        sw, V = ctx.saved_tensors
        K, T = ctx.cfg
        Bn, N, Kp = sw.shape
        L = V.shape[0]
        dest = dest.contiguous()
        dfr = torch.empty((Bn, L, Kp), dtype=torch.float32, device=sw.device)
        ops.lib.call("ctn_unfold", dest.data_ptr(), dfr.data_ptr(), Bn, T, L, L, K, Kp, ops._stream())
        dsw, _ = ops.pw_gemm(V, dfr, N, L, K, trans_w=True)
        return dsw, ops.pw_wgrad(dfr, sw, L, N, K), None, None


class TemporalConvNet(nn.Module):
    """mixture_w [M,N,K] -> est_mask [M,C,N,K]  (src/conv_tasnet.py:149-215)."""

    def __init__(self, N, B, H, P, X, R, C, norm_type="gLN", causal=False, mask_nonlinear='relu'):
        super().__init__()
        self.C = C
        self.mask_nonlinear = mask_nonlinear
        layer_norm = ChannelwiseLayerNorm(N)                      # always channel-wise
        bottleneck_conv1x1 = nn.Conv1d(N, B, 1, bias=False)
        repeats = []
        for _r in range(R):
            blocks = []
            for x in range(X):
                dilation = 2 ** x
                padding = (P - 1) * dilation if causal else (P - 1) * dilation // 2
                blocks.append(TemporalBlock(B, H, P, stride=1, padding=padding, dilation=dilation,
                                            norm_type=norm_type, causal=causal))
            repeats.append(nn.Sequential(*blocks))
        temporal_conv_net = nn.Sequential(*repeats)
        mask_conv1x1 = nn.Conv1d(B, C * N, 1, bias=False)
        self.network = nn.Sequential(layer_norm, bottleneck_conv1x1, temporal_conv_net, mask_conv1x1)

    def softmax_mask(self):
        if self.mask_nonlinear == 'softmax':
            return True
        if self.mask_nonlinear == 'relu':
            return False
        raise ValueError("Unsupported mask non-linear function")   # src/conv_tasnet.py:214

    # -- fused internal stages (padded [M,Ch,Kp] activations) --
    def frontend(self, mixture, U):
        ln, bn = self.network[0], self.network[1]
        w, x = ops.Frontend.apply(mixture, U, ln.gamma, ln.beta, bn.weight)
        L = U.shape[-1]
        K = (mixture.size(-1) - L) // (L // 2) + 1
        return w, x, K

    def blocks(self, x, K):
        blks = [blk for rep in self.network[2] for blk in rep]
        norms = set(b.norm_type for b in blks)
        if ops.composite_enabled() and blks and norms in ({"gLN"}, {"cLN"}):
            # the whole stack behind one C call per direction (ctn_tcn_{gln,cln}_fwd / _bwd); bitwise the per-block path
            params = [p for b in blks for p in b.fused_params()]
            dil = [b.dilation for b in blks]
            gln = norms == {"gLN"}
            if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
                return (ops.TcnGln if gln else ops.TcnCln).apply(x, K, dil, blks[0].causal, *params)
            return (ops.tcn_gln_infer if gln else ops.tcn_cln_infer)(x.contiguous(), K, dil, blks[0].causal, params)
        for blk in blks:
            x = blk.fused(x, K)
        return x

    def forward(self, mixture_w):
        """Reference API: un-padded mixture_w in, mask out."""
        M, N, K = mixture_w.size()
        self.softmax_mask()
        ln, bn = self.network[0], self.network[1]
        w = _pad_frames(mixture_w, K)
        y0 = _ClnOnly.apply(w, ln.gamma, ln.beta, K)
        x = _Pointwise.apply(y0, bn.weight, K)
        x = self.blocks(x, K)
        score = _Pointwise.apply(x, self.network[3].weight, K)                     # [M, C*N, Kp]
        ones = torch.ones((M, N, score.shape[-1]), dtype=torch.float32, device=score.device)
        mask = _MaskMul.apply(score, ones, self.C, 1 if self.mask_nonlinear == 'softmax' else 0)   # relu | softmax over C
        return mask[..., :K]


class _Pointwise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, K):
        x = x.contiguous()
        out, _ = ops.pw_gemm(W, x, W.shape[0], W.shape[1], K)
        ctx.save_for_backward(x, W)
        ctx.K = K
        return out

    @staticmethod
    def backward(ctx, dout):
        x, W = ctx.saved_tensors
        dout = _pad_zero_tail(dout.contiguous(), ctx.K)
        R, Cn = W.shape[0], W.shape[1]
        dx, _ = ops.pw_gemm(W, dout, Cn, R, ctx.K, trans_w=True)
        return dx, ops.pw_wgrad(dout, x, R, Cn, ctx.K).view_as(W), None


def _pad_zero_tail(t, K):
    """Re-establish the zero-padding invariant on a gradient that came from outside our kernels."""
    if t.shape[-1] > K:
        t = t.clone()
        t[..., K:] = 0
    return t


class _ClnOnly(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, K):
        out, mean, rstd = ops.cln_fwd(y, gamma, beta, None, K)
        ctx.save_for_backward(y, mean, rstd, gamma)
        ctx.K = K
        return out

    @staticmethod
    def backward(ctx, dout):
        y, mean, rstd, gamma = ctx.saved_tensors
        dy, dg, db, _ = ops.cln_bwd(dout.contiguous(), y, mean, rstd, gamma, None, ctx.K)
        return dy, dg.view_as(gamma), db.view_as(gamma), None


class TemporalBlock(nn.Module):
    """x + pw2(norm(prelu(dw(norm(prelu(pw1(x)))))))  (src/conv_tasnet.py:218-244)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, norm_type="gLN",
                 causal=False):
        super().__init__()
        conv1x1 = nn.Conv1d(in_channels, out_channels, 1, bias=False)
        prelu = nn.PReLU()
        norm = chose_norm(norm_type, out_channels)
        dsconv = DepthwiseSeparableConv(out_channels, in_channels, kernel_size, stride, padding, dilation,
                                        norm_type, causal)
        self.net = nn.Sequential(conv1x1, prelu, norm, dsconv)
        self.dilation, self.causal, self.norm_type = dilation, bool(causal), norm_type

    def fused_params(self):
        """The 9 parameter tensors of a gLN / cLN block in the C ABI's order (include/ctn_hip.h, ctn_tcn_gln_fwd)."""
        ds = self.net[3]
        norm1, norm2 = self.net[2], ds.norm()
        return (self.net[0].weight, self.net[1].weight, norm1.gamma, norm1.beta, ds.net[0].weight, ds.prelu().weight,
                norm2.gamma, norm2.beta, ds.pointwise().weight)

    def fused(self, x, K):
        ds = self.net[3]
        norm1, norm2 = self.net[2], ds.norm()
        if self.norm_type == "gLN":
            fn = ops.GlnBlock
        elif self.norm_type == "cLN":
            fn = ops.ClnBlock
        else:       # chose_norm's else-branch: BatchNorm1d (src/conv_tasnet.py:305-309)
            return ops.BnBlock.apply(x, self.net[0].weight, self.net[1].weight, norm1.weight, norm1.bias,
                                     ds.net[0].weight, ds.prelu().weight, norm2.weight, norm2.bias,
                                     ds.pointwise().weight, K, self.dilation, self.causal,
                                     norm1.step_state(), norm2.step_state())
        return fn.apply(x, self.net[0].weight, self.net[1].weight, norm1.gamma, norm1.beta,
                        ds.net[0].weight, ds.prelu().weight, norm2.gamma, norm2.beta, ds.pointwise().weight,
                        K, self.dilation, self.causal)

    def forward(self, x):
        K = x.size(-1)
        return self.fused(_pad_frames(x, K), K)[..., :K]


class DepthwiseSeparableConv(nn.Module):
    """Parameter layout of src/conv_tasnet.py:247-278 (Chomp1d shifts the integer names when causal)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, norm_type="gLN",
                 causal=False):
        super().__init__()
        depthwise_conv = nn.Conv1d(in_channels, in_channels, kernel_size, stride=stride, padding=padding,
                                   dilation=dilation, groups=in_channels, bias=False)
        prelu = nn.PReLU()
        norm = chose_norm(norm_type, in_channels)
        pointwise_conv = nn.Conv1d(in_channels, out_channels, 1, bias=False)
        if causal:
            self.net = nn.Sequential(depthwise_conv, Chomp1d(padding), prelu, norm, pointwise_conv)
        else:
            self.net = nn.Sequential(depthwise_conv, prelu, norm, pointwise_conv)
        self._o = 1 if causal else 0

    def prelu(self):
        return self.net[1 + self._o]

    def norm(self):
        return self.net[2 + self._o]

    def pointwise(self):
        return self.net[3 + self._o]


class Chomp1d(nn.Module):
    """Kept for the state-dict numbering; the causal left-pad is folded into the depthwise kernel."""

    def __init__(self, chomp_size):
        super().__init__()
        self.chomp_size = chomp_size

    def forward(self, x):
        return x[:, :, :-self.chomp_size].contiguous()


def chose_norm(norm_type, channel_size):
    if norm_type == "gLN":
        return GlobalLayerNorm(channel_size)
    elif norm_type == "cLN":
        return ChannelwiseLayerNorm(channel_size)
    else:
        return BatchNorm1d(channel_size)


class _BnOnly(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, weight, bias, K, state):
        y = y.contiguous()
        out, mr = ops.bn_fwd(y, None, weight, bias, state[0], state[1], state[2], state[3], state[4], K)
        ctx.save_for_backward(y, weight, mr)
        ctx.cfg = (K, bool(state[2]))
        return out

    @staticmethod
    def backward(ctx, dout):
        y, weight, mr = ctx.saved_tensors
        K, training = ctx.cfg
        dy, dg, db, _ = ops.bn_bwd(_pad_zero_tail(dout.contiguous(), K), y, None, weight, mr, training, K)
        return dy, dg, db, None, None


class BatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d's parameters, buffers and state-dict keys (what chose_norm returns for norm_type="BN",
    src/conv_tasnet.py:305-309) over the HIP kernels: statistics per channel over (M, K) of a [M, Ch, K] input."""

    def __init__(self, channel_size):
        super().__init__(channel_size)      # affine, track_running_stats, eps 1e-5, momentum 0.1: the reference's defaults

    def step_state(self):
        """(running_mean, running_var, use_batch_stats, eps, momentum) for one forward call; counts the batch."""
        training = self.training or self.running_mean is None
        momentum = 0.0 if self.momentum is None else self.momentum
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
            if self.momentum is None:       # cumulative moving average
                momentum = 1.0 / float(self.num_batches_tracked)
        update = self.training and self.track_running_stats
        rm = self.running_mean if (update or not training) else None
        rv = self.running_var if (update or not training) else None
        return (rm, rv, training, self.eps, momentum)

    def forward(self, y):
        if y.dim() != 3:
            raise ValueError("expected [M, channels, K]")
        K = y.size(-1)
        return _BnOnly.apply(_pad_frames(y, K), self.weight, self.bias, K, self.step_state())[..., :K]


class _NormParams(nn.Module):
    def __init__(self, channel_size):
        super().__init__()
        self.gamma = nn.Parameter(torch.Tensor(1, channel_size, 1))
        self.beta = nn.Parameter(torch.Tensor(1, channel_size, 1))
        self.reset_parameters()

    def reset_parameters(self):
        self.gamma.data.fill_(1)
        self.beta.data.zero_()


class ChannelwiseLayerNorm(_NormParams):
    """cLN, src/conv_tasnet.py:313-335."""

    def forward(self, y):
        K = y.size(-1)
        return _ClnOnly.apply(_pad_frames(y, K), self.gamma, self.beta, K)[..., :K]


class _GlnOnly(torch.autograd.Function):
    """Stand-alone gLN with its backward (src/conv_tasnet.py:338-361 is an ordinary autograd module).  Forward: statistics
    and apply as two passes of the depthwise kernel with a unit tap; backward: ctn_gln_bwd_sums (per-row S1, S2 and the
    dgamma / dbeta partials) + ctn_gln_prelu_bwd with a PReLU slope of 1 (= identity) + the fixed-order reductions."""

    @staticmethod
    def forward(ctx, yp, gamma, beta, K):
        M, Ch, Kp = yp.shape
        dev = yp.device
        one_tap = torch.ones((Ch, 1, 1), dtype=torch.float32, device=dev)
        one = torch.ones((1,), dtype=torch.float32, device=dev)           # PReLU slope 1 = identity
        ms = torch.empty((M, 2), dtype=torch.float32, device=dev)
        _, stats = ops.dw_fwd(yp, one_tap, K, 1, False, epi_alpha=one)
        out, _ = ops.dw_fwd(yp, one_tap, K, 1, False, pro=(stats, gamma, beta, one), ms_out=ms)
        ctx.save_for_backward(yp, gamma, ms, one)
        ctx.K = K
        return out

    @staticmethod
    def backward(ctx, dout):
        yp, gamma, ms, one = ctx.saved_tensors
        K = ctx.K
        M, Ch, Kp = yp.shape
        dev = yp.device
        dout = _pad_zero_tail(dout.contiguous(), K)
        sums = torch.empty((M, Ch, 2), dtype=torch.float64, device=dev)
        pc = torch.empty((2, M, Ch), dtype=torch.float32, device=dev)
        dy = torch.empty_like(yp)
        dap = torch.empty((M * Ch,), dtype=torch.float32, device=dev)
        ops._chk(dout, yp, gamma, ms)
        ops.lib.call("ctn_gln_bwd_sums", dout.data_ptr(), yp.data_ptr(), M, Ch, K, Kp, gamma.data_ptr(), one.data_ptr(),
                     ms.data_ptr(), sums.data_ptr(), pc.data_ptr(), ops._stream())
        ops.lib.call("ctn_gln_prelu_bwd", dout.data_ptr(), yp.data_ptr(), dy.data_ptr(), M, Ch, K, Kp, gamma.data_ptr(),
                     one.data_ptr(), ms.data_ptr(), sums.data_ptr(), Ch, dap.data_ptr(), 0, ops._stream())
        dgb = ops.reduce_mid(pc, 2, M, Ch)
        return dy, dgb[0].view_as(gamma), dgb[1].view_as(gamma), None


class GlobalLayerNorm(_NormParams):
    """gLN, src/conv_tasnet.py:338-361.  Inside a TemporalBlock it is fused into the neighbouring kernels; called on its
    own it is an ordinary differentiable module (_GlnOnly)."""

    def forward(self, y):
        K = y.size(-1)
        yp = _pad_frames(y.to(torch.float32), K)
        return _GlnOnly.apply(yp, self.gamma, self.beta, K)[..., :K]
