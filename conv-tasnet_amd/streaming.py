"""Chunked (streaming) inference for the causal / cLN variant.

With ``causal=True, norm_type='cLN'`` every layer of the reference is frame-local (1x1 convs, PReLU, the
per-frame cLN of src/conv_tasnet.py:313-335 -- which is NOT cumulative) except the depthwise conv,
which looks (P-1)*dilation frames into the past (src/conv_tasnet.py:182,253-256,281-295).  So a stream can be
separated chunk by chunk, exactly, by carrying per block the last (P-1)*dilation frames of the depthwise input,
plus the encoder's L-S input samples and the decoder's L-S overlap-add samples.  Total left context is
R*(P-1)*(2^X-1) frames (2040 at the paper config) -- but it is state, not recomputation.

All arithmetic runs through the same HIP entry points as training (ops.py); torch only slices / concatenates.
"""
import torch

from . import ops
from ._lib import lib


def _padded(t, K):
    """[M,Ch,K] (any stride) -> contiguous zero-padded [M,Ch,Kp]."""
    Kp = ops.padded_frames(K)
    out = t.new_zeros(t.shape[:-1] + (Kp,))
    out[..., :K] = t
    return out


class StreamingSeparator:
    """Stateful chunk-wise separation with a causal ConvTasNet.

        s = StreamingSeparator(model, batch=1)
        for chunk in stream:            # chunk: [M, n*S] samples, S = L//2, n >= 1
            out = s.push(chunk)         # [M, C, n*S]  (delayed by L-S samples)
        tail = s.flush()                # [M, C, L-S]
    Concatenating every `out` and `tail` reproduces ``model(full_mixture)`` on the whole signal.

    ``graph=True``: a chunk is ~300 small launches issued from Python (4 ms of host time for 100 ms of audio); every chunk after
    the first has the same shape, so from the third chunk of a given length on the whole step -- kernels, state updates, the
    copies between them -- is replayed as ONE HIP graph (the second chunk runs eagerly and sizes the workspaces).  The state
    lives in fixed buffers updated in place, so eager and replayed chunks can be mixed; same kernels, same order: same bits.
    """

    def __init__(self, model, batch=1, graph=False):
        if not model.causal or model.norm_type != "cLN":
            raise ValueError("streaming needs the causal cLN variant (gLN statistics span the whole utterance)")
        self.m = model
        self.M = batch
        self.L, self.S = model.L, model.L // 2
        self.dev = next(model.parameters()).device
        self.soft = model.separator.softmax_mask()
        self.use_graph = bool(graph) and self.dev.type == "cuda"
        self.reset()

    def reset(self):
        m, dev, M = self.m, self.dev, self.M
        if getattr(self, "hist", None) is None:         # fixed buffers: a captured graph keeps their addresses
            self.in_tail = torch.zeros((M, self.L - self.S), device=dev)
            self.ola_tail = torch.zeros((M, m.C, self.L - self.S), device=dev)
            self.hist = []
            for rep in m.separator.network[2]:
                for blk in rep:
                    halo = (m.P - 1) * blk.dilation
                    self.hist.append(torch.zeros((M, m.H, halo), device=dev))
            self._graphs = {}            # chunk length -> (graph, static chunk, static output)
            self._seen = {}              # chunk length -> eager steady-state chunks so far
        else:
            self.in_tail.zero_()
            self.ola_tail.zero_()
            for h in self.hist:
                h.zero_()
        self.first = True

    @torch.no_grad()
    def push(self, chunk):
        S, L, M = self.S, self.L, self.M
        assert chunk.shape[0] == M and chunk.shape[1] % S == 0 and chunk.shape[1] >= L
        chunk = chunk.to(self.dev, torch.float32)
        if self.first:
            self.first = False
            return self._step(chunk, True).clone()      # the very first frame starts at sample 0
        n = chunk.shape[1]
        if not self.use_graph:
            return self._step(chunk, False).clone()
        if n not in self._graphs:
            if self._seen.get(n, 0) < 1:                 # one eager chunk of this length first: workspaces, allocator
                self._seen[n] = self._seen.get(n, 0) + 1
                return self._step(chunk, False).clone()
            static_in = chunk.clone()
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):                   # records, does not run: the state is untouched until the replay
                static_out = self._step(static_in, False)
            self._graphs[n] = (g, static_in, static_out)
        g, static_in, static_out = self._graphs[n]
        static_in.copy_(chunk)
        g.replay()
        return static_out.clone()

    def _step(self, chunk, first):
        """One chunk through the network; carried state updated in place.  Returns a view of this step's output buffer."""
        m, S, L, M = self.m, self.S, self.L, self.M
        x = chunk if first else torch.cat([self.in_tail, chunk], dim=1)
        self.in_tail.copy_(x[:, x.shape[1] - (L - S):])
        T = x.shape[1]
        K = (T - L) // S + 1
        Kp = ops.padded_frames(K)
        sep = m.separator
        # encoder -> input cLN -> bottleneck
        xcol = torch.empty((M, L, Kp), device=self.dev)
        x = x.contiguous()
        lib.call("ctn_im2col", x.data_ptr(), xcol.data_ptr(), M, T, L, L, K, Kp, ops._stream())
        w, _ = ops.pw_gemm(m.encoder.conv1d_U.weight, xcol, m.N, L, K, relu_out=True)
        y, _, _ = ops.cln_fwd(w, sep.network[0].gamma, sep.network[0].beta, None, K)
        y, _ = ops.pw_gemm(sep.network[1].weight, y, m.B, m.N, K)
        # temporal blocks with carried depthwise history
        i = 0
        for rep in sep.network[2]:
            for blk in rep:
                ds = blk.net[3]
                halo = (m.P - 1) * blk.dilation
                h, _ = ops.pw_gemm(blk.net[0].weight, y, m.H, m.B, K)
                n1, _, _ = ops.cln_fwd(h, blk.net[2].gamma, blk.net[2].beta, blk.net[1].weight, K)
                cat = torch.cat([self.hist[i], n1[..., :K]], dim=2)          # [M, H, halo + K]
                self.hist[i].copy_(cat[..., cat.shape[2] - halo:])
                Kc = halo + K
                z, _ = ops.dw_fwd(_padded(cat, Kc), ds.net[0].weight, Kc, blk.dilation, True)
                z = _padded(z[..., halo:Kc], K)
                n2, _, _ = ops.cln_fwd(z, ds.norm().gamma, ds.norm().beta, ds.prelu().weight, K)
                y, _ = ops.pw_gemm(ds.pointwise().weight, n2, m.B, m.H, K, residual=y)
                i += 1
        # mask -> decoder frames -> overlap-add with carry
        score, _ = ops.pw_gemm(sep.network[3].weight, y, m.C * m.N, m.B, K)
        sw = ops.mask_apply(score, w, m.C, self.soft)
        fr, _ = ops.pw_gemm(m.decoder.basis_signals.weight, sw.view(M * m.C, m.N, Kp), L, m.N, K)
        Tc = (K - 1) * S + L
        est = torch.empty((M, m.C, Tc), device=self.dev)
        lib.call("ctn_ola", fr.data_ptr(), est.data_ptr(), M * m.C, Tc, L, L, K, Kp, ops._stream())
        est[..., : L - S] += self.ola_tail
        self.ola_tail.copy_(est[..., K * S:])
        return est[..., : K * S]

    @torch.no_grad()
    def flush(self):
        """The last L-S output samples (their second overlap-add tap never arrives)."""
        out = self.ola_tail.clone()
        self.ola_tail.zero_()
        return out
