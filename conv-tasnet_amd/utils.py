"""overlap_and_add / remove_pad with the reference signatures (src/utils.py:9-67)."""
import torch

from . import ops
from ._lib import lib


class _Ola(torch.autograd.Function):
    """frames [Bn, K, L] -> [Bn, (K-1)*step + L]; gather formulation, deterministic."""

    @staticmethod
    def forward(ctx, frames, step):
        Bn, K, L = frames.shape
        if step != L // 2:
            raise NotImplementedError("HIP overlap_and_add implements the model's frame_step = frame_length // 2")
        Kp = ops.padded_frames(K)
        fr = frames.new_zeros((Bn, L, Kp))
        fr[:, :, :K] = frames.transpose(1, 2)
        T = (K - 1) * step + L
        out = torch.empty((Bn, T), dtype=torch.float32, device=frames.device)
        ops._chk(fr)
        lib.call("ctn_ola", fr.data_ptr(), out.data_ptr(), Bn, T, L, L, K, Kp, ops._stream())
        ctx.cfg = (K, L, Kp, T)
        return out

    @staticmethod
    def backward(ctx, dout):
        K, L, Kp, T = ctx.cfg
        dout = dout.contiguous()
        Bn = dout.shape[0]
        dfr = torch.empty((Bn, L, Kp), dtype=torch.float32, device=dout.device)
        lib.call("ctn_unfold", dout.data_ptr(), dfr.data_ptr(), Bn, T, L, L, K, Kp, ops._stream())
        return dfr[:, :, :K].transpose(1, 2), None


def overlap_and_add(signal, frame_step):
    """signal [..., frames, frame_length] -> [..., (frames-1)*frame_step + frame_length]  (src/utils.py:9-47)."""
    outer = signal.size()[:-2]
    frames, frame_length = signal.size()[-2:]
    flat = signal.reshape(-1, frames, frame_length).to(torch.float32)
    return _Ola.apply(flat, frame_step).view(*outer, -1)


def remove_pad(inputs, inputs_lengths):
    """[B,C,T] or [B,T] + lengths [B] -> list of numpy arrays with the padding cut (src/utils.py:50-67)."""
    results = []
    dim = inputs.dim()
    for inp, length in zip(inputs, inputs_lengths):
        n = int(length)
        if dim == 3:
            results.append(inp[:, :n].reshape(inputs.size(1), -1).cpu().numpy())
        elif dim == 2:
            results.append(inp[:n].reshape(-1).cpu().numpy())
    return results
