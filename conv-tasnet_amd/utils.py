"""overlap_and_add / remove_pad with the reference signatures (src/utils.py:9-67)."""
import torch

from . import ops
from ._lib import lib


class _Ola(torch.autograd.Function):
    """frames [Bn, F, L] -> [Bn, (F-1)*step + L] for ANY frame_step (src/utils.py:9-47): ctn_overlap_add, a gather in
    ascending frame order (deterministic, unlike the reference's index_add_ on a GPU)."""

    @staticmethod
    def forward(ctx, frames, step):
        Bn, F, L = frames.shape
        frames = frames.contiguous()
        T = (F - 1) * step + L
        out = torch.empty((Bn, T), dtype=torch.float32, device=frames.device)
        ops._chk(frames)
        lib.call("ctn_overlap_add", frames.data_ptr(), out.data_ptr(), Bn, F, L, step, ops._stream())
        ctx.cfg = (F, L, step)
        return out

    @staticmethod
    def backward(ctx, dout):
        F, L, step = ctx.cfg
        dout = dout.contiguous()
        Bn = dout.shape[0]
        dsig = torch.empty((Bn, F, L), dtype=torch.float32, device=dout.device)
        ops._chk(dout)
        lib.call("ctn_overlap_add_bwd", dout.data_ptr(), dsig.data_ptr(), Bn, F, L, step, ops._stream())
        return dsig, None


def overlap_and_add(signal, frame_step):
    """signal [..., frames, frame_length] -> [..., (frames-1)*frame_step + frame_length]  (src/utils.py:9-47)."""
    frame_step = int(frame_step)
    if frame_step < 1:
        raise ValueError("frame_step must be positive")
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
