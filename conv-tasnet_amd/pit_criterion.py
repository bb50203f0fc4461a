"""cal_loss and friends with the reference signatures (src/pit_criterion.py:12-114), on the HIP loss kernels."""
import torch

from . import ops

EPS = 1e-8


def _pit(source, estimate_source, source_lengths):
    assert source.size() == estimate_source.size()           # src/pit_criterion.py:34
    if not (estimate_source.is_contiguous() and estimate_source.dtype == torch.float32):
        raise ops.CtnError("estimate_source must be contiguous fp32 (it is masked in place)")
    return ops.SiSnrPit.apply(source, estimate_source, source_lengths)   # loss, max_snr, est (masked), idx


def cal_loss(source, estimate_source, source_lengths):
    """-> (loss, max_snr [B,1], estimate_source masked IN PLACE like the reference, reorder_estimate_source)."""
    loss, max_snr, est, idx = _pit(source, estimate_source, source_lengths)   # loss = 0 - mean(max_snr), in-kernel
    perms = ops._perms(source.size(1), source.device)[1]
    return loss, max_snr, est, reorder_source(est, perms, idx)


def cal_si_snr_with_pit(source, estimate_source, source_lengths):
    """-> (max_snr [B,1], perms [C!,C], max_snr_idx [B]).  estimate_source is length-masked in place."""
    _, max_snr, _, idx = _pit(source, estimate_source, source_lengths)
    return max_snr, ops._perms(source.size(1), source.device)[1], idx


def reorder_source(source, perms, max_snr_idx):
    """out[b,c] = source[b, perms[idx[b]][c]] -- the reference's rule (src/pit_criterion.py:80-99), which applies
    the permutation rather than its inverse (differs for 3-cycles at C=3; kept for parity)."""
    sel = torch.index_select(perms, dim=0, index=max_snr_idx)          # [B, C]
    return torch.gather(source, 1, sel.unsqueeze(-1).expand_as(source))


def get_mask(source, source_lengths):
    """[B,1,T] ones where t < length (src/pit_criterion.py:102-114)."""
    T = source.size(-1)
    t = torch.arange(T, device=source.device).view(1, 1, T)
    return (t < source_lengths.to(source.device).view(-1, 1, 1)).to(source.dtype)
