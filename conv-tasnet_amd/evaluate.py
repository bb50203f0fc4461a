"""SI-SNRi evaluation (src/evaluate.py:21-130): forward + PIT reorder on the GPU, the numpy SI-SNR metric on host.

``evaluate(model_or_path, data_loader, ...)`` takes any iterable of (padded_mixture, mixture_lengths, padded_source)
batches -- the output contract of the reference's AudioDataLoader (src/data.py:264-300); wav/json reading (librosa)
and SDRi (mir_eval, "very very slow", :78) are outside the hot-path scope.
"""
import numpy as np
import torch

from .conv_tasnet import ConvTasNet
from .pit_criterion import cal_loss
from .utils import remove_pad


def cal_SISNR(ref_sig, out_sig, eps=1e-8):
    """Scale-invariant SNR in dB of one signal pair, float64 numpy (src/evaluate.py:114-130)."""
    assert len(ref_sig) == len(out_sig)
    ref = ref_sig - np.mean(ref_sig)
    out = out_sig - np.mean(out_sig)
    proj = np.sum(ref * out) * ref / (np.sum(ref ** 2) + eps)
    noise = out - proj
    ratio = np.sum(proj ** 2) / (np.sum(noise ** 2) + eps)
    return 10 * np.log(ratio + eps) / np.log(10.0)


def cal_SISNRi(src_ref, src_est, mix):
    """Mean SI-SNR improvement over using the mixture itself; two speakers, like src/evaluate.py:94-111."""
    gains = [cal_SISNR(src_ref[c], src_est[c]) - cal_SISNR(src_ref[c], mix) for c in range(2)]
    return (gains[0] + gains[1]) / 2


def evaluate(model, data_loader, use_cuda=True, verbose=True):
    """-> average SI-SNRi over every utterance of the loader.  `model` is a ConvTasNet or a checkpoint path."""
    if isinstance(model, str):
        model = ConvTasNet.load_model(model)
    model.eval()
    if use_cuda:
        model.cuda()
    dev = next(model.parameters()).device
    total, count = 0.0, 0
    with torch.no_grad():
        for padded_mixture, mixture_lengths, padded_source in data_loader:
            padded_mixture = padded_mixture.to(dev)
            mixture_lengths = mixture_lengths.to(dev)
            padded_source = padded_source.to(dev)
            estimate_source = model(padded_mixture)
            _, _, _, reorder = cal_loss(padded_source, estimate_source, mixture_lengths)
            mixture = remove_pad(padded_mixture, mixture_lengths)
            source = remove_pad(padded_source, mixture_lengths)
            est = remove_pad(reorder, mixture_lengths)       # NOTE: the reordered estimate, as the reference does
            for mix, ref, out in zip(mixture, source, est):
                v = cal_SISNRi(ref.astype(np.float64), out.astype(np.float64), mix.astype(np.float64))
                if verbose:
                    print("Utt %d\tSI-SNRi=%.2f" % (count + 1, v))
                total += v
                count += 1
    avg = total / max(count, 1)
    if verbose:
        print("Average SISNR improvement: {0:.2f}".format(avg))
    return avg
