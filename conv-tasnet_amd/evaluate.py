"""SI-SNRi evaluation (src/evaluate.py:21-130): forward + PIT reorder on the GPU, the numpy SI-SNR metric on host.

``evaluate(model_path, data_dir, calc_sdr, use_cuda, sample_rate, batch_size)`` is the reference's entry point
(src/evaluate.py:21): it loads the checkpoint, reads data_dir/{mix,s1,s2}.json through data.AudioDataset (full
utterances, scipy wav reader) and prints per-utterance and average SI-SNRi.  ``evaluate_loader(model, data_loader)`` is
the same loop over any iterable of (padded_mixture, mixture_lengths, padded_source) batches.  calc_sdr needs mir_eval's
bss_eval_sources (third party, CPU, "very very slow" :78): outside the hot-path scope, an explicit error here.
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


def evaluate(model_path, data_dir, calc_sdr=0, use_cuda=1, sample_rate=8000, batch_size=1):
    """The reference's signature (src/evaluate.py:21-73).  -> average SI-SNRi over data_dir's utterances."""
    if calc_sdr:
        raise NotImplementedError("calc_sdr needs mir_eval.separation.bss_eval_sources (third party, CPU only): outside "
                                  "the hot-path scope -- SI-SNRi is computed, SDRi is not")
    from .data import AudioDataLoader, AudioDataset
    model = ConvTasNet.load_model(model_path)
    print(model)
    dataset = AudioDataset(data_dir, batch_size, sample_rate=sample_rate, segment=-1)
    data_loader = AudioDataLoader(dataset, batch_size=1, num_workers=2)
    return evaluate_loader(model, data_loader, use_cuda=bool(use_cuda))


def evaluate_loader(model, data_loader, use_cuda=True, verbose=True):
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
