"""separate(): same signature and output naming as the reference (src/separate.py:17-57).

Writes <base>.wav (the mixture) and <base>_s{c}.wav per speaker into out_dir.  The forward pass is the HIP path.
wav IO uses scipy (librosa is not a dependency here; the reference's librosa.output.write_wav no longer exists
upstream).  File-name quirk kept: ``basename.strip('.wav')`` strips CHARACTERS from both ends (:52-53).
"""
import os

import numpy as np
import torch

from .conv_tasnet import ConvTasNet
from .data import EvalDataLoader, EvalDataset
from .utils import remove_pad


def _write_wav(path, x, sample_rate):
    from scipy.io import wavfile
    wavfile.write(path, sample_rate, np.asarray(x, dtype=np.float32))


def separate(model_path, mix_dir, mix_json, out_dir, use_cuda, sample_rate, batch_size):
    if mix_dir is None and mix_json is None:
        print("Must provide mix_dir or mix_json! When providing mix_dir, mix_json is ignored.")
    model = ConvTasNet.load_model(model_path)
    model.eval()
    if use_cuda:
        model.cuda()
    dev = next(model.parameters()).device
    eval_loader = EvalDataLoader(EvalDataset(mix_dir, mix_json, batch_size=batch_size, sample_rate=sample_rate))
    os.makedirs(out_dir, exist_ok=True)
    with torch.no_grad():
        for mixture, lens, filenames in eval_loader:
            mixture, lens = mixture.to(dev), lens.to(dev)
            estimate_source = model(mixture)                     # [B, C, T]
            flat_estimate = remove_pad(estimate_source, lens)
            mixture_np = remove_pad(mixture, lens)
            for i, path in enumerate(filenames):
                filename = os.path.join(out_dir, os.path.basename(path).strip('.wav'))
                _write_wav(filename + '.wav', mixture_np[i], sample_rate)
                for c in range(flat_estimate[i].shape[0]):
                    _write_wav(filename + '_s{}.wav'.format(c + 1), flat_estimate[i][c], sample_rate)
