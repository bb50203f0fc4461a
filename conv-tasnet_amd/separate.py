"""separate(): same signature and output naming as the reference (src/separate.py:17-57).

Writes <base>.wav (the mixture) and <base>_s{c}.wav per speaker into out_dir.  The forward pass is the HIP path.
wav IO uses scipy (librosa is not a dependency here; the reference's librosa.output.write_wav no longer exists
upstream).  File-name quirk kept: ``basename.strip('.wav')`` strips CHARACTERS from both ends (:52-53).
"""
import json
import os

import numpy as np
import torch

from .conv_tasnet import ConvTasNet
from .utils import remove_pad


def _read_wav(path, sample_rate):
    from scipy.io import wavfile
    sr, x = wavfile.read(path)
    if sr != sample_rate:
        raise ValueError("%s: sample rate %d != %d (resampling is outside the hot-path scope)" % (path, sr, sample_rate))
    if x.dtype == np.int16:
        x = x.astype(np.float32) / 32768.0
    elif x.dtype == np.int32:
        x = x.astype(np.float32) / 2147483648.0
    x = x.astype(np.float32)
    return x.mean(axis=1) if x.ndim == 2 else x


def _write_wav(path, x, sample_rate):
    from scipy.io import wavfile
    wavfile.write(path, sample_rate, np.asarray(x, dtype=np.float32))


def _mixture_list(mix_dir, mix_json):
    """(path, n_samples) entries, longest first, as EvalDataset sorts them (src/data.py:186-225)."""
    assert mix_dir is not None or mix_json is not None
    if mix_dir is not None:
        from scipy.io import wavfile
        infos = []
        for name in sorted(os.listdir(mix_dir)):
            if name.endswith('.wav'):
                p = os.path.join(os.path.abspath(mix_dir), name)
                infos.append((p, len(wavfile.read(p)[1])))
    else:
        with open(mix_json, 'r') as f:
            infos = [tuple(e) for e in json.load(f)]
    return sorted(infos, key=lambda info: int(info[1]), reverse=True)


def separate(model_path, mix_dir, mix_json, out_dir, use_cuda, sample_rate, batch_size):
    if mix_dir is None and mix_json is None:
        print("Must provide mix_dir or mix_json! When providing mix_dir, mix_json is ignored.")
    model = ConvTasNet.load_model(model_path)
    model.eval()
    if use_cuda:
        model.cuda()
    dev = next(model.parameters()).device
    infos = _mixture_list(mix_dir, mix_json)
    os.makedirs(out_dir, exist_ok=True)
    with torch.no_grad():
        for start in range(0, len(infos), batch_size):
            batch = infos[start:start + batch_size]
            waves = [_read_wav(p, sample_rate) for p, _ in batch]
            lens = torch.tensor([len(w) for w in waves], dtype=torch.long)
            mixture = torch.zeros(len(waves), int(lens.max()))
            for i, w in enumerate(waves):
                mixture[i, :len(w)] = torch.from_numpy(w)
            mixture, lens = mixture.to(dev), lens.to(dev)
            estimate_source = model(mixture)                     # [B, C, T]
            flat_estimate = remove_pad(estimate_source, lens)
            mixture_np = remove_pad(mixture, lens)
            for i, (path, _) in enumerate(batch):
                filename = os.path.join(out_dir, os.path.basename(path).strip('.wav'))
                _write_wav(filename + '.wav', mixture_np[i], sample_rate)
                for c in range(flat_estimate[i].shape[0]):
                    _write_wav(filename + '_s{}.wav'.format(c + 1), flat_estimate[i][c], sample_rate)
