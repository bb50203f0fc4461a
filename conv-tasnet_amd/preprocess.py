"""Manifest writer for the loaders: <json_dir>/<split>/{mix,s1,...}.json = [[abs_wav_path, n_samples], ...].

Same file format and directory convention as the reference's src/preprocess.py:12-36 (consumed by
src/data.py:52-60 and by conv_tasnet_amd.data.AudioDataset), so manifests written by either side are
interchangeable.  The sample count comes from the wav header instead of decoding the audio with librosa; files
must already be at `sample_rate` (resampling is outside the hot-path scope, and a mismatch is an error rather
than a silent length change).
"""
import argparse
import json
import os
import wave


def wav_num_samples(path, sample_rate):
    try:
        with wave.open(path, "rb") as w:
            sr, n = w.getframerate(), w.getnframes()
    except wave.Error:                       # float / extensible wav: fall back to scipy's reader
        from scipy.io import wavfile
        sr, x = wavfile.read(path, mmap=True)
        n = x.shape[0]
    if sr != sample_rate:
        raise ValueError("%s is at %d Hz, expected %d" % (path, sr, sample_rate))
    return int(n)


def preprocess_one_dir(data_dir, json_dir, json_filename, sample_rate=8000):
    data_dir = os.path.abspath(data_dir)
    infos = []
    for name in os.listdir(data_dir):
        if name.endswith(".wav"):
            p = os.path.join(data_dir, name)
            infos.append((p, wav_num_samples(p, sample_rate)))
    os.makedirs(json_dir, exist_ok=True)
    with open(os.path.join(json_dir, json_filename + ".json"), "w") as f:
        json.dump(infos, f, indent=4)
    return infos


def preprocess(data_dir, json_dir, sample_rate=8000, splits=("tr", "cv", "tt"), num_speakers=2):
    for split in splits:
        for spk in ["mix"] + ["s%d" % (c + 1) for c in range(num_speakers)]:
            preprocess_one_dir(os.path.join(data_dir, split, spk), os.path.join(json_dir, split), spk, sample_rate)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="write mix/s1/s2 json manifests for tr, cv, tt")
    ap.add_argument("data_dir")
    ap.add_argument("json_dir")
    ap.add_argument("--sample-rate", type=int, default=8000)
    ap.add_argument("--num-speakers", type=int, default=2)
    a = ap.parse_args()
    preprocess(a.data_dir, a.json_dir, a.sample_rate, num_speakers=a.num_speakers)
