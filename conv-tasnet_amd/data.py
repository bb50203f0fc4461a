"""Minibatch planning and collation with the reference's loader contract (src/data.py:32-331).

Output contract consumed by Solver / evaluate / separate:
    training / cv batch : (padded_mixture [B,T] f32, mixture_lengths [B] i64, padded_source [B,C,T] f32)
    eval batch          : (padded_mixture [B,T] f32, mixture_lengths [B] i64, filenames list[str])

Planning rules kept from the reference:
  * manifests are json lists of (wav_path, n_samples), sorted longest first (bucketing);
  * training (segment >= 0): a minibatch collects consecutive utterances until their 4 s segment count reaches
    `batch_size`; utterances shorter than one segment are skipped; an utterance that alone exceeds the batch is
    only admitted as the first of a minibatch; inside a minibatch every utterance contributes its consecutive full
    segments (at most batch_size of them) plus, if it is not a whole number of segments and shorter than the batch,
    its LAST segment_len samples (overlapping the previous segment);
  * validation (segment < 0): whole utterances, `batch_size` per minibatch, minibatches whose first (longest)
    utterance exceeds cv_maxlen seconds are skipped;
  * the torch DataLoader on top always runs with batch_size=1: one item IS one minibatch.

What is different: wav files are read with scipy (librosa is not a dependency; files must already be at
`sample_rate`), every dataset can shard its minibatches over data-parallel ranks (rank r takes minibatches
r, r+world, ...), and the planning is a pure function that can be tested without audio files.
"""
import json
import math
import os

import numpy as np
import torch
import torch.utils.data as data


# ----------------------------------------------------------------------------------------------------
# pure planning helpers
# ----------------------------------------------------------------------------------------------------
def sort_infos(infos):
    return sorted(infos, key=lambda info: int(info[1]), reverse=True)


def plan_training_minibatches(lengths, batch_size, segment_len, sample_rate=8000, max_hours=None):
    """lengths: utterance lengths sorted longest first -> list of lists of utterance indices (src/data.py:78-115)."""
    plan, start, hours, n = [], 0, 0.0, len(lengths)
    while True:
        segs, i, part = 0, start, []
        while segs < batch_size and i < n:
            ulen = int(lengths[i])
            if ulen >= segment_len:
                segs += math.ceil(ulen / segment_len)
                if segs > batch_size and start != i:
                    break
                part.append(i)
                hours += min(ulen, segment_len * batch_size) / sample_rate / 3600
            i += 1
        if part:
            plan.append(part)
        if i == n:
            break
        if max_hours is not None and hours > max_hours:
            break
        start = i
    return plan


def plan_full_utterance_minibatches(lengths, batch_size, sample_rate=8000, cv_maxlen=8.0, max_hours=None):
    """Whole-utterance minibatches for validation / test (src/data.py:116-139)."""
    plan, start, hours, n = [], 0, 0.0, len(lengths)
    while True:
        end = min(n, start + batch_size)
        if int(lengths[start]) > cv_maxlen * sample_rate:
            start = end
            if start >= n:
                break
            continue
        hours += int(lengths[start]) / sample_rate / 3600
        plan.append(list(range(start, end)))
        if end == n:
            break
        if max_hours is not None and hours > max_hours:
            break
        start = end
    return plan


def segment_slices(utt_len, segment_len, batch_size):
    """(start, stop) sample ranges one utterance contributes to a training minibatch (src/data.py:287-296)."""
    out = []
    max_index = min(utt_len - segment_len + 1, (batch_size - 1) * segment_len + 1)
    for i in range(0, max_index, segment_len):
        out.append((i, i + segment_len))
    if utt_len % segment_len != 0 and utt_len < batch_size * segment_len:
        out.append((utt_len - segment_len, utt_len))
    return out


def pad_stack(arrays, pad_value=0.0):
    """list of [T_i, ...] arrays -> zero-padded tensor [B, T_max, ...] (src/data.py:320-331)."""
    tmax = max(a.shape[0] for a in arrays)
    out = torch.full((len(arrays), tmax) + tuple(arrays[0].shape[1:]), float(pad_value), dtype=torch.float32)
    for i, a in enumerate(arrays):
        out[i, :a.shape[0]] = torch.as_tensor(a, dtype=torch.float32)
    return out


# ----------------------------------------------------------------------------------------------------
# wav reading
# ----------------------------------------------------------------------------------------------------
def read_wav(path, sample_rate):
    """mono float32 in [-1, 1) at `sample_rate`, like librosa.load(path, sr) for files already at that rate."""
    from scipy.io import wavfile
    sr, x = wavfile.read(path)
    if sr != sample_rate:
        raise ValueError("%s is at %d Hz, expected %d (resampling is outside the hot-path scope)" % (path, sr, sample_rate))
    if x.dtype == np.int16:
        x = x.astype(np.float32) / 32768.0
    elif x.dtype == np.int32:
        x = x.astype(np.float32) / 2147483648.0
    elif x.dtype == np.uint8:
        x = (x.astype(np.float32) - 128.0) / 128.0
    x = x.astype(np.float32)
    return x.mean(axis=1) if x.ndim == 2 else x


def _read_manifest(json_dir, name):
    with open(os.path.join(json_dir, name + ".json"), "r") as f:
        return json.load(f)


def shard_plan(plan, rank, world, equal_counts, epoch=0):
    """This rank's minibatches of a planned epoch.

    equal_counts (training): every step ends in a gradient all-reduce, so all ranks must run the SAME number of steps --
    the remainder len(plan) % world is dropped (at most world - 1 minibatches per epoch; the reference's single process
    has no such constraint, src/data.py:82-113).  The plan is built from length-sorted utterances, so a static cut would
    drop the same minibatches and deal every rank the same subset in every epoch: the plan is first permuted with a
    generator seeded by the epoch alone (identical on every rank), so the dropped remainder and the rank assignment change
    from epoch to epoch and every minibatch is trained on over a run (AudioDataset.set_epoch).  Ragged minibatch SIZES stay
    exact: the Solver weights each rank's gradient by its share of the global minibatch.  Validation minibatches have no
    per-step collective, keep their order and keep all."""
    if world <= 1:
        return plan
    if equal_counts:
        g = torch.Generator().manual_seed(0x5eed + int(epoch))
        order = torch.randperm(len(plan), generator=g).tolist()
        plan = [plan[i] for i in order][: len(plan) // world * world]
    return plan[rank::world]


# ----------------------------------------------------------------------------------------------------
# datasets / loaders
# ----------------------------------------------------------------------------------------------------
class AudioDataset(data.Dataset):
    """json_dir holds mix.json, s1.json, s2.json (... sC.json).  One item = one planned minibatch."""

    def __init__(self, json_dir, batch_size, sample_rate=8000, segment=4.0, cv_maxlen=8.0, max_hours=None,
                 num_speakers=2, rank=0, world=1, reader=read_wav):
        super().__init__()
        self.sample_rate, self.batch_size, self.reader = sample_rate, batch_size, reader
        self.mix = sort_infos(_read_manifest(json_dir, "mix"))
        self.srcs = [sort_infos(_read_manifest(json_dir, "s%d" % (c + 1))) for c in range(num_speakers)]
        lengths = [int(i[1]) for i in self.mix]
        if segment >= 0.0:
            self.segment_len = int(segment * sample_rate)
            plan = plan_training_minibatches(lengths, batch_size, self.segment_len, sample_rate, max_hours)
        else:
            self.segment_len = -1
            plan = plan_full_utterance_minibatches(lengths, batch_size, sample_rate, cv_maxlen, max_hours)
        self.full_plan, self.rank, self.world = plan, rank, world
        self.plan = shard_plan(plan, rank, world, equal_counts=segment >= 0.0)

    def set_epoch(self, epoch):
        """Re-deal the training minibatches over the ranks for this epoch (same permutation on every rank; a no-op for one
        process and for full-utterance validation sets).  The Solver calls it at the top of every epoch."""
        self.plan = shard_plan(self.full_plan, self.rank, self.world, equal_counts=self.segment_len >= 0, epoch=epoch)

    def __len__(self):
        return len(self.plan)

    def __getitem__(self, index):
        return self.plan[index]

    def load(self, utt_indices):
        """-> (mixture segments [T_i], source segments [T_i, C]) of one minibatch."""
        mixes, sources = [], []
        for u in utt_indices:
            assert all(s[u][1] == self.mix[u][1] for s in self.srcs)
            mix = self.reader(self.mix[u][0], self.sample_rate)
            src = np.stack([self.reader(s[u][0], self.sample_rate) for s in self.srcs], axis=1)   # [T, C]
            if self.segment_len >= 0:
                for a, b in segment_slices(mix.shape[-1], self.segment_len, self.batch_size):
                    mixes.append(mix[a:b])
                    sources.append(src[a:b])
            else:
                mixes.append(mix)
                sources.append(src)
        return mixes, sources

    def collate(self, batch):
        assert len(batch) == 1
        mixes, sources = self.load(batch[0])
        lengths = torch.from_numpy(np.array([m.shape[0] for m in mixes]))
        return pad_stack(mixes), lengths, pad_stack(sources).permute(0, 2, 1).contiguous()


class AudioDataLoader(data.DataLoader):
    """DataLoader(batch_size=1) whose collate turns one planned minibatch into tensors (src/data.py:142-182)."""

    def __init__(self, dataset, *args, **kwargs):
        kwargs.setdefault("batch_size", 1)
        super().__init__(dataset, *args, **kwargs)
        self.collate_fn = dataset.collate


class EvalDataset(data.Dataset):
    """Mixtures only, longest first, `batch_size` per minibatch (src/data.py:186-225)."""

    def __init__(self, mix_dir, mix_json, batch_size, sample_rate=8000, reader=read_wav):
        super().__init__()
        assert mix_dir is not None or mix_json is not None
        self.sample_rate, self.reader = sample_rate, reader
        if mix_dir is not None:
            from scipy.io import wavfile
            infos = []
            for name in sorted(os.listdir(mix_dir)):
                if name.endswith(".wav"):
                    p = os.path.join(os.path.abspath(mix_dir), name)
                    infos.append((p, len(wavfile.read(p)[1])))
        else:
            with open(mix_json, "r") as f:
                infos = json.load(f)
        self.infos = sort_infos(infos)
        self.plan = [list(range(s, min(len(self.infos), s + batch_size))) for s in range(0, len(self.infos), batch_size)]

    def __len__(self):
        return len(self.plan)

    def __getitem__(self, index):
        return self.plan[index]

    def collate(self, batch):
        assert len(batch) == 1
        paths = [self.infos[u][0] for u in batch[0]]
        mixes = [self.reader(p, self.sample_rate) for p in paths]
        lengths = torch.from_numpy(np.array([m.shape[0] for m in mixes]))
        return pad_stack(mixes), lengths, paths


class EvalDataLoader(data.DataLoader):
    def __init__(self, dataset, *args, **kwargs):
        kwargs.setdefault("batch_size", 1)
        super().__init__(dataset, *args, **kwargs)
        self.collate_fn = dataset.collate
