"""CPU: minibatch planning / segmenting / collation vs the reference loaders (golden made by oracle/make_golden.py
with librosa stubbed by a synthetic reader, so only the reference's own planning logic produced it)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from conv_tasnet_amd import data as D


def synth_reader(path, sr):
    base = os.path.basename(path)[:-4]
    _, uid, tag, n = base.split("_")
    rs = np.random.RandomState(1000 * int(uid) + {"mix": 0, "s1": 1, "s2": 2}[tag])
    return rs.uniform(-1, 1, int(n)).astype(np.float32)


@pytest.fixture()
def manifest(tmp_path):
    g = load_golden("data_plan")
    for tag in ("mix", "s1", "s2"):
        infos = [["/fake/utt_%d_%s_%d.wav" % (i, tag, n), int(n)] for i, n in enumerate(g["lens"])]
        (tmp_path / (tag + ".json")).write_text(json.dumps(infos))
    return g, str(tmp_path)


@pytest.mark.parametrize("name,kw", [("tr_b3", dict(batch_size=3, segment=0.5)), ("tr_b5", dict(batch_size=5, segment=0.5)),
                                     ("cv_b2", dict(batch_size=2, segment=-1, cv_maxlen=2.0)),
                                     ("cv_b4", dict(batch_size=4, segment=-1, cv_maxlen=8.0))])
def test_plan_and_collation_match_reference(manifest, name, kw):
    g, jdir = manifest
    ds = D.AudioDataset(jdir, sample_rate=int(g["sample_rate"]), reader=synth_reader, **kw)
    off = g[name + ":off"]
    assert len(ds) == len(off) - 1
    loader = D.AudioDataLoader(ds, shuffle=False, num_workers=0)
    for b, (mix, lens, src) in enumerate(loader):
        ids = [int(os.path.basename(ds.mix[u][0]).split("_")[1]) for u in ds[b]]
        assert ids == list(g[name + ":ids"][off[b]:off[b + 1]])
        ref_lens = [int(v) for v in g[name + ":lens"][b] if v >= 0]
        assert lens.tolist() == ref_lens and lens.dtype == torch.int64
        assert mix.dtype == torch.float32 and mix.shape == (len(ref_lens), max(ref_lens))
        assert src.shape == (len(ref_lens), 2, max(ref_lens)) and src.is_contiguous()
        assert abs(float(mix.double().sum()) - float(g[name + ":sum_mix"][b])) < 1e-6
        w = torch.arange(1, 3).view(1, -1, 1)
        assert abs(float((src.double() * w).sum()) - float(g[name + ":sum_src"][b])) < 1e-6


def test_segment_rule():
    # 2.6 segments, batch of 3: two full segments + the tail segment (overlapping)
    assert D.segment_slices(10400, 4000, 3) == [(0, 4000), (4000, 8000), (6400, 10400)]
    # exact multiple: no tail
    assert D.segment_slices(8000, 4000, 3) == [(0, 4000), (4000, 8000)]
    # longer than the batch: only batch_size segments, no tail
    assert D.segment_slices(17000, 4000, 3) == [(0, 4000), (4000, 8000), (8000, 12000)]


def test_rank_sharding_partitions_the_plan(manifest):
    """Training plans: disjoint shards with EQUAL step counts (every step ends in a collective; the remainder of
    len(plan) % world minibatches is dropped).  Validation plans (segment < 0): every minibatch kept, counts may differ."""
    g, jdir = manifest
    for world in (2, 3):
        full = D.AudioDataset(jdir, batch_size=3, segment=0.5, reader=synth_reader).plan
        parts = [D.AudioDataset(jdir, batch_size=3, segment=0.5, reader=synth_reader, rank=r, world=world).plan
                 for r in range(world)]
        assert len(set(len(p) for p in parts)) == 1 and len(parts[0]) == len(full) // world
        dealt = sum(parts, [])
        assert len(dealt) == len(full) // world * world                       # disjoint: every kept minibatch exactly once
        assert len(set(map(tuple, dealt))) == len(dealt) and set(map(tuple, dealt)) <= set(map(tuple, full))
        # the deal changes with the epoch (same permutation on every rank), so over a run every minibatch is trained on and
        # no rank keeps a fixed subset of the length-sorted plan
        dss = [D.AudioDataset(jdir, batch_size=3, segment=0.5, reader=synth_reader, rank=r, world=world) for r in range(world)]
        seen, rank0 = set(), []
        for epoch in range(12):
            for ds in dss:
                ds.set_epoch(epoch)
            ep = sum((ds.plan for ds in dss), [])
            assert len(set(map(tuple, ep))) == len(ep) == len(full) // world * world
            seen |= set(map(tuple, ep))
            rank0.append(tuple(map(tuple, dss[0].plan)))
        assert len(set(rank0)) > 1
        if len(full) % world:
            assert seen == set(map(tuple, full))
        one = D.AudioDataset(jdir, batch_size=3, segment=0.5, reader=synth_reader)
        one.set_epoch(5)
        assert one.plan == full                                               # single process: the reference's plan, untouched
        cv_full = D.AudioDataset(jdir, batch_size=1, segment=-1, cv_maxlen=100, reader=synth_reader).plan
        cv_parts = [D.AudioDataset(jdir, batch_size=1, segment=-1, cv_maxlen=100, reader=synth_reader, rank=r, world=world).plan
                    for r in range(world)]
        assert sorted(map(tuple, sum(cv_parts, []))) == sorted(map(tuple, cv_full))


def test_eval_dataset_and_wav_reader(tmp_path):
    from scipy.io import wavfile
    rs = np.random.RandomState(0)
    for i, n in enumerate((900, 1500, 1200)):
        wavfile.write(str(tmp_path / ("m%d.wav" % i)), 8000, (rs.uniform(-0.5, 0.5, n) * 32767).astype(np.int16))
    ds = D.EvalDataset(str(tmp_path), None, batch_size=2, sample_rate=8000)
    batches = list(D.EvalDataLoader(ds))
    assert [b[1].tolist() for b in batches] == [[1500, 1200], [900]]
    mix, lens, names = batches[0]
    assert mix.shape == (2, 1500) and float(mix[1, 1200:].abs().max()) == 0.0
    assert names[0].endswith("m1.wav") and abs(float(mix.abs().max())) <= 0.5 + 1e-4
    with pytest.raises(ValueError):
        D.read_wav(str(tmp_path / "m0.wav"), 16000)


def test_preprocess_writes_reference_format_manifests(tmp_path):
    """tr/cv/tt x mix/s1/s2 json lists of [abs path, n_samples] (src/preprocess.py:12-36), readable by AudioDataset."""
    import json
    from scipy.io import wavfile
    from conv_tasnet_amd.preprocess import preprocess
    from conv_tasnet_amd.data import AudioDataset
    rs = np.random.RandomState(1)
    lens = {"a.wav": 33000, "b.wav": 40000, "c.wav": 9000}
    for split in ("tr", "cv", "tt"):
        for spk in ("mix", "s1", "s2"):
            d = tmp_path / "wav" / split / spk
            d.mkdir(parents=True)
            for name, n in lens.items():
                wavfile.write(str(d / name), 8000, (rs.uniform(-0.3, 0.3, n) * 32767).astype(np.int16))
            (d / "notes.txt").write_text("ignored")
    preprocess(str(tmp_path / "wav"), str(tmp_path / "json"), 8000)
    got = json.load(open(tmp_path / "json" / "tr" / "s2.json"))
    assert sorted((os.path.basename(p), n) for p, n in got) == sorted(lens.items())
    assert all(os.path.isabs(p) for p, _ in got)
    ds = AudioDataset(str(tmp_path / "json" / "tr"), batch_size=3, sample_rate=8000, segment=4.0)
    mix, lengths, src = ds.collate([ds[0]])
    assert mix.shape[1] == 32000 and src.shape[1:] == (2, 32000) and int(lengths[0]) == 32000
    with pytest.raises(ValueError):
        preprocess(str(tmp_path / "wav"), str(tmp_path / "json16"), 16000)
