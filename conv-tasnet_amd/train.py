"""train(): launcher mirroring src/train.py:14-102 (paper config, Adam lr 1e-3, clip 5, half-lr, early stop).

The reference's train() hard-codes librosa/json data loading; here the loaders are arguments (any iterables of
(padded_mixture [B,T], mixture_lengths [B], padded_source [B,C,T]) -- the AudioDataLoader contract) and a synthetic
loader is provided for smoke runs and benchmarks.  One process per GPU:

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m conv_tasnet_amd.train --epochs 1
"""
import argparse

import torch

from . import parallel
from .conv_tasnet import ConvTasNet
from .optim import FlatAdam
from .solver import Solver

PAPER = dict(N=256, L=20, B=256, H=512, P=3, X=8, R=4, C=2, norm_type='gLN', causal=0, mask_nonlinear='relu')


class SyntheticLoader:
    """Deterministic harmonic 2-speaker mixtures, sharded by rank; yields AudioDataLoader-style batches."""

    def __init__(self, n_batches, batch_size, samples=32000, C=2, sample_rate=8000, first_utt=0, rank=0, world=1):
        self.n_batches, self.batch_size, self.samples, self.C, self.sr = n_batches, batch_size, samples, C, sample_rate
        self.first, self.rank, self.world = first_utt, rank, world

    def __len__(self):
        return self.n_batches

    def _utt(self, u):
        import math
        g = torch.Generator().manual_seed(1234 + u)
        t = torch.arange(self.samples, dtype=torch.float64) / self.sr
        out = torch.empty(self.C, self.samples, dtype=torch.float64)
        for c in range(self.C):
            f0 = 80 + 320 * torch.rand(1, generator=g, dtype=torch.float64)
            ph = 2 * math.pi * torch.rand(3, generator=g, dtype=torch.float64)
            s = sum(a * torch.sin(2 * math.pi * (h + 1) * f0 * t + ph[h]) for h, a in enumerate((1.0, 0.5, 0.25)))
            out[c] = s + 0.01 * torch.randn(self.samples, generator=g, dtype=torch.float64)
        return out.float()

    def __iter__(self):
        per = self.batch_size
        for b in range(self.n_batches):
            base = self.first + (b * self.world + self.rank) * per
            src = torch.stack([self._utt(base + i) for i in range(per)])
            yield src.sum(1), torch.full((per,), self.samples, dtype=torch.long), src


def train(data, epochs, model_path, save_folder="exp/models", continue_from="", config=None, lr=1e-3,
          max_grad_norm=5, half_lr=1, early_stop=1, print_freq=10, enable_checkpoint=0):
    """data = {'tr_loader': ..., 'cv_loader': ...}.  Returns the Solver after training."""
    world, rank, device = parallel.init_distributed()
    cfg = dict(PAPER if config is None else config)
    torch.manual_seed(0)
    model = ConvTasNet(cfg['N'], cfg['L'], cfg['B'], cfg['H'], cfg['P'], cfg['X'], cfg['R'], cfg['C'],
                       norm_type=cfg.get('norm_type', 'gLN'), causal=cfg.get('causal', 0),
                       mask_nonlinear=cfg.get('mask_nonlinear', 'relu')).to(device)
    optimizer = FlatAdam(model.parameters(), lr=lr)
    parallel.broadcast_parameters(optimizer.flat_params)
    arg_solver = (1, epochs, half_lr, early_stop, max_grad_norm, save_folder, enable_checkpoint, continue_from,
                  model_path, print_freq, 0, 0, "Conv-TasNet Training")
    solver = Solver(data, model, optimizer, arg_solver)
    solver.train()
    return solver


def main():
    ap = argparse.ArgumentParser(description="Conv-TasNet training on MI355X (synthetic data demo)")
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--batches", type=int, default=10)
    ap.add_argument("--batch-size", type=int, default=8, help="utterances per GPU per step")
    ap.add_argument("--data-dir", default=None, help="directory with tr/ and cv/ sub-directories of {mix,s1,s2}.json "
                    "(the reference's manifest layout); default: synthetic mixtures")
    ap.add_argument("--model-path", default="final.pth.tar")
    ap.add_argument("--save-folder", default="exp/models")
    a = ap.parse_args()
    world, rank, _ = parallel.init_distributed()
    if a.data_dir:
        import os
        from .data import AudioDataLoader, AudioDataset
        tr = AudioDataLoader(AudioDataset(os.path.join(a.data_dir, "tr"), a.batch_size, segment=4.0, rank=rank, world=world),
                             shuffle=True, num_workers=4)
        cv = AudioDataLoader(AudioDataset(os.path.join(a.data_dir, "cv"), 1, segment=-1, cv_maxlen=6, rank=rank, world=world),
                             num_workers=0)
    else:
        tr = SyntheticLoader(a.batches, a.batch_size, rank=rank, world=world)
        cv = SyntheticLoader(1, a.batch_size, first_utt=10 ** 6, rank=rank, world=world)
    train({'tr_loader': tr, 'cv_loader': cv}, a.epochs, a.model_path, save_folder=a.save_folder)


if __name__ == "__main__":
    main()
