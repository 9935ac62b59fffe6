"""Dataset / collate with the contract of the reference (RecurrentGAN.py:98-308): CSV rows of
the 23-column telemetry format are segmented at `total_duration == 0`, turned into
(dx, dy, dt) sequences normalised by the screen size, mirrored x4, and collated into
{starts, ends, sequences, lengths} sorted by length (descending) with zero padding."""
from __future__ import annotations

import csv
from pathlib import Path
from typing import Dict, Iterator, List, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset, Sampler

from .config import Config


class MouseTrajectoryDataset(Dataset):
    def __init__(self, data_path: str, config: Config, normalize: bool = True, augment: bool = True):
        self.config, self.normalize, self.augment = config, normalize, augment
        self.trajectories: List[Dict] = []
        path = Path(data_path)
        files = [path] if path.is_file() else list(path.glob('*.csv'))
        for f in files:
            self._load_csv(f)
        base = len(self.trajectories)
        if augment:
            self._augment_trajectories()
        print(f'Loaded {base} base trajectories from {len(files)} files')
        if augment:
            print(f'After augmentation: {len(self.trajectories)} trajectories (4x via mirroring)')

    def _load_csv(self, csv_path: Path) -> None:
        xs: List[float] = []; ys: List[float] = []; dts: List[float] = []
        with open(csv_path, 'r') as fh:
            for row in csv.DictReader(fh):
                if float(row['total_duration']) == 0 and xs:          # RG.py:151: reset = new trajectory
                    self._process_trajectory(xs, ys, dts)
                    xs, ys, dts = [], [], []
                xs.append(float(row['position_x'])); ys.append(float(row['position_y']))
                dts.append(float(row['time_between_movements']))
        if xs:
            self._process_trajectory(xs, ys, dts)

    def _process_trajectory(self, xs, ys, dts) -> None:
        c = self.config
        n = len(xs)
        if n < c.min_trajectory_length:
            return
        n = min(n, c.max_trajectory_length)
        pos = np.stack([np.asarray(xs[:n], np.float32), np.asarray(ys[:n], np.float32)], axis=1)
        dt = np.maximum(np.asarray(dts[:n], np.float32)[1:], 1e-6)                  # RG.py:179-182
        deltas = np.diff(pos, axis=0)
        if self.normalize:
            scale = np.array([c.screen_width, c.screen_height])
            start, end, deltas = pos[0] / scale, pos[-1] / scale, deltas / scale
        else:
            start, end = pos[0], pos[-1]
        self.trajectories.append({'start': torch.tensor(start, dtype=torch.float32),
                                  'end': torch.tensor(end, dtype=torch.float32),
                                  'deltas': torch.tensor(deltas, dtype=torch.float32),
                                  'dts': torch.tensor(dt, dtype=torch.float32), 'length': len(deltas)})

    def _augment_trajectories(self) -> None:
        """Mirrors of RG.py:205-272: h-flip, v-flip, both."""
        extra = []
        for tr in self.trajectories:
            for fx, fy in ((True, False), (False, True), (True, True)):
                s, e, d = tr['start'].clone(), tr['end'].clone(), tr['deltas'].clone()
                if fx:
                    s[0], e[0] = 1.0 - s[0].item(), 1.0 - e[0].item()
                    d[:, 0] = -d[:, 0]
                if fy:
                    s[1], e[1] = 1.0 - s[1].item(), 1.0 - e[1].item()
                    d[:, 1] = -d[:, 1]
                extra.append({'start': s, 'end': e, 'deltas': d, 'dts': tr['dts'].clone(), 'length': tr['length']})
        self.trajectories.extend(extra)

    def __len__(self) -> int:
        return len(self.trajectories)

    def __getitem__(self, idx: int) -> Dict:
        return self.trajectories[idx]


def collate_trajectories(batch: List[Dict]) -> Dict:
    """RG.py:281-308."""
    batch = sorted(batch, key=lambda it: it['length'], reverse=True)
    B, T = len(batch), batch[0]['length']
    seq = torch.zeros(B, T, 3)
    for i, it in enumerate(batch):
        n = it['length']
        seq[i, :n, :2] = it['deltas']
        seq[i, :n, 2] = it['dts']
    return {'starts': torch.stack([it['start'] for it in batch]), 'ends': torch.stack([it['end'] for it in batch]),
            'sequences': seq, 'lengths': torch.tensor([it['length'] for it in batch])}


class LengthBucketedBatchSampler(Sampler):
    """Batches of trajectories of similar length (not in the reference, which draws batches at
    random and pads each to its longest member, RG.py:301 - and the critic then encodes and runs
    its LSTM over every padded position, RG.py:761).  The persistent kernels skip whole steps only
    beyond the longest row of a 32-trajectory tile, so what padding costs is set by how ragged a
    batch is.  Indices are sorted by length, cut into batches, and the batches (not their
    content) are shuffled each epoch; with `jitter` > 0 the sort key is perturbed so that batch
    membership also changes between epochs.

        DataLoader(ds, batch_sampler=LengthBucketedBatchSampler(lengths, 64), collate_fn=collate_trajectories)

    Every index appears exactly once per epoch; the last batch may be short unless drop_last."""

    def __init__(self, lengths: Sequence[int], batch_size: int, shuffle: bool = True, drop_last: bool = False,
                 jitter: float = 0.0, seed: int = 0):
        self.lengths = np.asarray(lengths, np.float64)
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), shuffle, drop_last
        self.jitter, self.seed, self.epoch = float(jitter), int(seed), 0

    @classmethod
    def from_dataset(cls, dataset: 'MouseTrajectoryDataset', batch_size: int, **kw) -> 'LengthBucketedBatchSampler':
        return cls([tr['length'] for tr in dataset.trajectories], batch_size, **kw)

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def __len__(self) -> int:
        n, b = len(self.lengths), self.batch_size
        return n // b if self.drop_last else (n + b - 1) // b

    def __iter__(self) -> Iterator[List[int]]:
        rng = np.random.default_rng([self.seed, self.epoch])
        key = self.lengths
        if self.jitter > 0:
            key = key * (1.0 + self.jitter * rng.uniform(-1.0, 1.0, len(key)))
        order = np.argsort(-key, kind='stable')
        b = self.batch_size
        batches = [order[i:i + b].tolist() for i in range(0, len(order), b)]
        if self.drop_last and batches and len(batches[-1]) < b:
            batches.pop()
        if self.shuffle:
            batches = [batches[i] for i in rng.permutation(len(batches))]
        self.epoch += 1
        return iter(batches)


def padding_fraction(lengths: Sequence[int], batches: Sequence[Sequence[int]]) -> float:
    """Share of padded (wasted) positions when each batch is padded to its longest member."""
    lengths = np.asarray(lengths)
    used = sum(int(lengths[list(b)].sum()) for b in batches)
    padded = sum(int(lengths[list(b)].max()) * len(b) for b in batches)
    return 1.0 - used / max(padded, 1)
