"""Memmap dataset over the trainer's on-disk format (mirror of reference dataset.py:6-73).

Reads what ``collect.TupleSink`` (and the reference's convert.py:84-99) writes: ``states.npy``
float16 [N,17,7,10,9], ``mcts.npy`` [N,2086], ``winners.npy`` float32 [N]. Arrays are opened lazily per
process so that DataLoader workers can pickle the dataset (dataset.py:27-44 of the reference).
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import Dataset


class NpyMemmapDataset(Dataset):
    def __init__(self, data_dir: str):
        self.data_dir = data_dir
        self.paths = {k: os.path.join(data_dir, f"{k}.npy") for k in ("states", "mcts", "winners")}
        for p in self.paths.values():
            if not os.path.exists(p):
                raise FileNotFoundError(p)
        self._arrays = None
        n = {k: np.load(p, mmap_mode="r").shape[0] for k, p in self.paths.items()}
        if len(set(n.values())) != 1:  # dataset.py:57-61
            raise ValueError(f"inconsistent lengths: {n}")
        self.length = n["states"]

    def _open(self):
        if self._arrays is None:
            self._arrays = {k: np.load(p, mmap_mode="r") for k, p in self.paths.items()}
        return self._arrays

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_arrays"] = None  # memmaps are re-opened in the worker
        return d

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        a = self._open()
        return (torch.from_numpy(np.array(a["states"][i])), torch.from_numpy(np.array(a["mcts"][i], dtype=np.float32)),
                torch.tensor(float(a["winners"][i])))
