"""A test double for the handful of h5py calls the reference's index file code makes
(/root/reference/scaling_retriever/utils/inverted_index.py:22-55, :84-105): `File(path, mode)` as a context manager,
`create_dataset(name, data=...)`, `name in f`, `f[name]` (array-like; `[()]` for scalars), `keys()`.  h5py is not installed in this
image; with this double the h5py branches of scaling_retriever_amd/utils/inverted_index.py at least EXECUTE in the CPU suite
(against the real library the same test runs unchanged).  Datasets persist as a pickle at the file's path."""
import os
import pickle

import numpy as np


class _Dataset:
    def __init__(self, a):
        self._a = np.asarray(a)

    def __getitem__(self, key):
        return self._a[key]

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)

    def __len__(self):
        return len(self._a)

    @property
    def shape(self):
        return self._a.shape


class File:
    def __init__(self, path, mode="r"):
        self._path, self._mode, self._d = path, mode, {}
        if mode == "r":
            if not os.path.exists(path):
                raise OSError(f"unable to open {path}")
            with open(path, "rb") as f:
                self._d = pickle.load(f)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self._mode in ("w", "a"):
            with open(self._path, "wb") as f:
                pickle.dump(self._d, f)
        return False

    def create_dataset(self, name, data=None):
        if name in self._d:
            raise ValueError(f"dataset {name} exists")
        self._d[name] = np.asarray(data)

    def __contains__(self, name):
        return name in self._d

    def __getitem__(self, name):
        return _Dataset(self._d[name])

    def keys(self):
        return self._d.keys()
