"""In-memory stand-in for the h5py group protocol (TEST INFRASTRUCTURE; h5py is not part of this image).

Implements what the reference's HDF5 helpers call (`src/aspire/utils.py:733-887`): `require_group`, `create_group`,
`create_dataset(name, data=... | shape=..., maxshape=..., dtype=...)`, `in`, `[]`, `items()`, dataset `resize`,
`shape`, `dtype`, `[...]`, `[()]`, slice assignment.  Slash-separated paths address nested groups.  Used by
`oracle/make_golden.py` (to record the layout the REAL reference writes) and by the tests of `aspire_amd/io.py`.
"""
from __future__ import annotations

import numpy as np


class FakeDataset:
    def __init__(self, data=None, shape=None, maxshape=None, dtype=None):
        if data is not None:
            if isinstance(data, (list, tuple)) and any(isinstance(v, (list, tuple, np.ndarray)) and np.ndim(v) != np.ndim(data[0])
                                                       for v in data):
                raise ValueError("ragged data")  # h5py refuses ragged nested lists
            if isinstance(data, str):
                self._a = np.array(data, dtype=object)
            else:
                self._a = np.array(data)
                # real h5py stores an object array only when its dtype carries the vlen-string tag that
                # h5py.string_dtype() attaches (np.dtype("O", metadata={"vlen": str})); a bare dtype=object array raises
                meta = self._a.dtype.metadata or {}
                if self._a.dtype == object and (meta.get("vlen") not in (str, bytes)
                                                or not all(isinstance(v, (str, bytes)) for v in self._a.ravel())):
                    raise TypeError("Object dtype dtype('O') has no native HDF5 equivalent")
        else:
            self._a = np.zeros(shape, dtype=dtype)
        self.maxshape = maxshape

    @property
    def shape(self):
        return self._a.shape

    @property
    def dtype(self):
        return self._a.dtype

    def resize(self, shape):
        if self.maxshape is None:
            raise TypeError("Only chunked datasets can be resized")
        self._a = np.resize(self._a, shape)

    def __getitem__(self, key):
        if key == () and self._a.shape == ():
            v = self._a[()]
            return v.encode("utf-8") if isinstance(v, str) else v  # h5py hands strings back as bytes
        out = self._a[key]
        return out.copy() if isinstance(out, np.ndarray) else out

    def __setitem__(self, key, value):
        self._a[key] = value


class FakeGroup:
    def __init__(self):
        self._items: dict = {}
        self.attrs: dict = {}

    def _walk(self, path, create):
        node = self
        parts = [p for p in path.split("/") if p]
        for p in parts[:-1]:
            if p not in node._items:
                if not create:
                    raise KeyError(path)
                node._items[p] = FakeGroup()
            node = node._items[p]
        return node, (parts[-1] if parts else "")

    def require_group(self, path):
        node, leaf = self._walk(path, True)
        if leaf not in node._items:
            node._items[leaf] = FakeGroup()
        return node._items[leaf]

    def create_group(self, path):
        node, leaf = self._walk(path, True)
        if leaf in node._items:
            raise ValueError(f"group {path} exists")
        node._items[leaf] = FakeGroup()
        return node._items[leaf]

    def create_dataset(self, name, data=None, shape=None, maxshape=None, dtype=None):
        node, leaf = self._walk(name, True)
        if leaf in node._items:
            raise ValueError(f"dataset {name} exists")
        node._items[leaf] = FakeDataset(data=data, shape=shape, maxshape=maxshape, dtype=dtype)
        return node._items[leaf]

    def __contains__(self, path):
        try:
            node, leaf = self._walk(path, False)
        except KeyError:
            return False
        return leaf in node._items

    def __getitem__(self, path):
        node, leaf = self._walk(path, False)
        return node._items[leaf]

    def items(self):
        return self._items.items()

    def keys(self):
        return self._items.keys()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def layout(self, prefix=""):
        """{full path: (dtype kind, shape)} of every dataset below this group."""
        out = {}
        for k, v in self._items.items():
            p = f"{prefix}/{k}" if prefix else k
            if isinstance(v, FakeGroup):
                out.update(v.layout(p))
            else:
                out[p] = (v.dtype.kind, tuple(v.shape))
        return out


class FakeFile(FakeGroup):
    """`h5py.File(path, mode)` over the in-memory groups above; contents persist per path for the life of the process, so a
    file re-opened in append mode shows what an earlier `with` block wrote (the reference's `AspireFile` subclasses this
    through the shim's h5py stub and opens its checkpoint file several times per `sample_posterior` call)."""

    _store: dict = {}

    def __init__(self, name, mode="r", *args, **kwargs):
        super().__init__()
        name = str(name)
        if mode in ("w", "w-"):
            FakeFile._store.pop(name, None)
        if name not in FakeFile._store:
            if mode in ("r", "r+"):
                raise FileNotFoundError(name)
            FakeFile._store[name] = ({}, {})
        self._items, self.attrs = FakeFile._store[name]
        self.mode, self.filename = ("r+" if mode == "a" else mode), name

    def __delitem__(self, path):
        node, leaf = self._walk(path, False)
        del node._items[leaf]

    def close(self):
        pass
