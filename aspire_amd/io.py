"""HDF5 layout of the reference's saved state, written through the h5py *protocol*.

Reference: `src/aspire/utils.py:652-887` (`encode_for_hdf5`, `dump_pickle_to_hdf`, `dump_state`,
`recursively_save_to_h5_file`, `load_from_h5_file`) and the checkpoint layout documented in
`docs/checkpointing.rst:17-26`: the sampler state is ONE pickled blob stored as an `S1` byte dataset
`/checkpoint/state`; histories are groups of flattened `a.b.c` datasets.

Everything here only uses the group protocol (`require_group`, `create_dataset`, `__contains__`, `__getitem__`,
`resize`, slice assignment, `items`), so it works on a real `h5py.File` when h5py is installed and on any object that
implements the protocol (tests use an in-memory one: h5py is not part of this image).
"""
from __future__ import annotations

import pickle
from io import BytesIO
from pathlib import Path
from typing import Any

import numpy as np

NONE_TOKEN = "__none__"  # utils.py:656-657
EMPTY_DICT_TOKEN = "__empty_dict__"


def open_h5(path, mode: str = "r"):
    """An open `h5py.File` (with the reference's `aspire_version` attribute on write, utils.py:910-921)."""
    try:
        import h5py
    except ImportError as exc:  # pragma: no cover - h5py is absent from the build image
        raise RuntimeError("HDF5 files need h5py, which is not installed; use a .pkl checkpoint path or pass an open "
                           "h5py-compatible group") from exc
    f = h5py.File(path, mode)
    if f.mode in {"r+", "w", "w-", "a"}:
        from . import __version__

        f.attrs["aspire_version"] = __version__
    return f


def _string_array(values) -> np.ndarray:
    """A list of str as an array h5py can store.  A bare `dtype=object` array carries no vlen-string metadata and real h5py
    refuses it ("Object dtype dtype('O') has no native HDF5 equivalent"): with h5py importable this is the reference's
    `h5py.string_dtype("utf-8")` array (utils.py:665-668); without it, fixed-width UTF-8 bytes (`dtype="S"`), which
    `decode_from_hdf5` reads back to the same list."""
    try:
        import h5py

        return np.array(list(values), dtype=h5py.string_dtype(encoding="utf-8"))
    except ImportError:
        return np.array([v.encode("utf-8") for v in values], dtype="S")


def encode_for_hdf5(value: Any) -> Any:
    """utils.py:652-688 (the cases the SMC path produces: arrays, scalars, strings, lists, dicts, None)."""
    try:
        import torch

        if isinstance(value, torch.Tensor):
            return value.detach().cpu().numpy()
    except ImportError:  # pragma: no cover
        pass
    if isinstance(value, np.ndarray):
        return value
    if isinstance(value, Path):
        value = str(value)
    if isinstance(value, (int, float, str)):
        return value
    if isinstance(value, (list, tuple)):
        if all(isinstance(v, str) for v in value):
            return _string_array(value)
        return [encode_for_hdf5(v) for v in value]
    if isinstance(value, dict):
        if not value:
            return EMPTY_DICT_TOKEN
        return {k: encode_for_hdf5(v) for k, v in value.items()}
    if value is None:
        return NONE_TOKEN
    return value


def decode_from_hdf5(value: Any) -> Any:
    """utils.py:691-730."""
    if isinstance(value, bytes):
        value = value.decode("utf-8")
    if isinstance(value, str):
        if value == NONE_TOKEN:
            return None
        if value == EMPTY_DICT_TOKEN:
            return {}
        return value
    if isinstance(value, np.ndarray):
        if value.shape == ():
            return decode_from_hdf5(value.item())
        if value.dtype.kind in {"S", "O", "U"}:
            try:
                return value.astype(str).tolist()
            except Exception:
                return value
        return value
    if isinstance(value, list):
        return [decode_from_hdf5(v) for v in value]
    if isinstance(value, tuple):
        return tuple(decode_from_hdf5(v) for v in value)
    if isinstance(value, dict):
        return {(k.decode("utf-8") if isinstance(k, bytes) else k): decode_from_hdf5(v) for k, v in value.items()}
    return value


def recursively_save_to_h5_file(h5_file, path: str, dictionary: dict) -> None:
    """utils.py:841-872: nested dictionaries become flattened `a.b.c` datasets under the group `path`."""
    group = h5_file.require_group(path)

    def _save(prefix, d):
        for key, value in d.items():
            full_key = f"{prefix}.{key}" if prefix else key
            if isinstance(value, dict):
                _save(full_key, value)
            else:
                try:
                    group.create_dataset(full_key, data=encode_for_hdf5(value))
                except (TypeError, ValueError):
                    group.create_dataset(full_key, data=_string_array([str(value)])[0])

    _save("", dictionary)


def load_from_h5_file(h5_file, path: str) -> dict:
    """utils.py:875-887."""
    result: dict = {}
    for key, dataset in h5_file[path].items():
        parts = key.split(".")
        d = result
        for part in parts[:-1]:
            d = d.setdefault(part, {})
        d[parts[-1]] = decode_from_hdf5(dataset[()])
    return result


def dump_pickle_to_hdf(memfp: BytesIO, fp, path: str | None = None, dsetname: str = "state") -> None:
    """utils.py:733-757: the pickled bytes as an `S1` dataset, created resizable and overwritten in place."""
    memfp.seek(0)
    bdata = np.frombuffer(memfp.read(), dtype="S1")
    target = fp.require_group(path) if path is not None else fp
    if dsetname not in target:
        target.create_dataset(dsetname, shape=bdata.shape, maxshape=(None,), dtype=bdata.dtype)
    elif bdata.size != target[dsetname].shape[0]:
        target[dsetname].resize((bdata.size,))
    target[dsetname][:] = bdata


def dump_state(state, fp, path: str | None = None, dsetname: str = "state", protocol: int = pickle.HIGHEST_PROTOCOL) -> None:
    """utils.py:760-770."""
    memfp = BytesIO()
    pickle.dump(state, memfp, protocol=protocol)
    dump_pickle_to_hdf(memfp, fp, path=path, dsetname=dsetname)


def load_state(fp, path: str = "checkpoint", dsetname: str = "state"):
    """The inverse of `dump_state` (samplers/base.py:236-247)."""
    data = fp[path][dsetname][...]
    return pickle.loads(np.asarray(data).tobytes())
