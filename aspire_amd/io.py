"""Reading and writing the reference's HDF5 layout through the h5py *group protocol*.

The layout (what a file must look like so that the reference, or a user of it, can open it) is:

* sampler checkpoints: ONE byte dataset `/checkpoint/state` (`docs/checkpointing.rst:17-26`) holding the pickle of the state
  dictionary, one byte per element (numpy `S1`), created with an unlimited first axis so that the next checkpoint can resize
  it and overwrite it in place (`src/aspire/utils.py:733-770`);
* dictionaries (`Samples.save`, `SMCHistory.save`, the config groups): one group per dictionary; every LEAF becomes one dataset
  whose name is the dot-joined key path (`a.b.c`), nested dictionaries contribute only to the names
  (`src/aspire/utils.py:841-887`);
* leaves (`src/aspire/utils.py:652-730`): arrays as they are (device tensors via the host), numbers and strings as scalars,
  a list of strings as a variable-length UTF-8 string array, `None` and `{}` as the two marker strings below.

Only the protocol is used (`require_group`, `create_dataset`, `in`, `[]`, `items`, dataset `resize` / `shape` / slicing), so a
real `h5py.File` and the in-memory stand-in of the tests both work; h5py itself is not part of the build image.
"""
from __future__ import annotations

import pickle
from pathlib import Path
from typing import Any, Iterator

import numpy as np

NONE_TOKEN = "__none__"
EMPTY_DICT_TOKEN = "__empty_dict__"
_TOKENS = {NONE_TOKEN: lambda: None, EMPTY_DICT_TOKEN: dict}


def h5py_available() -> bool:
    try:
        import h5py  # noqa: F401
    except ImportError:
        return False
    return True


def open_h5(path, mode: str = "r"):
    """`h5py.File(path, mode)`, stamped with `aspire_version` when it is writable (the reference's `AspireFile`)."""
    try:
        import h5py
    except ImportError as exc:  # pragma: no cover - h5py is absent from the build image
        raise RuntimeError("HDF5 files need h5py, which is not installed; use a .pkl checkpoint path or pass an open "
                           "h5py-compatible group") from exc
    handle = h5py.File(path, mode)
    if handle.mode != "r":
        from . import __version__

        handle.attrs["aspire_version"] = __version__
    return handle


# ---- leaves ----------------------------------------------------------------------------------------------------------
def _string_array(values) -> np.ndarray:
    """Strings in a form h5py stores: its variable-length UTF-8 dtype when h5py is importable (an object array WITHOUT that
    dtype's metadata is refused by h5py), fixed-width UTF-8 bytes otherwise; both decode to the same list of str."""
    values = list(values)
    try:
        import h5py
    except ImportError:
        return np.array([s.encode("utf-8") for s in values], dtype="S")
    return np.array(values, dtype=h5py.string_dtype(encoding="utf-8"))


def _is_tensor(value) -> bool:
    return type(value).__module__.split(".")[0] == "torch" and hasattr(value, "detach")


def encode_for_hdf5(value: Any) -> Any:
    """One Python value -> what `create_dataset(data=...)` receives (the leaf rules in the module docstring)."""
    if value is None:
        return NONE_TOKEN
    if _is_tensor(value):
        return value.detach().cpu().numpy()
    if isinstance(value, Path):
        return str(value)
    if isinstance(value, dict):
        return {key: encode_for_hdf5(item) for key, item in value.items()} if value else EMPTY_DICT_TOKEN
    if isinstance(value, set):  # (utils.py:679-680: element-wise; no HDF5 type takes a set, so the writer's str() fall-back stores it)
        return {encode_for_hdf5(item) for item in value}
    if isinstance(value, (list, tuple)):
        if value and all(isinstance(item, str) for item in value):
            return _string_array(value)
        if not value:  # an empty sequence: the reference's all() is vacuously true there - an empty string array
            return _string_array(())
        return [encode_for_hdf5(item) for item in value]
    return value  # arrays, numbers, strings: stored as they are


def _decode_text(text: str) -> Any:
    make = _TOKENS.get(text)
    return make() if make is not None else text


def decode_from_hdf5(value: Any) -> Any:
    """What a dataset read returns -> the Python value that was saved."""
    if isinstance(value, bytes):
        return _decode_text(value.decode("utf-8"))
    if isinstance(value, str):
        return _decode_text(value)
    if isinstance(value, np.ndarray):
        if value.ndim == 0:
            return decode_from_hdf5(value.item())
        if value.dtype.kind in "SOU":
            try:
                return value.astype(str).tolist()
            except (TypeError, ValueError, UnicodeError):
                return value
        return value
    if isinstance(value, dict):
        return {(key.decode("utf-8") if isinstance(key, bytes) else key): decode_from_hdf5(item) for key, item in value.items()}
    if isinstance(value, (list, tuple)):
        return type(value)(decode_from_hdf5(item) for item in value)
    if isinstance(value, set):
        return {decode_from_hdf5(item) for item in value}
    return value


# ---- dictionaries <-> groups of dotted datasets ----------------------------------------------------------------------------
def _leaves(tree: dict) -> Iterator[tuple[str, Any]]:
    """(dotted name, leaf) for every non-dictionary value of a nested dictionary, depth first in insertion order."""
    pending = [((), iter(tree.items()))]
    while pending:
        trail, it = pending[-1]
        for key, item in it:
            if isinstance(item, dict):
                pending.append((trail + (str(key),), iter(item.items())))
                break
            yield ".".join(trail + (str(key),)), item
        else:
            pending.pop()


def recursively_save_to_h5_file(h5_file, path: str, dictionary: dict) -> None:
    """Write `dictionary` below the group `path`: one dataset per leaf, named by its dotted key path.  A leaf the backend
    cannot store natively is stored as its `str()` (the reference's fall-back); a second failure is an error."""
    group = h5_file.require_group(path)
    for name, leaf in _leaves(dictionary):
        try:
            group.create_dataset(name, data=encode_for_hdf5(leaf))
        except (TypeError, ValueError) as first:
            try:
                group.create_dataset(name, data=_string_array([str(leaf)])[0])
            except Exception:
                raise RuntimeError(f"Cannot save key {name} with value {leaf!r} to HDF5 file.") from first


def load_from_h5_file(h5_file, path: str) -> dict:
    """Read a group written by `recursively_save_to_h5_file` back into the nested dictionary."""
    tree: dict = {}
    for name, dataset in h5_file[path].items():
        *branch, last = name.split(".")
        node = tree
        for part in branch:
            node = node.setdefault(part, {})
        node[last] = decode_from_hdf5(dataset[()])
    return tree


# ---- the checkpoint blob ---------------------------------------------------------------------------------------------------
def _write_byte_dataset(where, name: str, payload: bytes) -> None:
    """`payload` as a 1-D `S1` dataset `name`: created resizable on first use, resized when the length changed, then
    overwritten whole (the file keeps ONE checkpoint)."""
    blob = np.frombuffer(payload, dtype="S1")
    if name in where:
        dset = where[name]
        if dset.shape[0] != blob.size:
            dset.resize((blob.size,))
    else:
        dset = where.create_dataset(name, shape=blob.shape, maxshape=(None,), dtype=blob.dtype)
    dset[:] = blob


def dump_pickle_to_hdf(memfp, fp, path: str | None = None, dsetname: str = "state") -> None:
    """The bytes of a pickle stream -> `<path>/<dsetname>`.  `memfp`: a `BytesIO` (`getvalue()`), or - as the reference
    accepts (`utils.py:733-757`: `seek(0)`, `read()`) - any seekable binary stream."""
    if hasattr(memfp, "getvalue"):
        payload = memfp.getvalue()
    else:
        memfp.seek(0)
        payload = memfp.read()
    _write_byte_dataset(fp if path is None else fp.require_group(path), dsetname, payload)


def dump_state(state, fp, path: str | None = None, dsetname: str = "state", protocol: int = pickle.HIGHEST_PROTOCOL) -> None:
    """Pickle `state` into `<path>/<dsetname>` (`/checkpoint/state` for the samplers)."""
    _write_byte_dataset(fp if path is None else fp.require_group(path), dsetname, pickle.dumps(state, protocol=protocol))


def load_state(fp, path: str = "checkpoint", dsetname: str = "state"):
    """The state dictionary `dump_state` wrote."""
    return pickle.loads(np.asarray(fp[path][dsetname][...]).tobytes())
