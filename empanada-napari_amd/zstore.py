"""zarr v2 directory stores without the zarr package.

The reference hands its dense outputs to zarr: ``zarr.open(store_url, mode='w')`` then
``create_array(name, shape=, dtype=, chunks=, overwrite=True)`` (empanada_napari/inference.py:100-103,
474-489; the multi-GPU flavour calls the v2 spelling ``create_dataset``, multigpu.py:202-205) and fills the
array chunk by chunk (empanada/zarr_utils.py:97-184); the widgets read volumes from such stores too.  zarr is
not part of this image, and the hot path needs only the on-disk FORMAT, which is small: this module reads and
writes the zarr v2 *directory store* layout

    <store>/.zgroup                      {"zarr_format": 2}
    <store>/<array>/.zarray              {"zarr_format": 2, "shape", "chunks", "dtype" (numpy typestr),
                                          "compressor": null, "fill_value": 0, "order": "C", "filters": null,
                                          "dimension_separator": "."}
    <store>/<array>/<i>.<j>.<k>          one raw C-order chunk of the full chunk shape (edge chunks padded)

so that any zarr v2 reader (zarr-python 2.x / 3.x, napari, dask) opens what the engine wrote, and the engine
opens uncompressed v2 stores written by them.  Compressed chunks are refused loudly (no codec library here).
When the real ``zarr`` package is importable, ``open_store`` returns zarr's own objects instead.

Only what the reference's path touches is implemented: groups, n-d arrays, basic indexing with integers and
unit-step slices (``array[z0:z1] = block``, ``array[...]``, ``array[i]``), ``shape / dtype / chunks / nchunks``.
"""
import itertools
import json
import math
import os
import shutil

import numpy as np

__all__ = ['open_store', 'DirGroup', 'DirArray']


def _write_json(path, obj):
    tmp = path + '.tmp'
    with open(tmp, 'w') as f:
        json.dump(obj, f, indent=4, sort_keys=True)
    os.replace(tmp, path)


class DirArray:
    """One zarr v2 array in a directory (uncompressed, C order)."""

    def __init__(self, path, mode='r'):
        self.path = path
        self.read_only = mode == 'r'
        meta_path = os.path.join(path, '.zarray')
        if not os.path.isfile(meta_path):
            raise FileNotFoundError(f'{path} is not a zarr v2 array (.zarray missing)')
        with open(meta_path) as f:
            m = json.load(f)
        if m.get('zarr_format') != 2:
            raise ValueError(f'{path}: zarr_format {m.get("zarr_format")} (only the v2 layout is written/read here)')
        if m.get('compressor') is not None or m.get('filters'):
            raise NotImplementedError(f'{path}: compressed / filtered chunks need the zarr package (compressor='
                                      f'{m.get("compressor")})')
        if m.get('order', 'C') != 'C':
            raise NotImplementedError(f'{path}: only C-order chunks')
        self.shape = tuple(int(s) for s in m['shape'])
        self.chunks = tuple(int(c) for c in m['chunks'])
        self.dtype = np.dtype(m['dtype'])
        fv = m.get('fill_value', 0)
        self.fill_value = 0 if fv is None else fv
        self.sep = m.get('dimension_separator', '.')
        self.ndim = len(self.shape)

    # ---- creation ----
    @classmethod
    def create(cls, path, shape, dtype, chunks, overwrite=False, fill_value=0):
        if os.path.exists(path):
            if not overwrite:
                raise FileExistsError(path)
            shutil.rmtree(path)
        os.makedirs(path)
        shape = tuple(int(s) for s in (shape if np.iterable(shape) else (shape,)))
        if chunks is None or chunks is True:
            chunks = tuple(min(s, 256) for s in shape)
        chunks = tuple(int(c) for c in (chunks if np.iterable(chunks) else (chunks,)))
        if len(chunks) != len(shape):
            raise ValueError(f'chunks {chunks} do not match shape {shape}')
        chunks = tuple(max(1, min(c, s)) if s > 0 else max(1, c) for c, s in zip(chunks, shape))
        dt = np.dtype(dtype)
        _write_json(os.path.join(path, '.zarray'), {
            'zarr_format': 2, 'shape': list(shape), 'chunks': list(chunks), 'dtype': dt.str, 'compressor': None,
            'fill_value': fill_value, 'order': 'C', 'filters': None, 'dimension_separator': '.'})
        return cls(path, mode='a')

    # ---- geometry ----
    @property
    def cdata_shape(self):
        return tuple(math.ceil(s / c) for s, c in zip(self.shape, self.chunks))

    @property
    def nchunks(self):
        return int(np.prod(self.cdata_shape)) if self.shape else 1

    @property
    def size(self):
        return int(np.prod(self.shape))

    def __len__(self):
        return self.shape[0]

    def _chunk_path(self, idx):
        return os.path.join(self.path, self.sep.join(str(i) for i in idx))

    def _read_chunk(self, idx):
        p = self._chunk_path(idx)
        if not os.path.isfile(p):
            return None
        a = np.fromfile(p, dtype=self.dtype)
        if a.size != int(np.prod(self.chunks)):
            raise ValueError(f'{p}: {a.size} items, expected a full chunk of {self.chunks}')
        return a.reshape(self.chunks)

    def _write_chunk(self, idx, block):
        p = self._chunk_path(idx)
        d = os.path.dirname(p)
        if d != self.path:
            os.makedirs(d, exist_ok=True)
        tmp = p + '.partial'
        np.ascontiguousarray(block, dtype=self.dtype).tofile(tmp)
        os.replace(tmp, p)

    def _normalise(self, key):
        """basic selection -> (per-axis (start, stop), axes dropped by integer indices)"""
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            i = [k is Ellipsis for k in key].index(True)
            fill = self.ndim - (len(key) - 1)
            key = key[:i] + (slice(None),) * fill + key[i + 1:]
        if len(key) > self.ndim:
            raise IndexError(f'too many indices for a {self.ndim}-d array')
        key = key + (slice(None),) * (self.ndim - len(key))
        sel, drop = [], []
        for ax, (k, n) in enumerate(zip(key, self.shape)):
            if isinstance(k, (int, np.integer)):
                k = int(k) + (n if k < 0 else 0)
                if not 0 <= k < n:
                    raise IndexError(f'index {k} out of bounds for axis {ax} with size {n}')
                sel.append((k, k + 1))
                drop.append(ax)
            elif isinstance(k, slice):
                a, b, st = k.indices(n)
                if st != 1:
                    raise NotImplementedError('only unit-step slices')
                sel.append((a, max(a, b)))
            else:
                raise NotImplementedError(f'only integers and slices index a DirArray (got {type(k).__name__})')
        return sel, tuple(drop)

    def _chunk_iter(self, sel):
        """chunks touched by ``sel``: (chunk index, slices inside the chunk, slices inside the selection)"""
        per_axis = []
        for (a, b), c in zip(sel, self.chunks):
            items = []
            for ci in range(a // c, (b - 1) // c + 1 if b > a else a // c):
                lo, hi = max(a, ci * c), min(b, (ci + 1) * c)
                items.append((ci, slice(lo - ci * c, hi - ci * c), slice(lo - a, hi - a)))
            per_axis.append(items)
        for combo in itertools.product(*per_axis):
            yield (tuple(x[0] for x in combo), tuple(x[1] for x in combo), tuple(x[2] for x in combo))

    # ---- numpy-style access ----
    def __getitem__(self, key):
        sel, drop = self._normalise(key)
        out = np.full([b - a for a, b in sel], self.fill_value, dtype=self.dtype)
        for idx, cs, os_ in self._chunk_iter(sel):
            blk = self._read_chunk(idx)
            if blk is not None:
                out[os_] = blk[cs]
        return out.reshape([n for ax, n in enumerate(out.shape) if ax not in drop]) if drop else out

    def __setitem__(self, key, value):
        if self.read_only:
            raise PermissionError(f'{self.path} was opened read-only')
        sel, drop = self._normalise(key)
        shp = tuple(b - a for a, b in sel)
        value = np.asarray(value)
        if drop and value.ndim == len(shp) - len(drop):
            value = value.reshape([1 if ax in drop else n for ax, n in enumerate(shp)])
        value = np.broadcast_to(value, shp)
        for idx, cs, os_ in self._chunk_iter(sel):
            full = all(s.start == 0 and s.stop == c for s, c in zip(cs, self.chunks))
            if full:
                self._write_chunk(idx, value[os_])
                continue
            blk = self._read_chunk(idx)
            if blk is None:
                blk = np.full(self.chunks, self.fill_value, dtype=self.dtype)
            else:
                blk = blk.copy()
            blk[cs] = value[os_]
            self._write_chunk(idx, blk)

    def __array__(self, dtype=None, copy=None):
        a = self[...]
        return a if dtype is None else a.astype(dtype)

    def __repr__(self):
        return f'<DirArray {self.path} {self.shape} {self.dtype} chunks={self.chunks}>'


class DirGroup:
    """A zarr v2 group in a directory: ``create_array`` (zarr-python 3 spelling, inference.py:100) and
    ``create_dataset`` (zarr-python 2 spelling, multigpu.py:202) both create a ``DirArray``."""

    def __init__(self, path, mode='a'):
        self.path = path
        self.mode = mode
        if mode == 'w':
            if os.path.isdir(path):
                # zarr.open(mode='w') deletes what is there -- only ever a store (has .zgroup/.zarray or is empty)
                entries = os.listdir(path)
                if entries and not any(e in ('.zgroup', '.zarray', '.zattrs') for e in entries):
                    raise FileExistsError(f'{path} exists and is not a zarr store; refusing to overwrite it')
                shutil.rmtree(path)
            os.makedirs(path)
            _write_json(os.path.join(path, '.zgroup'), {'zarr_format': 2})
        elif mode == 'a':
            os.makedirs(path, exist_ok=True)
            if not os.path.isfile(os.path.join(path, '.zgroup')):
                _write_json(os.path.join(path, '.zgroup'), {'zarr_format': 2})
        elif not os.path.isfile(os.path.join(path, '.zgroup')):
            raise FileNotFoundError(f'{path} is not a zarr v2 group (.zgroup missing)')

    def create_array(self, name, shape, dtype, chunks=None, overwrite=False, fill_value=0, **ignored):
        if self.mode == 'r':
            raise PermissionError(f'{self.path} was opened read-only')
        return DirArray.create(os.path.join(self.path, name), shape, dtype, chunks, overwrite=overwrite,
                               fill_value=fill_value)

    create_dataset = create_array

    def create_group(self, name, overwrite=False):
        p = os.path.join(self.path, name)
        if overwrite and os.path.isdir(p):
            shutil.rmtree(p)
        return DirGroup(p, mode='a')

    def __contains__(self, name):
        p = os.path.join(self.path, name)
        return os.path.isfile(os.path.join(p, '.zarray')) or os.path.isfile(os.path.join(p, '.zgroup'))

    def __getitem__(self, name):
        p = os.path.join(self.path, name)
        if os.path.isfile(os.path.join(p, '.zarray')):
            return DirArray(p, mode='r' if self.mode == 'r' else 'a')
        if os.path.isfile(os.path.join(p, '.zgroup')):
            return DirGroup(p, mode='r' if self.mode == 'r' else 'a')
        raise KeyError(name)

    def array_keys(self):
        return sorted(n for n in os.listdir(self.path) if os.path.isfile(os.path.join(self.path, n, '.zarray')))

    keys = array_keys

    def __repr__(self):
        return f'<DirGroup {self.path} arrays={self.array_keys()}>'


def open_store(store_url, mode=None):
    """``zarr.open(store_url[, mode])`` of the reference (inference.py:58,113,404,464): the zarr package when it is
    installed, otherwise the directory-store reader / writer above.  ``mode``: 'w' create (delete what is there),
    'a' read / write (create if missing), 'r' read only; None = zarr's default 'a'."""
    try:
        import zarr
    except ImportError:
        zarr = None
    if zarr is not None:
        return zarr.open(store_url, mode=mode) if mode else zarr.open(store_url)
    url = str(store_url)
    if '://' in url and not url.startswith('file://'):
        raise NotImplementedError(f'{url}: only local directory stores without the zarr package')
    path = url[len('file://'):] if url.startswith('file://') else url
    mode = mode or 'a'
    if mode in ('r', 'r+', 'a') and os.path.isfile(os.path.join(path, '.zarray')):
        return DirArray(path, mode='r' if mode == 'r' else 'a')
    if mode == 'r+':
        mode = 'a' if os.path.isdir(path) else 'r'
    return DirGroup(path, mode=mode)
