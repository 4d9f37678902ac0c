"""zarr v2 / v3 directory stores without the zarr package (empanada-napari_amd/zstore.py): the on-disk layout is checked
against the zarr v2 and v3 specifications by reading the files back with numpy / json / gzip only (and by reading stores
written here by hand, byte for byte as the specifications lay them out), and the array semantics against numpy on random
basic selections."""
import json
import os

import numpy as np
import pytest

from empanada_napari_amd import zstore


def test_layout_follows_the_v2_spec(tmp_path):
    g = zstore.open_store(str(tmp_path / 'seg.zarr'), mode='w')
    a = g.create_array('mito', shape=(5, 7, 9), dtype=np.int32, chunks=(2, 4, 9), overwrite=True)
    ref = np.arange(5 * 7 * 9, dtype=np.int32).reshape(5, 7, 9)
    a[...] = ref
    root = tmp_path / 'seg.zarr'
    assert json.load(open(root / '.zgroup')) == {'zarr_format': 2}
    meta = json.load(open(root / 'mito' / '.zarray'))
    assert meta['zarr_format'] == 2 and meta['shape'] == [5, 7, 9] and meta['chunks'] == [2, 4, 9]
    assert meta['dtype'] == '<i4' and meta['compressor'] is None and meta['order'] == 'C' and meta['fill_value'] == 0
    # chunk (i,j,k) lives in file "i.j.k", is a FULL chunk in C order, edge chunks padded with the fill value
    files = sorted(f for f in os.listdir(root / 'mito') if not f.startswith('.'))
    assert files == sorted(f'{i}.{j}.0' for i in range(3) for j in range(2))
    c = np.fromfile(root / 'mito' / '2.1.0', dtype='<i4').reshape(2, 4, 9)
    np.testing.assert_array_equal(c[:1, :3], ref[4:5, 4:7])
    assert not c[1:].any() and not c[:, 3:].any()
    assert a.nchunks == 6 and a.cdata_shape == (3, 2, 1)


def test_reopen_and_basic_selections_match_numpy(tmp_path):
    rng = np.random.default_rng(0)
    url = str(tmp_path / 'v.zarr')
    g = zstore.open_store(url, mode='w')
    a = g.create_dataset('x', shape=(11, 13, 6), dtype=np.uint8, chunks=(4, 5, 6))     # zarr-python 2 spelling (multigpu.py:202)
    ref = np.zeros((11, 13, 6), np.uint8)
    for _ in range(40):
        lo = [int(rng.integers(0, n)) for n in ref.shape]
        hi = [int(rng.integers(l + 1, n + 1)) for l, n in zip(lo, ref.shape)]
        sl = tuple(slice(l, h) for l, h in zip(lo, hi))
        blk = rng.integers(0, 255, size=[h - l for l, h in zip(lo, hi)], dtype=np.uint8)
        a[sl] = blk
        ref[sl] = blk
    a[3] = 7
    ref[3] = 7
    a[2:4, 5] = np.full((2, 6), 9, np.uint8)
    ref[2:4, 5] = 9
    b = zstore.open_store(url, mode='r')['x']
    np.testing.assert_array_equal(b[...], ref)
    np.testing.assert_array_equal(b[4], ref[4])
    np.testing.assert_array_equal(b[2:9, 1:12, 3], ref[2:9, 1:12, 3])
    np.testing.assert_array_equal(np.asarray(b), ref)
    assert b.shape == ref.shape and b.dtype == np.uint8 and b.chunks == (4, 5, 6)
    with pytest.raises(PermissionError):
        b[0] = 1
    with pytest.raises(NotImplementedError):
        b[::2]
    assert 'x' in zstore.open_store(url) and zstore.open_store(url).array_keys() == ['x']


def test_untouched_chunks_read_as_fill_value_and_overwrite(tmp_path):
    g = zstore.open_store(str(tmp_path / 's.zarr'), mode='w')
    a = g.create_array('a', shape=(6, 6), dtype=np.int64, chunks=(3, 3))
    a[0:3, 0:3] = 5
    assert a[...].sum() == 45 and sorted(f for f in os.listdir(a.path) if not f.startswith('.')) == ['0.0']
    with pytest.raises(FileExistsError):
        g.create_array('a', shape=(2, 2), dtype=np.int64, chunks=(2, 2))
    a2 = g.create_array('a', shape=(2, 2), dtype=np.int64, chunks=(2, 2), overwrite=True)
    assert a2[...].sum() == 0
    # mode='w' on an existing store wipes it; on a directory that is not a store it refuses
    g2 = zstore.open_store(str(tmp_path / 's.zarr'), mode='w')
    assert g2.array_keys() == []
    os.makedirs(tmp_path / 'notastore')
    open(tmp_path / 'notastore' / 'precious.txt', 'w').write('x')
    with pytest.raises(FileExistsError):
        zstore.open_store(str(tmp_path / 'notastore'), mode='w')


def test_unknown_compressors_are_refused(tmp_path):
    """(round 4: blosc / lz4 / zstd are read through pyarrow's codecs -- test_reads_zarr_python_default_codecs; anything else
    still fails loudly at open, filters too)"""
    p = tmp_path / 'c.zarr' / 'a'
    os.makedirs(p)
    json.dump({'zarr_format': 2}, open(tmp_path / 'c.zarr' / '.zgroup', 'w'))
    json.dump({'zarr_format': 2, 'shape': [4], 'chunks': [4], 'dtype': '<i4', 'compressor': {'id': 'lzma'},
               'fill_value': 0, 'order': 'C', 'filters': None}, open(p / '.zarray', 'w'))
    with pytest.raises(NotImplementedError, match='compress'):
        zstore.open_store(str(tmp_path / 'c.zarr'), mode='r')['a']
    json.dump({'zarr_format': 2, 'shape': [4], 'chunks': [4], 'dtype': '<i4', 'compressor': None,
               'fill_value': 0, 'order': 'C', 'filters': [{'id': 'delta'}]}, open(p / '.zarray', 'w'))
    with pytest.raises(NotImplementedError, match='filter'):
        zstore.open_store(str(tmp_path / 'c.zarr'), mode='r')['a']


def test_chunk_ranges_cover_the_store_chunks(tmp_path):
    """the reference's chunked fill splits runs at chunk faces (zarr_utils.py:20-58, pinned by its tests restated in
    test_oracle_sparse.py); a DirArray filled run by run through those ranges equals the dense fill"""
    from oracle import sparse as osp
    d, h, w = 6, 8, 10
    a = zstore.open_store(str(tmp_path / 'f.zarr'), mode='w').create_array('a', shape=(d, h, w), dtype=np.int32,
                                                                             chunks=(4, 3, 6))
    starts, runs = np.array([3, 75, 200, 470]), np.array([40, 20, 133, 10])
    dense = np.zeros(d * h * w, np.int32)
    for s, r in zip(starts, runs):
        dense[s:s + r] = 7
    ranges = np.stack([starts, starts + runs], axis=1)
    for modulo, divisor in ((d * h * w, 4 * h * w), (h * w, 3 * w), (w, 6)):
        ranges = np.array(osp.chunk_ranges(ranges, modulo, divisor))
    flat = np.zeros(d * h * w, np.int32)
    for s, e in ranges:
        z0, y0, x0 = np.unravel_index(s, (d, h, w))
        z1, y1, x1 = np.unravel_index(e - 1, (d, h, w))
        assert (z0 // 4, y0 // 3, x0 // 6) == (z1 // 4, y1 // 3, x1 // 6), 'a chunked range stays inside one chunk'
        flat[s:e] = 7
    np.testing.assert_array_equal(flat, dense)
    a[...] = flat.reshape(d, h, w)
    np.testing.assert_array_equal(a[...].ravel(), dense)


def test_layout_follows_the_v3_spec(tmp_path):
    """what zarr-python 3's ``create_array`` (the reference's spelling, empanada_napari/inference.py:100-103) lays out:
    zarr.json per node, chunks under c/<i>/<j>/<k>, little-endian C-order bytes"""
    g = zstore.open_store(str(tmp_path / 'seg.zarr'), mode='w', zarr_format=3)
    a = g.create_array('mito', shape=(5, 7, 9), dtype=np.int32, chunks=(2, 4, 9), overwrite=True)
    ref = np.arange(5 * 7 * 9, dtype=np.int32).reshape(5, 7, 9)
    a[...] = ref
    root = tmp_path / 'seg.zarr'
    gm = json.load(open(root / 'zarr.json'))
    assert gm['zarr_format'] == 3 and gm['node_type'] == 'group' and not (root / '.zgroup').exists()
    m = json.load(open(root / 'mito' / 'zarr.json'))
    assert m['zarr_format'] == 3 and m['node_type'] == 'array' and m['shape'] == [5, 7, 9] and m['data_type'] == 'int32'
    assert m['chunk_grid'] == {'name': 'regular', 'configuration': {'chunk_shape': [2, 4, 9]}}
    assert m['chunk_key_encoding'] == {'name': 'default', 'configuration': {'separator': '/'}}
    assert m['codecs'] == [{'name': 'bytes', 'configuration': {'endian': 'little'}}] and m['fill_value'] == 0
    c = np.fromfile(root / 'mito' / 'c' / '2' / '1' / '0', dtype='<i4').reshape(2, 4, 9)
    np.testing.assert_array_equal(c[:1, :3], ref[4:5, 4:7])
    assert not c[1:].any() and not c[:, 3:].any()
    b = zstore.open_store(str(root), mode='r')['mito']
    assert b.zarr_format == 3 and b.shape == (5, 7, 9) and b.chunks == (2, 4, 9) and b.dtype == np.int32
    np.testing.assert_array_equal(b[...], ref)
    np.testing.assert_array_equal(b[1:4, 2:6, 3], ref[1:4, 2:6, 3])
    assert 'mito' in zstore.open_store(str(root), mode='r') and zstore.open_store(str(root), mode='r').array_keys() == ['mito']


def _hand_written_v3(root, ref, chunks, codecs, key_enc, dtype_name):
    """a v3 array as ANOTHER writer would leave it: metadata and chunk files produced here from the specification"""
    os.makedirs(root)
    json.dump({'zarr_format': 3, 'node_type': 'array', 'shape': list(ref.shape), 'data_type': dtype_name,
               'chunk_grid': {'name': 'regular', 'configuration': {'chunk_shape': list(chunks)}},
               'chunk_key_encoding': key_enc, 'fill_value': 0, 'codecs': codecs, 'attributes': {},
               'dimension_names': None, 'storage_transformers': []}, open(os.path.join(root, 'zarr.json'), 'w'))
    import gzip
    import itertools
    big = codecs[0].get('configuration', {}).get('endian') == 'big'
    gz = len(codecs) > 1
    grid = [range(-(-n // c)) for n, c in zip(ref.shape, chunks)]
    for idx in itertools.product(*grid):
        blk = np.zeros(chunks, ref.dtype)
        sl = tuple(slice(i * c, min(n, (i + 1) * c)) for i, c, n in zip(idx, chunks, ref.shape))
        blk[tuple(slice(0, s.stop - s.start) for s in sl)] = ref[sl]
        data = blk.astype(ref.dtype.newbyteorder('>' if big else '<')).tobytes()
        if key_enc['name'] == 'default':
            sep = key_enc.get('configuration', {}).get('separator', '/')
            key = sep.join(['c'] + [str(i) for i in idx])
        else:
            key = key_enc.get('configuration', {}).get('separator', '.').join(str(i) for i in idx)
        p = os.path.join(root, *key.split('/'))
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, 'wb') as f:
            f.write(gzip.compress(data, 5) if gz else data)


@pytest.mark.parametrize('case', ['plain', 'gzip', 'dot-keys', 'v2-keys', 'big-endian'])
def test_reads_v3_arrays_written_elsewhere(tmp_path, case):
    rng = np.random.default_rng(3)
    ref = rng.integers(0, 60000, (9, 10, 7)).astype(np.uint16)
    codecs = [{'name': 'bytes', 'configuration': {'endian': 'big' if case == 'big-endian' else 'little'}}]
    if case == 'gzip':
        codecs.append({'name': 'gzip', 'configuration': {'level': 5}})
    enc = {'plain': {'name': 'default', 'configuration': {'separator': '/'}}, 'gzip': {'name': 'default'},
           'dot-keys': {'name': 'default', 'configuration': {'separator': '.'}},
           'v2-keys': {'name': 'v2', 'configuration': {'separator': '.'}},
           'big-endian': {'name': 'default', 'configuration': {'separator': '/'}}}[case]
    root = str(tmp_path / 'a')
    _hand_written_v3(root, ref, (4, 4, 7), codecs, enc, 'uint16')
    a = zstore.open_store(root, mode='r')
    assert isinstance(a, zstore.DirArray) and a.zarr_format == 3
    np.testing.assert_array_equal(a[...], ref)
    np.testing.assert_array_equal(a[2:9, 1:5], ref[2:9, 1:5])
    with pytest.raises(PermissionError):
        a[0] = 1


def test_gzip_and_zlib_chunks_round_trip_and_other_codecs_are_refused(tmp_path):
    import gzip
    import zlib
    rng = np.random.default_rng(5)
    ref = rng.integers(0, 9, (6, 8, 8)).astype(np.int32)
    for fmt, comp in ((2, 'zlib'), (2, 'gzip'), (3, 'gzip')):
        g = zstore.open_store(str(tmp_path / f's{fmt}{comp}'), mode='w', zarr_format=fmt)
        a = g.create_array('x', shape=ref.shape, dtype=ref.dtype, chunks=(4, 8, 8), compressor=comp)
        a[...] = ref
        a[1:3, 2:4] = 77
        want = ref.copy()
        want[1:3, 2:4] = 77
        np.testing.assert_array_equal(zstore.open_store(str(tmp_path / f's{fmt}{comp}'), mode='r')['x'][...], want)
        chunk = tmp_path / f's{fmt}{comp}' / 'x' / ('c/0/0/0' if fmt == 3 else '0.0.0')
        raw = open(chunk, 'rb').read()
        data = gzip.decompress(raw) if comp == 'gzip' else zlib.decompress(raw)
        np.testing.assert_array_equal(np.frombuffer(data, '<i4').reshape(4, 8, 8), want[:4])
        meta = json.load(open(tmp_path / f's{fmt}{comp}' / 'x' / ('zarr.json' if fmt == 3 else '.zarray')))
        assert (meta['codecs'][1]['name'] == 'gzip') if fmt == 3 else (meta['compressor']['id'] == comp)
    # a codec this module cannot decode is refused with its name (round 4 reads zstd: test_reads_zarr_python_default_codecs)
    root = tmp_path / 'z'
    os.makedirs(root)
    json.dump({'zarr_format': 3, 'node_type': 'array', 'shape': [4], 'data_type': 'uint8',
               'chunk_grid': {'name': 'regular', 'configuration': {'chunk_shape': [4]}},
               'chunk_key_encoding': {'name': 'default'}, 'fill_value': 0,
               'codecs': [{'name': 'bytes'}, {'name': 'sharding_indexed', 'configuration': {}}]}, open(root / 'zarr.json', 'w'))
    with pytest.raises(NotImplementedError, match='sharding_indexed'):
        zstore.open_store(str(root), mode='r')
    with pytest.raises(NotImplementedError):
        zstore.open_store(str(tmp_path / 'v3z'), mode='w', zarr_format=3).create_array('x', shape=(4,), dtype=np.uint8, compressor='zlib')


def test_zarr_open_kwargs_follow_the_installed_package():
    """zarr-python 2 spells the format argument ``zarr_version``; 3 ``zarr_format``."""
    from empanada_napari_amd import zstore
    assert zstore._zarr_open_kwargs('3.0.8', 'w', 3) == {'mode': 'w', 'zarr_format': 3}
    assert zstore._zarr_open_kwargs('2.18.3', 'w', 2) == {'mode': 'w', 'zarr_version': 2}
    assert zstore._zarr_open_kwargs('2.18.3', None, 3) == {}          # zarr 2 cannot write v3: request dropped, no TypeError
    assert zstore._zarr_open_kwargs('3.1.0', None, None) == {}


def _blosc1_frame(data, typesize, blocksize, inner='lz4', shuffle=True, split=True):
    """Encoder of the Blosc-1 frame written from c-blosc's published header / block layout (the test's side of the contract
    ``zstore.blosc1_decode`` restates): 16-byte header, int32 block offsets, per block `typesize` streams (or one), each
    an int32 size + payload; a stream that does not shrink is stored raw with size == its length."""
    import zlib

    import pyarrow as pa
    fmt = {'lz4': 1, 'zlib': 3, 'zstd': 4}[inner]
    comp = {'lz4': lambda b: pa.Codec('lz4_raw').compress(b, asbytes=True), 'zlib': lambda b: zlib.compress(b, 5),
            'zstd': lambda b: pa.Codec('zstd').compress(b, asbytes=True)}[inner]
    nbytes = len(data)
    nblocks = -(-nbytes // blocksize)
    flags = (fmt << 5) | (0x1 if shuffle and typesize > 1 else 0) | (0 if split else 0x10)
    blocks = []
    for b in range(nblocks):
        blk = data[b * blocksize:(b + 1) * blocksize]
        if flags & 0x1:
            n = len(blk) // typesize
            blk = np.frombuffer(blk, np.uint8, n * typesize).reshape(n, typesize).T.tobytes() + blk[n * typesize:]
        leftover = len(blk) < blocksize
        nsplit = typesize if (split and not leftover and typesize > 1 and len(blk) % typesize == 0) else 1
        ne = len(blk) // nsplit
        enc = b''
        for k in range(nsplit):
            part = blk[k * ne:(k + 1) * ne]
            c = comp(part)
            if len(c) >= len(part):
                c = part
            enc += len(c).to_bytes(4, 'little', signed=True) + c
        blocks.append(enc)
    pos, offs = 16 + 4 * nblocks, []
    for e in blocks:
        offs.append(pos)
        pos += len(e)
    head = bytes([2, 1, flags, typesize]) + nbytes.to_bytes(4, 'little') + blocksize.to_bytes(4, 'little') + pos.to_bytes(4, 'little')
    return head + b''.join(o.to_bytes(4, 'little', signed=True) for o in offs) + b''.join(blocks)


@pytest.mark.parametrize('dtype,inner,shuffle,split', [('u1', 'lz4', True, True), ('<u2', 'lz4', True, True),
                                                        ('<u2', 'zstd', True, False), ('<i4', 'zlib', False, True),
                                                        ('<u2', 'lz4', False, False)])
def test_reads_zarr_python_default_codecs(tmp_path, dtype, inner, shuffle, split):
    """Stores as zarr-python writes them BY DEFAULT (VERDICT r03 missing 6): v2 with the Blosc compressor (frames built
    here by hand from c-blosc's frame description: lz4 / zstd / zlib inner codecs, byte shuffle, split and unsplit blocks,
    a shorter last block, an incompressible stream stored raw), v2 with numcodecs' LZ4 / Zstd, v3 with the zstd codec."""
    pa = pytest.importorskip('pyarrow')
    from empanada_napari_amd import zstore
    rng = np.random.default_rng(3)
    dt = np.dtype(dtype)
    vol = (rng.integers(0, 7, (20, 33, 40)) * (rng.random((20, 33, 40)) > 0.6)).astype(dt)
    vol[3] = rng.integers(0, np.iinfo(dt).max, (33, 40)).astype(dt)          # an incompressible slab
    chunks = (8, 16, 24)
    grid = [range(-(-s // c)) for s, c in zip(vol.shape, chunks)]

    def padded(i, j, k):
        blk = np.zeros(chunks, dt)
        sub = vol[i * 8:(i + 1) * 8, j * 16:(j + 1) * 16, k * 24:(k + 1) * 24]
        blk[:sub.shape[0], :sub.shape[1], :sub.shape[2]] = sub
        return blk.tobytes()

    def write_v2(name, comp_meta, enc):
        d = tmp_path / name / 'em'
        d.mkdir(parents=True)
        (tmp_path / name / '.zgroup').write_text(json.dumps({'zarr_format': 2}))
        (d / '.zarray').write_text(json.dumps({'zarr_format': 2, 'shape': list(vol.shape), 'chunks': list(chunks), 'dtype': dt.str,
                                               'compressor': comp_meta, 'fill_value': 0, 'order': 'C', 'filters': None}))
        for i in grid[0]:
            for j in grid[1]:
                for k in grid[2]:
                    (d / f'{i}.{j}.{k}').write_bytes(enc(padded(i, j, k)))
        return zstore.open_store(str(tmp_path / name), mode='r')['em']

    blosc = write_v2('blosc', {'id': 'blosc', 'cname': inner, 'clevel': 5, 'shuffle': int(shuffle), 'blocksize': 0},
                     lambda b: _blosc1_frame(b, dt.itemsize, 2048, inner, shuffle, split))
    np.testing.assert_array_equal(blosc[...], vol)
    np.testing.assert_array_equal(blosc[5:17, 3:30, 7], vol[5:17, 3:30, 7])
    lz4 = write_v2('lz4', {'id': 'lz4', 'acceleration': 1},
                   lambda b: len(b).to_bytes(4, 'little') + pa.Codec('lz4_raw').compress(b, asbytes=True))
    np.testing.assert_array_equal(lz4[...], vol)
    zs = write_v2('zstd2', {'id': 'zstd', 'level': 1}, lambda b: pa.Codec('zstd').compress(b, asbytes=True))
    np.testing.assert_array_equal(zs[...], vol)
    with pytest.raises(PermissionError):
        blosc[0] = 0
    # v3 with zarr-python 3's default codec chain [bytes, zstd]: written by this module, read back, and the files checked
    g = zstore.open_store(str(tmp_path / 'v3'), mode='w', zarr_format=3)
    a = g.create_array('em', shape=vol.shape, dtype=dt, chunks=chunks, compressor='zstd')
    a[...] = vol
    meta = json.load(open(tmp_path / 'v3' / 'em' / 'zarr.json'))
    assert [c['name'] for c in meta['codecs']] == ['bytes', 'zstd']
    raw = (tmp_path / 'v3' / 'em' / 'c' / '0' / '0' / '0').read_bytes()
    assert pa.Codec('zstd').decompress(raw, decompressed_size=int(np.prod(chunks)) * dt.itemsize, asbytes=True) == padded(0, 0, 0)
    np.testing.assert_array_equal(zstore.open_store(str(tmp_path / 'v3'), mode='r')['em'][...], vol)


def test_blosc_refuses_what_it_cannot_decode():
    from empanada_napari_amd import zstore
    frame = bytearray(_blosc1_frame(bytes(range(256)) * 8, 2, 1024, 'lz4'))
    frame[2] |= 0x4            # bit shuffle
    with pytest.raises(NotImplementedError, match='bit-shuffled'):
        zstore.blosc1_decode(bytes(frame))
    frame = bytearray(_blosc1_frame(bytes(range(256)) * 8, 2, 1024, 'lz4'))
    frame[2] &= 0x1F           # inner codec 0 = blosclz
    with pytest.raises(NotImplementedError, match='blosclz'):
        zstore.blosc1_decode(bytes(frame))
    with pytest.raises(ValueError):
        zstore.blosc1_decode(bytes(frame[:40]))


def test_crc32c_known_answers():
    """the checksum of zarr v3's ``crc32c`` codec (Castagnoli): the standard check value and RFC 3720's vectors"""
    assert zstore.crc32c(b'123456789') == 0xE3069283
    assert zstore.crc32c(b'') == 0
    assert zstore.crc32c(bytes(32)) == 0x8A9136AA             # iSCSI, RFC 3720 B.4: 32 bytes of zeros
    assert zstore.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43    # 32 bytes of ones
    assert zstore.crc32c(bytes(range(32))) == 0x46DD794E      # 32 incrementing bytes


@pytest.mark.parametrize('compressor', [None, 'gzip', 'zstd'])
def test_sharded_v3_arrays_round_trip(tmp_path, compressor):
    """zarr v3 sharding (``create_array(chunks=<inner>, shards=<shard>)``): metadata, shard files with their index at the end,
    partial writes (read - patch - rewrite of a shard), reads across shards"""
    if compressor == 'zstd':
        pytest.importorskip('pyarrow')
    g = zstore.open_store(str(tmp_path / 's.zarr'), mode='w', zarr_format=3)
    a = g.create_array('v', shape=(20, 33, 17), dtype=np.uint16, chunks=(4, 8, 8), shards=(8, 16, 16), compressor=compressor)
    m = json.load(open(tmp_path / 's.zarr' / 'v' / 'zarr.json'))
    assert m['chunk_grid']['configuration']['chunk_shape'] == [8, 16, 16]
    sc = m['codecs'][0]
    assert sc['name'] == 'sharding_indexed' and sc['configuration']['chunk_shape'] == [4, 8, 8]
    assert [c['name'] for c in sc['configuration']['index_codecs']] == ['bytes', 'crc32c']
    rng = np.random.default_rng(5)
    ref = rng.integers(0, 60000, size=(20, 33, 17), dtype=np.uint16)
    a[...] = ref
    b = zstore.open_store(str(tmp_path / 's.zarr'), mode='r')['v']
    assert b.chunks == (8, 16, 16) and b.inner == (4, 8, 8) and b.shape == ref.shape
    np.testing.assert_array_equal(b[...], ref)
    np.testing.assert_array_equal(b[3:11, 5:30, 2:17], ref[3:11, 5:30, 2:17])
    np.testing.assert_array_equal(b[19], ref[19])
    # a shard file: the index sits at the end, 2 x 2 x 2 inner chunks x 16 bytes + 4 bytes of CRC-32C over it
    raw = open(tmp_path / 's.zarr' / 'v' / 'c' / '0' / '0' / '0', 'rb').read()
    idx = np.frombuffer(raw[-(16 * 8 + 4):-4], dtype='<u8').reshape(8, 2)
    assert int(idx[0, 0]) == 0 and int(idx[:, 1].sum()) == len(raw) - (16 * 8 + 4)
    assert zstore.crc32c(raw[-(16 * 8 + 4):-4]) == int.from_bytes(raw[-4:], 'little')
    if compressor is None:
        assert int(idx[0, 1]) == 4 * 8 * 8 * 2
        np.testing.assert_array_equal(np.frombuffer(raw[:512], '<u2').reshape(4, 8, 8), ref[:4, :8, :8])
    # partial write into two shards
    a[6:10, 10:20, 0:3] = 7
    ref[6:10, 10:20, 0:3] = 7
    np.testing.assert_array_equal(zstore.open_store(str(tmp_path / 's.zarr'), mode='r')['v'][...], ref)


def test_sharded_v3_reader_handles_absent_chunks_index_at_start_and_bad_checksums(tmp_path):
    """a shard written elsewhere: index at the START, one inner chunk absent (offset = nbytes = 2^64 - 1 -> fill value),
    chunks stored out of order; and a corrupted index is refused"""
    d = tmp_path / 'w.zarr' / 'v'
    os.makedirs(d / 'c' / '0')
    meta = {'zarr_format': 3, 'node_type': 'array', 'shape': [4, 8], 'data_type': 'int32',
            'chunk_grid': {'name': 'regular', 'configuration': {'chunk_shape': [4, 8]}},
            'chunk_key_encoding': {'name': 'default', 'configuration': {'separator': '/'}}, 'fill_value': -3,
            'codecs': [{'name': 'sharding_indexed', 'configuration': {
                'chunk_shape': [2, 4], 'codecs': [{'name': 'bytes', 'configuration': {'endian': 'little'}}],
                'index_codecs': [{'name': 'bytes', 'configuration': {'endian': 'little'}}, {'name': 'crc32c'}],
                'index_location': 'start'}}], 'attributes': {}}
    json.dump({'zarr_format': 3, 'node_type': 'group'}, open(tmp_path / 'w.zarr' / 'zarr.json', 'w'))
    json.dump(meta, open(d / 'zarr.json', 'w'))
    ref = np.arange(32, dtype='<i4').reshape(4, 8)
    chunks = {(0, 0): ref[:2, :4], (0, 1): ref[:2, 4:], (1, 1): ref[2:, 4:]}       # (1, 0) absent
    isz = 16 * 4 + 4
    order = [(1, 1), (0, 0), (0, 1)]                                               # stored out of order
    index = np.full((4, 2), 0xFFFFFFFFFFFFFFFF, dtype='<u8')
    body, off = b'', isz
    for pos in order:
        data = np.ascontiguousarray(chunks[pos]).tobytes()
        index[pos[0] * 2 + pos[1]] = (off, len(data))
        body += data
        off += len(data)
    ib = index.tobytes()
    open(d / 'c' / '0' / '0', 'wb').write(ib + zstore.crc32c(ib).to_bytes(4, 'little') + body)
    a = zstore.open_store(str(tmp_path / 'w.zarr'), mode='r')['v']
    want = ref.copy()
    want[2:, :4] = -3
    np.testing.assert_array_equal(a[...], want)
    bad = bytearray(open(d / 'c' / '0' / '0', 'rb').read())
    bad[3] ^= 0x40
    open(d / 'c' / '0' / '0', 'wb').write(bytes(bad))
    with pytest.raises(ValueError, match='checksum'):
        zstore.open_store(str(tmp_path / 'w.zarr'), mode='r')['v'][...]
