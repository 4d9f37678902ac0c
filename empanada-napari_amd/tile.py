"""2-D tiled inference: tile layout, overlap band, tile -> image translation
(reference: empanada/inference/tile.py:8-195).

The reference delegates the layout to cztile's ``AlmostEqualBorderFixedTotalAreaStrategy2D``
(``cztile>=2.0.0``, not vendored): ``tile_ranges_1d`` keeps its contract -- every tile has the
full size, neighbours overlap by at least ``overlap_width``, overlaps almost equal -- but the
exact pixel positions are PARITY UNPINNED.  ``Tiler`` also accepts explicit ranges, so a caller
holding cztile's rectangles can pass them in.  Everything downstream of the rectangles
(overlap band, translation, tile consensus) is pinned by ``tests/golden/tiles.npz``.
"""
import numpy as np

from . import sparse


def tile_ranges_1d(length, tile, overlap):
    tile = min(tile, length)
    if length <= tile:
        return [(0, length)]
    n = -(-(length - overlap) // (tile - overlap))
    return [((i * (length - tile)) // (n - 1), (i * (length - tile)) // (n - 1) + tile) for i in range(n)]


def calculate_overlap_rle(yranges, xranges, image_shape):
    """tile.py:8-52: run-length encoding of the pixels covered by at least two tile rows / columns."""
    uy = np.unique(np.stack(yranges, axis=0), axis=0).astype(np.int64)
    ux = np.unique(np.stack(xranges, axis=0), axis=0).astype(np.int64)
    y = sparse.rle_voting(uy, 2)
    x = sparse.rle_voting(ux, 2)
    w = image_shape[1]
    parts = []
    if len(y) > 0:
        parts.append(np.stack([y[:, 0] * w, y[:, 1] * w], axis=1))
    if len(x) > 0:
        rows = np.arange(image_shape[0], dtype=np.int64)[:, None, None] * w
        parts.append((x[None] + rows).reshape(-1, 2))
    if not parts:
        return [], []
    j = sparse.join_ranges(parts)
    return j[:, 0], j[:, 1] - j[:, 0]


class Tiler:
    """tile.py:54-195."""

    def __init__(self, image_shape, tile_size=2048, overlap_width=128, yranges=None, xranges=None):
        if isinstance(tile_size, int):
            tile_size = (tile_size, tile_size)
        assert isinstance(overlap_width, int)
        assert len(image_shape) == 2, 'Tiler only works with 2D images'
        self.image_shape = tuple(image_shape)
        self.tile_size = tile_size
        self.overlap_width = overlap_width
        if yranges is None:
            ys = tile_ranges_1d(image_shape[0], tile_size[0], overlap_width)
            xs = tile_ranges_1d(image_shape[1], tile_size[1], overlap_width)
            yranges = [y for y in ys for _ in xs]
            xranges = [x for _ in ys for x in xs]
        self.yranges = [tuple(int(v) for v in r) for r in yranges]
        self.xranges = [tuple(int(v) for v in r) for r in xranges]
        self.overlap_rle = calculate_overlap_rle(self.yranges, self.xranges, self.image_shape)

    def __len__(self):
        return len(self.yranges)

    def overlap_mask(self):
        overlap = np.zeros(int(np.prod(self.image_shape)))
        for s, r in zip(self.overlap_rle[0], self.overlap_rle[1]):
            overlap[s:s + r] = 1
        return overlap.reshape(self.image_shape)

    def translate_rle_seg(self, rle_seg, tile_index):
        """Boxes and run starts from the tile frame to the image frame, in place (tile.py:133-172)."""
        ys, _ = self.yranges[tile_index]
        xs, xe = self.xranges[tile_index]
        w = xe - xs
        for labels in rle_seg.values():
            for a in labels.values():
                b = a['box']
                a['box'] = (b[0] + ys, b[1] + xs, b[2] + ys, b[3] + xs)
                st = np.asarray(a['starts'], dtype=np.int64)
                a['starts'] = (st // w + ys) * self.image_shape[1] + st % w + xs
        return rle_seg

    def __call__(self, image, tile_index):
        if tile_index >= len(self):
            raise IndexError('Tile index out of range')
        assert image.shape == self.image_shape, \
            f'Image shape of {image.shape} does not match tiler expected shape {self.image_shape}'
        return image[slice(*self.yranges[tile_index]), slice(*self.xranges[tile_index])]
