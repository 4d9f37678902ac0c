"""Seeded synthetic inputs (no datasets or checkpoints are reachable offline).

* ``em_tiles``     -- EM-like uint8 tiles: band-limited noise + dark ellipses
                      (SURVEY.md section 8d, config 2).
* ``blob_image`` / ``blob_volume`` -- Gaussian-blob images in the style of the
  reference's sanity fixtures (tests/test_button_widgets.py:26-50,119-140).
* ``head_outputs`` -- plausible network-head tensors (semantic logits, centre
  heatmap, offsets) with a known number of objects, for post-processing tests.
"""
import numpy as np


def _smooth(img, passes=2):
    for _ in range(passes):
        img = (img + np.roll(img, 1, 0) + np.roll(img, -1, 0) + np.roll(img, 1, 1) + np.roll(img, -1, 1)) / 5.0
    return img


def em_tiles(n, size, seed=1234, n_ellipses=150):
    """(n, size, size) uint8."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    out = np.empty((n, size, size), dtype=np.uint8)
    n_ell = max(1, int(n_ellipses * (size / 1024.0) ** 2))
    for t in range(n):
        img = _smooth(rng.standard_normal((size, size)).astype(np.float32), 3) * 60.0 + 150.0
        for _ in range(n_ell):
            cy, cx = rng.uniform(0, size, 2)
            a, b = rng.uniform(8, 40, 2)
            th = rng.uniform(0, np.pi)
            r = max(a, b) + 2
            y0, y1 = int(max(0, cy - r)), int(min(size, cy + r + 1))
            x0, x1 = int(max(0, cx - r)), int(min(size, cx + r + 1))
            if y0 >= y1 or x0 >= x1:
                continue
            dy = yy[y0:y1, x0:x1] - cy
            dx = xx[y0:y1, x0:x1] - cx
            u = (dx * np.cos(th) + dy * np.sin(th)) / a
            v = (-dx * np.sin(th) + dy * np.cos(th)) / b
            m = (u * u + v * v) < 1.0
            img[y0:y1, x0:x1][m] -= rng.uniform(50, 90)
        out[t] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def blob_image(h, w, seed=0, n_blobs=8):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w), dtype=np.float32)
    s = max(h, w) / 100.0
    for _ in range(n_blobs):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        sigma = rng.uniform(4, 10) * s
        amp = rng.uniform(120, 255)
        img += amp * np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / (2 * sigma ** 2))
    img += rng.normal(0, 10, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def blob_volume(d, h, w, seed=0, n_blobs=8, fast=False):
    """Gaussian blobs + noise, uint8 (the reference's test volumes, test_button_widgets.py:119-140, at any size).
    ``fast``: the same blobs summed as one GEMM of separable factors (float32 rounding differs in isolated voxels from
    the plain form the small test fixtures were made with); for the 512^3 bench volume this is seconds, not minutes."""
    rng = np.random.default_rng(seed)
    if fast:
        s = max(d, h, w) / 100.0
        par = np.array([[rng.uniform(0, w), rng.uniform(0, h), rng.uniform(0, d), rng.uniform(4, 10) * s,
                         rng.uniform(120, 255)] for _ in range(n_blobs)], dtype=np.float64)
        g = lambda n, c: np.exp(-((np.arange(n)[None, :] - c[:, None]) ** 2) / (2 * par[:, 3:4] ** 2)).astype(np.float32)
        ez, ey, ex = g(d, par[:, 2]), g(h, par[:, 1]), g(w, par[:, 0])
        plane = (ey[:, :, None] * ex[:, None, :]).reshape(n_blobs, h * w)
        vol = ((ez * par[:, 4:5].astype(np.float32)).T @ plane).reshape(d, h, w)
        vol += rng.normal(0, 10, size=vol.shape).astype(np.float32)
        return np.clip(vol, 0, 255).astype(np.uint8)
    z, y, x = np.mgrid[0:d, 0:h, 0:w]
    vol = np.zeros((d, h, w), dtype=np.float32)
    s = max(d, h, w) / 100.0
    for _ in range(n_blobs):
        cx, cy, cz = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(0, d)
        sigma = rng.uniform(4, 10) * s
        amp = rng.uniform(120, 255)
        vol += amp * np.exp(-((z - cz) ** 2 + (x - cx) ** 2 + (y - cy) ** 2) / (2 * sigma ** 2))
    vol += rng.normal(0, 10, size=vol.shape)
    return np.clip(vol, 0, 255).astype(np.uint8)


def head_outputs(H, W, n_inst, seed=0, coarse=True, num_classes=1, plateau=False):
    """Synthetic head tensors for an (H,W) slice.

    Returns fp32 ``sem_logits (1,C,H,W)``, ``ctr_hmp (1,1,h,w)``,
    ``offsets (1,2,h,w)`` with h,w = H/4,W/4 when ``coarse`` else H,W.
    ``n_inst`` discs with a heat-map peak each; ``plateau`` makes a few peaks
    flat 2x1 plateaus (equal values -> several centres).
    """
    rng = np.random.default_rng(seed)
    s = 4 if coarse else 1
    h, w = H // s, W // s
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    ctr = np.zeros((h, w), np.float32)
    off = rng.standard_normal((2, h, w)).astype(np.float32) * 0.5
    semq = np.full((h, w), -3.0, np.float32)
    cls = np.zeros((h, w), np.int64)
    nearest = np.full((h, w), np.inf, np.float32)
    for i in range(n_inst):
        cy, cx = rng.uniform(2, h - 2), rng.uniform(2, w - 2)
        r = rng.uniform(2.5, max(3.0, min(h, w) / 10.0))
        d2 = (yy - cy) ** 2 + (xx - cx) ** 2
        ctr = np.maximum(ctr, rng.uniform(0.3, 1.0) * np.exp(-d2 / (2 * (r / 2.5) ** 2)).astype(np.float32))
        inside = d2 < r * r
        closer = inside & (d2 < nearest)
        nearest[closer] = d2[closer]
        off[0][closer] = ((cy - yy) * s)[closer] + off[0][closer] * 0.2
        off[1][closer] = ((cx - xx) * s)[closer] + off[1][closer] * 0.2
        semq[inside] = 3.0
        cls[inside] = 1 + (i % max(1, num_classes - 1))
        if plateau and i % 3 == 0:
            iy, ix = int(round(cy)), int(round(cx))
            if 0 <= iy < h and 0 <= ix + 1 < w:
                ctr[iy, ix] = ctr[iy, ix + 1] = 1.5
    ctr = ctr + rng.uniform(0, 0.02, (h, w)).astype(np.float32)
    # full-resolution semantic logits: nearest-upsampled discs + noise
    if num_classes == 1:
        sem = np.repeat(np.repeat(semq, s, 0), s, 1)[None]
        sem = sem + rng.standard_normal(sem.shape).astype(np.float32) * 1.5
    else:
        clsf = np.repeat(np.repeat(cls, s, 0), s, 1)
        sem = rng.standard_normal((num_classes, H, W)).astype(np.float32)
        for c in range(num_classes):
            sem[c][clsf == c] += 3.0
    return (np.ascontiguousarray(sem[None].astype(np.float32)),
            np.ascontiguousarray(ctr[None, None].astype(np.float32)),
            np.ascontiguousarray(off[None].astype(np.float32)))


class ProceduralVolume:
    """A (D,H,W) uint8 EM-like volume that is never stored: every voxel is a pure function of (seed, z, y, x), so any
    rank synthesises any block of it (the 4096^3 volume of BASELINE configs[3] would be
    64 GiB).  Content: one Gaussian blob per cell of a ``cell``-sized lattice (centre, width and amplitude hashed
    from the cell index; a voxel sums the blobs of its 27 neighbouring cells) plus hashed per-voxel noise -- the
    reference's blob fixtures (tests/test_button_widgets.py:119-140) at any size.

    ``block(axis, lo, hi, device)`` -> uint8 torch tensor (hi-lo, A, B): slices lo..hi-1 along ``axis`` in the layout
    ``np.moveaxis(volume, axis, 0)`` gives, computed with torch on ``device`` (integer hashing is exact everywhere; the
    float blob sum is evaluated the same way on every rank of one device type).  ``numpy()`` materialises it whole
    (small shapes: tests)."""

    def __init__(self, shape, seed=0, cell=48, dtype=np.uint8, cache=False):
        self.shape = tuple(int(s) for s in shape)
        assert len(self.shape) == 3
        self.seed, self.cell = int(seed), int(cell)
        self.dtype = np.dtype(dtype)
        self.ndim = 3
        # cache=True keeps every block it has synthesised (per process, on the block's device): a benchmark pays for
        # the synthesis once, in its warm-up pass, not inside the timed region (the real volume would already exist)
        self.cache = {} if cache else None

    def __getstate__(self):
        d = dict(self.__dict__)
        if d.get('cache') is not None:
            d['cache'] = {}
        return d

    @staticmethod
    def _mix(x):
        """splitmix64 finaliser on int64 tensors (wrapping arithmetic)."""
        import torch
        x = (x ^ (x >> 30) & 0x3FFFFFFFF) * -4658895280553007687          # 0xBF58476D1CE4E5B9
        x = (x ^ (x >> 27) & 0x1FFFFFFFFF) * -7723592293110705685         # 0x94D049BB133111EB
        return x ^ ((x >> 31) & 0x1FFFFFFFF)

    def _u01(self, key, salt):
        """uniform [0,1) floats from int64 keys."""
        h = self._mix(key * 6364136223846793005 + (self.seed * 1442695040888963407 + salt * 1013904223))
        return ((h >> 11) & 0xFFFFFF).to(__import__('torch').float32) * (1.0 / 16777216.0)

    def block(self, axis, lo, hi, device='cpu'):
        import torch
        dev = torch.device(device)
        if self.cache is None:
            return self._block(axis, lo, hi, dev)
        key = (axis, lo, hi, str(dev))
        if key not in self.cache:
            self.cache[key] = self._block(axis, lo, hi, dev)
        return self.cache[key]

    def _block(self, axis, lo, hi, dev):
        import torch
        D, H, W = self.shape
        rng_ax = [torch.arange(n, device=dev, dtype=torch.int64) for n in (D, H, W)]
        rng_ax[axis] = rng_ax[axis][lo:hi]
        z, y, x = torch.meshgrid(*rng_ax, indexing='ij')
        c = self.cell
        cz, cy, cx = z // c, y // c, x // c
        val = torch.zeros(z.shape, dtype=torch.float32, device=dev)
        zf, yf, xf = z.float(), y.float(), x.float()
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    kz, ky, kx = cz + dz, cy + dy, cx + dx
                    key = (kz * 1000003 + ky) * 1000003 + kx
                    bz = (kz.float() + self._u01(key, 1)) * c
                    by = (ky.float() + self._u01(key, 2)) * c
                    bx = (kx.float() + self._u01(key, 3)) * c
                    sig = (0.18 + 0.14 * self._u01(key, 4)) * c
                    amp = 120.0 + 135.0 * self._u01(key, 5)
                    d2 = (zf - bz) ** 2 + (yf - by) ** 2 + (xf - bx) ** 2
                    val += amp * torch.exp(-d2 / (2.0 * sig * sig))
        lin = (z * H + y) * W + x
        noise = (self._u01(lin, 7) + self._u01(lin, 8) + self._u01(lin, 9) + self._u01(lin, 10) - 2.0) * 17.3   # ~N(0, 10)
        out = torch.clamp(val + noise, 0, 255).to(torch.uint8)
        return out.movedim(axis, 0).contiguous()

    def numpy(self):
        return self.block(0, 0, self.shape[0], 'cpu').numpy()

    def __getitem__(self, key):
        return self.numpy()[key]
