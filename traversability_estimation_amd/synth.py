"""Synthetic elevation maps for the benchmark configurations: gradient ("Perlin")
noise, 5 octaves, base wavelength 64 cells, persistence 0.5, amplitude 0.5 m, float32.

Deterministic in (rows, cols, seed) with numpy only, so the CPU oracle and the GPU see the same bits.
Arrays are returned in grid_map storage order: shape (cols, rows), C-contiguous, a[j, i] = value(row i, col j).
"""
import numpy as np


def _fade(t):
    return t * t * t * (t * (t * 6.0 - 15.0) + 10.0)


def _perlin_octave(rows, cols, wavelength, rng):
    gi = rows // wavelength + 2
    gj = cols // wavelength + 2
    ang = rng.uniform(0.0, 2.0 * np.pi, size=(gj, gi))
    gx, gy = np.cos(ang), np.sin(ang)
    x = np.arange(rows, dtype=np.float64) / wavelength
    y = np.arange(cols, dtype=np.float64) / wavelength
    xi, yi = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    xf, yf = (x - xi)[None, :], (y - yi)[:, None]
    u, v = _fade(xf), _fade(yf)

    def dot(ix, iy, dx, dy):
        return gx[np.ix_(iy, ix)] * dx + gy[np.ix_(iy, ix)] * dy

    n00 = dot(xi, yi, xf, yf)
    n10 = dot(xi + 1, yi, xf - 1.0, yf)
    n01 = dot(xi, yi + 1, xf, yf - 1.0)
    n11 = dot(xi + 1, yi + 1, xf - 1.0, yf - 1.0)
    nx0 = n00 + u * (n10 - n00)
    nx1 = n01 + u * (n11 - n01)
    return nx0 + v * (nx1 - nx0)


def perlin_elevation(rows, cols, seed, octaves=5, base_wavelength=64, persistence=0.5, amplitude=0.5):
    rng = np.random.default_rng(seed)
    out = np.zeros((cols, rows), dtype=np.float64)
    amp, wl, norm = 1.0, base_wavelength, 0.0
    for _ in range(octaves):
        out += amp * _perlin_octave(rows, cols, max(int(wl), 2), rng)
        norm += amp
        amp *= persistence
        wl /= 2
    out *= amplitude / (norm * 0.7071)  # Perlin noise spans about +-sqrt(2)/2 per octave
    return np.ascontiguousarray(out.astype(np.float32))


def with_holes(elev, fraction, seed):
    """Copy of `elev` with `fraction` of the cells invalid (NaN), like unobserved map cells."""
    rng = np.random.default_rng(seed)
    out = elev.copy()
    mask = rng.random(out.shape) < fraction
    out[mask] = np.nan
    return out


def with_steps(elev, n_boxes, seed, height=(0.05, 0.4)):
    """Copy of `elev` with `n_boxes` raised or lowered rectangles (kerbs, stairs, ditches)."""
    rng = np.random.default_rng(seed)
    out = elev.copy()
    cols, rows = out.shape
    for _ in range(n_boxes):
        h = rng.uniform(*height) * rng.choice([-1.0, 1.0])
        w, l = rng.integers(2, max(3, rows // 6)), rng.integers(2, max(3, cols // 6))
        i0, j0 = rng.integers(0, rows - 1), rng.integers(0, cols - 1)
        out[j0:j0 + l, i0:i0 + w] += np.float32(h)
    return out


def benchmark_radius(cells, res):
    """Tie-free radius of `cells` cells: r = cells * res * (1 + 1e-6)."""
    return cells * res * (1.0 + 1e-6)
