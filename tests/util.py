"""shared helpers for the tests (synthetic level data identical to oracle/gen_golden.py)"""
import numpy as np

from oracle import oracle_np as onp


def level_arrays(N, steps, M=1, nan_every=0, seed=1234):
    """list over levels of (fine[M, n], coarse[M, n] or None) -- same data as gen_golden._levels"""
    out = []
    for l in range(len(N)):
        fine, coarse = onp.synth_level_samples(l, int(N[l]), steps, seed=seed)
        f = np.empty((M, int(N[l])))
        c = np.empty((M, int(N[l])))
        for m in range(M):
            f[m] = fine + 0.125 * m
            c[m] = (coarse + 0.125 * m) if l > 0 else 0.0
        if nan_every:
            f[0, ::nan_every] = np.nan
            if l > 0:
                c[M - 1, 3::nan_every * 2] = np.nan
        out.append((f, c if l > 0 else None))
    return out


def to_chunks(levels):
    """(fine[M,n], coarse[M,n]|None) per level -> oracle raw chunks [[x[M, n, 2|1]], ...]"""
    chunks = []
    for f, c in levels:
        f = np.atleast_2d(f)
        if c is None:
            chunks.append([f[:, :, None]])
        else:
            chunks.append([np.stack([f, np.atleast_2d(c)], axis=-1)])
    return chunks


def close(a, b, scale=None, tol=1e-10):
    """|a - b| <= tol * max(|b|, scale) elementwise (SURVEY 8(d) parity gate)"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        a = a.reshape(b.shape)
    special = ~np.isfinite(b)                     # NaN / inf entries (empty or single-sample levels) must match exactly
    if not np.array_equal(a[special], b[special], equal_nan=True):
        return False
    ref = np.abs(b) if scale is None else np.maximum(np.abs(b), scale)
    ok = np.abs(a - b) <= tol * ref + 1e-300
    return bool(np.all(ok[~special]))
