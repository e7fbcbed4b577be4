"""numpy restatement of the device counter RNG (csrc/device_util.h: philox4x32_10 + wrnn_uniform)."""
import numpy as np


def _philox(c, k):
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
    c0, c1, c2, c3 = [x.astype(np.uint32) for x in c]
    k0, k1 = np.uint32(k[0]), np.uint32(k[1])
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def philox_uniform(seed: int, steps: int, rows: int, n: int) -> np.ndarray:
    """(steps, rows, n) float32 uniforms identical to wrnn_uniform(seed, t, row, k)."""
    t = np.arange(steps, dtype=np.uint64)[:, None, None]
    r = np.arange(rows, dtype=np.uint32)[None, :, None]
    k4 = np.arange((n + 3) // 4, dtype=np.uint32)[None, None, :]
    shape = (steps, rows, k4.shape[-1])
    c0 = np.broadcast_to((t & np.uint64(0xFFFFFFFF)).astype(np.uint32), shape)
    c1 = np.broadcast_to((t >> np.uint64(32)).astype(np.uint32), shape)
    c2 = np.broadcast_to(r, shape)
    c3 = np.broadcast_to(k4, shape)
    out = _philox((c0, c1, c2, c3), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    bits = np.stack(out, axis=-1).reshape(steps, rows, -1)[:, :, :n]
    return ((bits >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def philox_uniform_raw(seed: int, steps: int, rows: int, n: int) -> np.ndarray:
    """(steps, rows, n) uniforms identical to wrnn_uniform_raw(seed, t, row, k):
    block counter (t>>1, row, k>>1), element ((t&1)<<1)|(k&1)."""
    th = (np.arange(steps, dtype=np.uint64) >> np.uint64(1))[:, None, None]
    r = np.arange(rows, dtype=np.uint32)[None, :, None]
    k2 = np.arange((n + 1) // 2, dtype=np.uint32)[None, None, :]
    shape = (steps, rows, k2.shape[-1])
    c0 = np.broadcast_to((th & np.uint64(0xFFFFFFFF)).astype(np.uint32), shape)
    c1 = np.broadcast_to((th >> np.uint64(32)).astype(np.uint32), shape)
    c2 = np.broadcast_to(r, shape)
    c3 = np.broadcast_to(k2, shape)
    x, y, z, w = _philox((c0, c1, c2, c3), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    odd = (np.arange(steps) & 1).astype(bool)[:, None, None]
    e0 = np.where(odd, z, x)   # class 2j
    e1 = np.where(odd, w, y)   # class 2j + 1
    bits = np.stack([e0, e1], axis=-1).reshape(steps, rows, -1)[:, :, :n]
    return ((bits >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
