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


def _u01(bits):
    """u01_from_bits of device_util.h: (m + 0.5) * 2^-23, m = the top 23 bits (exact in fp32, never 0 or 1)."""
    return ((bits >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)


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
    return _u01(bits)


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
    return _u01(bits)


def philox_uniform_raw_torch(seed: int, t0: int, n: int, rows, device='cpu'):
    """The same draws as ``philox_uniform_raw`` for steps [t0, t0 + n) and the given GLOBAL row indices, evaluated with
    torch int64 arithmetic (on the GPU when ``device`` says so: the numpy replay of a 110 275-step clip costs minutes of
    host time on some boxes).  Returns a float32 tensor (n, len(rows), 1024) on ``device``.  An independent second
    implementation of the generator's specification: 32x32 -> 64-bit products in wrapped int64, hi/lo by shift and mask."""
    import torch
    M0, M1, W0, W1, MASK = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
    t = torch.arange(t0, t0 + n, dtype=torch.int64, device=device)
    th = (t >> 1)[:, None, None]
    r = torch.as_tensor(list(rows), dtype=torch.int64, device=device)[None, :, None]
    k2 = torch.arange(512, dtype=torch.int64, device=device)[None, None, :]
    shape = (n, r.shape[1], 512)
    c0 = (th & MASK).expand(shape).clone()
    c1 = ((th >> 32) & MASK).expand(shape).clone()
    c2 = r.expand(shape).clone()
    c3 = k2.expand(shape).clone()
    k0, k1 = seed & MASK, (seed >> 32) & MASK

    def mulhilo(a, m):
        # a < 2^32, m < 2^32: the int64 product wraps modulo 2^64, which keeps all 64 bits of the unsigned product
        lo_part = a * (m & 0xFFFF)                 # < 2^48
        hi_part = a * (m >> 16)                    # < 2^48
        full_lo = (lo_part + ((hi_part & 0xFFFF) << 16))            # bits 0..48 of the product (no wrap: < 2^49)
        lo = full_lo & MASK
        hi = ((hi_part >> 16) + (full_lo >> 32)) & MASK
        return hi, lo
    for _ in range(10):
        hi0, lo0 = mulhilo(c0, M0)
        hi1, lo1 = mulhilo(c2, M1)
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    odd = (t & 1).bool()[:, None, None]
    e0 = torch.where(odd, c2, c0)   # x | z : class 2j
    e1 = torch.where(odd, c3, c1)   # y | w : class 2j + 1
    bits = torch.stack([e0, e1], dim=-1).reshape(n, r.shape[1], 1024)
    return ((bits >> 9).to(torch.float32) + 0.5) * (1.0 / 8388608.0)
