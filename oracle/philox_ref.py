"""CPU oracle for the in-kernel noise of the stochastic integrators (TEST INFRASTRUCTURE -- see oracle/__init__.py).

The reference draws ``torch.randn_like(x)`` once per step on the device generator
(diffsci/models/karras/integrators.py:66-69 Euler-Maruyama, :103-104 sigma-churn); those draws are not
reproducible across backends, so what is restated here is OUR stream definition (include/diffsci_hip.h,
ds_philox_normal): Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the
published Random123 algorithm, pinned by its known-answer vectors in tests/test_oracle_golden.py) followed by
Box-Muller:

    counter = (base_offset + step_offset + e // 4) as (lo32, hi32, 0, 0),  key = (seed lo32, seed hi32)
    (r0, r1, r2, r3) = philox4x32_10(counter, key)
    u1 = fl(fl(float(r) * 2^-32) + 2^-33)   in (0, 1]        u2 = fl(float(r') * 2^-32)   in [0, 1]
    z  = sqrt(-2 ln u1) * (cos(2 pi u2), sin(2 pi u2))        (r0, r1) -> elements 4k, 4k+1; (r2, r3) -> 4k+2, 4k+3

numpy, vectorised; the uniform -> normal map is evaluated in float64 from the float32 uniforms (the kernel's
logf / sincospif differ from any host libm in the last ulp, so GPU tests compare within a few ulp)."""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(counter, key):
    """counter [n, 4] uint32 (as uint64 values), key [n, 2] -> [n, 4] uint32.  Ten rounds."""
    c = [np.asarray(counter[:, i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[:, 0], dtype=np.uint64)
    k1 = np.asarray(key[:, 1], dtype=np.uint64)
    for _ in range(10):
        p0 = np.uint64(M0) * c[0]
        p1 = np.uint64(M1) * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & np.uint64(MASK)
        hi1, lo1 = p1 >> np.uint64(32), p1 & np.uint64(MASK)
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + np.uint64(W0)) & np.uint64(MASK)
        k1 = (k1 + np.uint64(W1)) & np.uint64(MASK)
    return np.stack(c, axis=1).astype(np.uint32)


def uniforms(r0, r1):
    """The kernel's two uniforms of one Box-Muller pair, in its float32 operation order."""
    f = np.float32
    u1 = (r0.astype(f) * f(2.3283064365386963e-10)).astype(f) + f(1.1641532182693481e-10)
    u2 = (r1.astype(f) * f(2.3283064365386963e-10)).astype(f)
    return u1.astype(f), u2


def normal(seed, offset, n):
    """The n standard normals of ds_philox_normal(state = (seed, base), offset) with offset := base + philox_offset."""
    n4 = (n + 3) // 4
    ctr = (np.uint64(offset & (2**64 - 1)) + np.arange(n4, dtype=np.uint64))
    counter = np.stack([ctr & np.uint64(MASK), ctr >> np.uint64(32), np.zeros_like(ctr), np.zeros_like(ctr)], axis=1)
    seed = int(seed) & (2**64 - 1)
    key = np.tile(np.array([[seed & MASK, seed >> 32]], dtype=np.uint64), (n4, 1))
    r = philox4x32_10(counter, key)
    out = np.empty((n4, 4), dtype=np.float64)
    for j in (0, 2):
        u1, u2 = uniforms(r[:, j], r[:, j + 1])
        rad = np.sqrt(-2.0 * np.log(u1.astype(np.float64)))
        ang = 2.0 * np.pi * u2.astype(np.float64)
        out[:, j] = rad * np.cos(ang)
        out[:, j + 1] = rad * np.sin(ang)
    return out.reshape(-1)[:n]
