"""Philox4x32-10 counter RNG (Salmon et al., SC'11) — host-side (numpy) implementation.

The reference draws every random number from rand::thread_rng() (OS-seeded ChaCha12;
tests/test.rs:56, src/util.rs:5,16, src/camera.rs:61, src/bvh.rs:40, src/perlin.rs:5,13),
which is not reproducible.  This build replaces it with keyed counter streams:

    counter = (block, sample, pixel, stream_id), key = (seed_lo, seed_hi)
    stream_id 0: render path, one stream per (pixel, sample)   [device + oracle]
    stream_id 1: scene construction inside the host library    [BVH axes, Perlin tables]
    stream_id 2: scene construction in the scene builders      [positions, albedos]

The n-th draw of a stream is word n % 4 of block n // 4; a uniform is the top 24 bits
of the word times 2^-24 (exact in fp32 and f64).
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: 4 x uint32, key: 2 x uint32 -> 4 x uint32 (Python ints)."""
    c0, c1, c2, c3 = (int(x) & 0xFFFFFFFF for x in ctr)
    k0, k1 = (int(x) & 0xFFFFFFFF for x in key)
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        n0 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF
        n1 = p1 & 0xFFFFFFFF
        n2 = ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF
        n3 = p0 & 0xFFFFFFFF
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return [c0, c1, c2, c3]


class Stream:
    """Sequential draws from one Philox stream."""

    def __init__(self, seed, sample=0, pixel=0, stream_id=0):
        self.key = [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF]
        self.ctr = [0, sample & 0xFFFFFFFF, pixel & 0xFFFFFFFF, stream_id & 0xFFFFFFFF]
        self.buf = []

    def u32(self):
        if not self.buf:
            self.buf = philox4x32_10(self.ctr, self.key)
            self.ctr[0] = (self.ctr[0] + 1) & 0xFFFFFFFF
        return self.buf.pop(0)

    def uniform(self):
        """The build's `rng.gen::<f64>()`: 24-bit uniform in [0,1)."""
        return (self.u32() >> 8) * (1.0 / 16777216.0)

    def range(self, n):
        """The build's `rng.gen_range(0..n)`."""
        return (self.u32() * n) >> 32


class SceneRng(Stream):
    """thread_rng() of the scene builders (stream_id 2)."""

    def __init__(self, seed):
        super().__init__(seed, 0, 0, 2)

    def gen(self):
        return self.uniform()
