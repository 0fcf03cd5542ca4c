//! Philox4x32-10 (Salmon et al., SC'11) and the scene-construction streams — the counterpart of
//! raytracing_rust_amd/philox.py and of `rt::Rng` (raytracing_rust_amd/host/rt_host.hpp:65-77).
//!
//! The reference draws from `rand::thread_rng()` (OS-seeded ChaCha12: tests/test.rs:56, src/bvh.rs:40,
//! src/perlin.rs:5,13), which is not reproducible.  Scenes built through this crate use keyed counter streams
//!     counter = (block, sample, pixel, stream_id), key = (seed_lo, seed_hi)
//!     stream_id 1: BVH split axes (bvh.rs:40) and Perlin tables (perlin.rs:7,18-20)   [`SceneStreams::backend`]
//!     stream_id 2: the scene builders' own draws (tests/test.rs:105-143, 433, 510-512)   [`SceneStreams::builder`]
//! so that a scene built here with seed s is the scene the C++ mirror and the Python builders build with seed s.
//! UNVERIFIED SOURCE (no Rust toolchain in the build image).

const M0: u64 = 0xD251_1F53;
const M1: u64 = 0xCD9E_8D57;
const W0: u32 = 0x9E37_79B9;
const W1: u32 = 0xBB67_AE85;

pub fn philox4x32_10(ctr: [u32; 4], key: [u32; 2]) -> [u32; 4] {
    let (mut c0, mut c1, mut c2, mut c3) = (ctr[0], ctr[1], ctr[2], ctr[3]);
    let (mut k0, mut k1) = (key[0], key[1]);
    for _ in 0..10 {
        let p0 = M0 * c0 as u64;
        let p1 = M1 * c2 as u64;
        let n0 = ((p1 >> 32) as u32) ^ c1 ^ k0;
        let n1 = p1 as u32;
        let n2 = ((p0 >> 32) as u32) ^ c3 ^ k1;
        let n3 = p0 as u32;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 = k0.wrapping_add(W0);
        k1 = k1.wrapping_add(W1);
    }
    [c0, c1, c2, c3]
}

/// Sequential draws from one stream: the n-th draw is word n % 4 of block n / 4.
pub struct Stream {
    key: [u32; 2],
    ctr: [u32; 4],
    buf: [u32; 4],
    pos: usize,
}

impl Stream {
    pub fn new(seed: u64, sample: u32, pixel: u32, stream_id: u32) -> Self {
        Stream { key: [seed as u32, (seed >> 32) as u32], ctr: [0, sample, pixel, stream_id], buf: [0; 4], pos: 4 }
    }
    pub fn next_u32(&mut self) -> u32 {
        if self.pos == 4 {
            self.buf = philox4x32_10(self.ctr, self.key);
            self.ctr[0] = self.ctr[0].wrapping_add(1);
            self.pos = 0;
        }
        let w = self.buf[self.pos];
        self.pos += 1;
        w
    }
    /// The build's `rng.gen::<f64>()`: a 24-bit uniform in [0, 1) (exact in f32 and f64).
    pub fn gen(&mut self) -> f64 {
        (self.next_u32() >> 8) as f64 * (1.0 / 16_777_216.0)
    }
    /// The build's `rng.gen_range(0..n)`.
    pub fn gen_range(&mut self, n: u32) -> u32 {
        ((self.next_u32() as u64 * n as u64) >> 32) as u32
    }
}

/// The two construction streams of one scene seed.
pub struct SceneStreams {
    pub backend: Stream,
    pub builder: Stream,
}

impl SceneStreams {
    pub fn new(seed: u64) -> Self {
        SceneStreams { backend: Stream::new(seed, 0, 0, 1), builder: Stream::new(seed, 0, 0, 2) }
    }
}

#[cfg(test)]
mod tests {
    use super::*;
    // Random123 known-answer vectors (kat_vectors: philox4x32 10) — the same ones tests/test_philox.py checks
    #[test]
    fn kat() {
        assert_eq!(philox4x32_10([0; 4], [0; 2]), [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]);
        assert_eq!(philox4x32_10([0xffffffff; 4], [0xffffffff; 2]), [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]);
        assert_eq!(
            philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]),
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
        );
    }
}
