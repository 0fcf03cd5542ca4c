//! Canonical byte dump of a `FlatScene` — format "RTMIFLT1" of tools/dump_flat_scene.py — for diffing the Rust
//! lowering against the golden dumps of the C++ lowering (tests/golden/flat_*.bin.gz).
//! UNVERIFIED SOURCE (no Rust toolchain in the build image).
use crate::sys::*;
use crate::FlatScene;

fn fnv1a64(data: &[u8]) -> u64 {
    let mut h: u64 = 0xCBF2_9CE4_8422_2325;
    for b in data {
        h = (h ^ *b as u64).wrapping_mul(0x0000_0100_0000_01B3);
    }
    h
}
fn raw<T: Copy>(out: &mut Vec<u8>, v: &[T]) {
    // #[repr(C)] plain-old-data without padding bytes (all fields are 4- or 8-byte scalars, sizes multiples of 8 or 4)
    let bytes = unsafe { std::slice::from_raw_parts(v.as_ptr() as *const u8, std::mem::size_of_val(v)) };
    out.extend_from_slice(bytes);
}

pub fn flat_scene_bytes(s: &FlatScene) -> Vec<u8> {
    let mut out = Vec::new();
    out.extend_from_slice(b"RTMIFLT1");
    for n in [
        s.items.len(), s.prim_meta.len(), s.nodes.len(), s.alt_nodes.len(), s.xforms.len(), s.materials.len(), s.textures.len(),
        s.perlin.len(), s.images.len(), s.max_bvh_depth as usize, s.alt_max_depth as usize,
    ] {
        out.extend_from_slice(&(n as u32).to_le_bytes());
    }
    out.extend_from_slice(&s.bvh_time_lo.to_le_bytes());
    out.extend_from_slice(&s.bvh_time_hi.to_le_bytes());
    out.extend_from_slice(&(s.image_data.len() as u64).to_le_bytes());
    out.extend_from_slice(&fnv1a64(&s.image_data).to_le_bytes());
    raw(&mut out, &s.items);
    raw(&mut out, &s.prim_a);
    raw(&mut out, &s.prim_b);
    raw(&mut out, &s.prim_meta);
    raw(&mut out, &s.prim_gate);
    let nodes: Vec<RtmiBvhNode> = s.nodes.iter().map(|n| RtmiBvhNode { pad: [0; 2], ..*n }).collect();
    raw(&mut out, &nodes);
    let alt: Vec<RtmiBvh4Node> = s.alt_nodes.iter().map(|n| RtmiBvh4Node { pad: [0; 4], ..*n }).collect();
    raw(&mut out, &alt);
    raw(&mut out, &s.xforms);
    raw(&mut out, &s.materials);
    raw(&mut out, &s.textures);
    raw(&mut out, &s.perlin);
    raw(&mut out, &s.images);
    out
}

#[cfg(test)]
mod tests {
    //! `cargo test` on a machine with a toolchain: gunzip tests/golden/flat_*.bin.gz next to the crate first
    //! (`gzip -dk`), and provide the decoded earth texture as earthmap.rgb8 (1024 x 512 x 3, row-major) —
    //! `python -c "from raytracing_rust_amd import scenes; scenes.earthmap_rgb8()[0].tofile('earthmap.rgb8')"`.
    use super::*;
    use crate::{lower, scenes};
    fn golden(name: &str) -> Option<Vec<u8>> {
        std::fs::read(format!("{}/../../tests/golden/flat_{}.bin", env!("CARGO_MANIFEST_DIR"), name)).ok()
    }
    #[test]
    fn cornell_box_matches_the_cpp_lowering() {
        if let Some(want) = golden("cornell_box") {
            let got = flat_scene_bytes(&lower::lower_world(&scenes::cornell_box(1, false)).unwrap());
            assert!(got == want, "flat cornell_box differs from the C++ lowering");
        }
    }
    #[test]
    fn compositions_match_the_cpp_lowering() {
        // list leaves (tie order), instanced primitives, a flipped subtree, a medium inside transforms
        if let Some(want) = golden("compositions") {
            let got = flat_scene_bytes(&lower::lower_world(&scenes::compositions(1)).unwrap());
            assert!(got == want, "flat compositions differs from the C++ lowering");
        }
    }
    #[test]
    fn media_in_bvh_match_the_cpp_lowering() {
        // media as children of BVHNodes: DEFERRED items behind the BVH item, twice below a node referenced on both sides
        if let Some(want) = golden("media_in_bvh") {
            let got = flat_scene_bytes(&lower::lower_world(&scenes::media_in_bvh(1)).unwrap());
            assert!(got == want, "flat media_in_bvh differs from the C++ lowering");
        }
    }
    #[test]
    fn final_scene_matches_the_cpp_lowering() {
        let earth = std::fs::read(format!("{}/../../earthmap.rgb8", env!("CARGO_MANIFEST_DIR"))).ok();
        if let (Some(want), Some(earth)) = (golden("final_scene"), earth) {
            let got = flat_scene_bytes(&lower::lower_world(&scenes::final_scene(1, false, (earth, 1024, 512))).unwrap());
            assert!(got == want, "flat final_scene differs from the C++ lowering");
        }
    }
}
