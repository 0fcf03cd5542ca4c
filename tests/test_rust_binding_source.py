"""bindings/rust/src/sys.rs cannot be compiled here (no Rust toolchain), so keep it honest another way:
every #[repr(C)] struct must list the same fields, in the same order and with the same sizes, as the
tested ctypes structure of raytracing_rust_amd/abi.py, and every function include/rtmi.h declares must
have an extern declaration."""
import ctypes as C
import os
import re

from raytracing_rust_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYS = open(os.path.join(ROOT, "bindings", "rust", "src", "sys.rs")).read()
PAIRS = {"RtmiTexture": abi.Texture, "RtmiPerlin": abi.Perlin, "RtmiImage": abi.ImageDesc, "RtmiMaterial": abi.Material,
         "RtmiPrimMeta": abi.PrimMeta, "RtmiBvhNode": abi.BvhNode, "RtmiBvh4Node": abi.Bvh4Node, "RtmiXform": abi.Xform, "RtmiItem": abi.Item,
         "RtmiSceneDesc": abi.SceneDesc, "RtmiCamera": abi.Camera, "RtmiRenderParams": abi.RenderParams,
         "RtmiTexel": abi.Texel, "RtmiStats": abi.Stats}
SCALAR = {"i32": 4, "u32": 4, "f32": 4, "u64": 8, "f64": 8, "u8": 1}


def rust_fields(name):
    body = re.search(r"pub struct %s \{(.*?)\n\}" % name, SYS, re.S).group(1)
    out = []
    for fname, ty in re.findall(r"pub (r#\w+|\w+): ([^,\n]+),", body):
        ty = ty.strip()
        m = re.match(r"\[(\w+); (\d+)\]", ty)
        size = SCALAR[m.group(1)] * int(m.group(2)) if m else (8 if ty.startswith("*") else SCALAR[ty])
        out.append((fname.replace("r#", ""), size))
    return out


def test_structs_match_ctypes():
    for rname, cty in PAIRS.items():
        rf = rust_fields(rname)
        cf = [(n, C.sizeof(t)) for n, t in cty._fields_]
        assert [n for n, _ in rf] == [n for n, _ in cf], rname
        assert [s for _, s in rf] == [s for _, s in cf], rname


def test_every_entry_point_is_declared():
    for sym in abi.RTMI_SYMBOLS:
        assert re.search(r"pub fn %s\(" % sym, SYS), sym
    assert "RTMI_ABI_VERSION: u32 = %d" % abi.RTMI_ABI_VERSION in SYS
