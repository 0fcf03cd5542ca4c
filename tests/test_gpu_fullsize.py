"""BASELINE.json's FULL sizes on the device, through size-independent properties (the oracle cannot
render 10^9 paths):
  * the pruned + cooperative traversal equals the exact traversal bit-for-bit — radiance, quantised
    image and per-pixel path signatures — on C3 cornell_box 800x800x1000 (lit, 640 M paths), on
    C2 random_spheres 1200x800x500 and on C5 final_scene 1920x1080 (signatures pin the paths of the
    all-black image);
  * the result does not depend on how the sample range is chunked or the image is tiled;
  * the reference-faithful scenes stay black at full size (the reference's own golden property)."""
import numpy as np
import pytest

from raytracing_rust_amd import abi, scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,nx,ny,ns", [("cornell_box", 800, 800, 1000), ("random_spheres", 1200, 800, 500),
                                           ("cornell_smoke", 800, 800, 1000), ("final_scene", 1920, 1080, 1000)])
def test_fast_cooperative_equals_exact_at_full_size(host, name, nx, ny, ns):
    cam, world = scenes.build(host, name, nx, ny, seed=1)
    sc = host.lower(world).upload(0)
    fast = sc.render(cam, nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_FAST_CULL, sig=True)
    exact = sc.render(cam, nx, ny, ns, seed=42, flags=0, sig=True)
    print(name, "fast %.0f ms, exact %.0f ms, paths %.3g, mean radiance %.4f"
          % (fast["stats"]["render_ms"], exact["stats"]["render_ms"], nx * ny * ns, float(fast["linear"].mean())))
    assert np.array_equal(fast["sig"], exact["sig"])
    assert np.array_equal(fast["linear"], exact["linear"])
    assert np.array_equal(fast["rgb8"], exact["rgb8"])
    if name != "cornell_box":
        assert fast["rgb8"].max() == 0  # reference-faithful: no reachable emitter
        assert int(np.count_nonzero(fast["sig"])) > 0.5 * nx * ny  # ... but the paths themselves are pinned (0 = a camera ray that misses everything)
    else:
        assert 0.05 < float(fast["linear"].mean()) < 1.0


def test_chunking_invariance_full_resolution(host):
    nx, ny, ns = 1920, 1080, 48
    cam, world = scenes.build(host, "final_scene", nx, ny, seed=1)
    sc = host.lower(world).upload(0)
    a = sc.render(cam, nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_FAST_CULL, sig=True, spp_chunks=1)
    b = sc.render(cam, nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_FAST_CULL, sig=True, spp_chunks=6)
    assert np.array_equal(a["sig"], b["sig"]) and np.array_equal(a["rgb8"], b["rgb8"])


@pytest.mark.parametrize("rank", [0, 5])
def test_c5_per_rank_workload_of_the_8_gpu_config(host, rank):
    """BASELINE config C5: final_scene 1920x1080x5000spp tile-split over 8 GPUs.  One rank's share (every 8th tile,
    5000 spp: 1.3 G paths, 15.6 GB per-sample buffer) rendered on this GPU exactly as rank `rank` of 8 would:
    pruned/cooperative == exact bit-for-bit in radiance, quantised texels and path signatures."""
    import torch

    from raytracing_rust_amd import dist as rdist

    nx, ny, ns, world = 1920, 1080, 5000, 8
    cam, world_obj = scenes.build(host, "final_scene", nx, ny, seed=1)
    sc = host.lower(world_obj).upload(0)
    dev = torch.device("cuda", 0)
    out = {}
    for label, flags in (("fast", abi.RTMI_FLAG_FAST_CULL), ("exact", 0)):
        p = rdist.rank_params(nx, ny, ns, rank, world, seed=42, flags=flags | abi.RTMI_FLAG_PATH_SIG)
        local = rdist.new_local_framebuffer(p, dev)
        sig = torch.zeros(sc.local_tiles(p) * 64, dtype=torch.int64, device=dev)
        p.path_sig = sig.data_ptr()
        st = sc.render_device(cam, p, local.data_ptr(), torch.cuda.current_stream().cuda_stream, want_stats=True)
        torch.cuda.synchronize()
        out[label] = (local.cpu().numpy().view(np.uint32), sig.cpu().numpy(), st)
        print(label, "rank %d/%d: %.0f ms, %.3g paths" % (rank, world, st["render_ms"], st["samples"]))
    assert out["fast"][2]["samples"] == out["exact"][2]["samples"] > 1.29e9
    assert np.array_equal(out["fast"][1], out["exact"][1])
    assert np.array_equal(out["fast"][0], out["exact"][0])
    assert int(np.count_nonzero(out["fast"][1])) > 0.5 * sc.local_tiles(p) * 64
    assert int((out["fast"][0][:, 3] & 0xffffff).max()) == 0  # reference-faithful: black
