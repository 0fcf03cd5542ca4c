// rtmi_device.hip — gfx950 (MI355X, CDNA4) device path of the per-pixel render loop.
//
// Persistent wavefronts (256 CUs x 16) take units = (8x8 pixel tile, 16 samples) from a global counter; the
// 64 lanes of a wavefront take (sample, pixel) items of its unit dynamically: a lane whose path ended
// starts the next item at once, whatever its pixel, so all lanes stay busy although path lengths differ
// (1..51 hit queries, src/color.rs:6-23).  Every finished path stores its radiance in a per-sample
// buffer in HBM; a resolve kernel adds the samples of a pixel in sample order (tests/test.rs:65-70).
// The recursion of `color` is unrolled into the throughput form L += T*emitted; T *= attenuation.
// Random numbers are Philox4x32-10 counter streams keyed per (pixel, sample), so the result is
// independent of scheduling, tiling, unit size and the number of GPUs.  BVH traversal is cooperative:
// the lanes are workers on a wave-shared LIFO of (ray, node) entries in LDS (rtmi_bvh_coop.hpp).
// No MFMA: there is no dense contraction on this path.
//
// Arithmetic follows the fp32 contract of DESIGN.md: op order as written here, no FMA
// contraction (-ffp-contract=off), IEEE / and sqrt, transcendental functions from
// include/rtmi_math.h.  Every device function cites the reference lines it implements.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <time.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "rtmi.h"
#include "rtmi_math.h"

#include "rtmi_kernels.hpp"

// ======================================================================================
// host side of the C ABI
// ======================================================================================
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return fail(RTMI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
    } while (0)

struct rtmi_scene {
    int device = 0;
    DevScene dev{};
    std::vector<void *> allocs;
    rtmi_scene_desc meta{}; // counts only (pointers nulled)
    double *partial = nullptr; // f64 radiance sums [local tile][3][64], carried between passes
    size_t partial_bytes = 0;
    Rad3 *samples = nullptr; // per-sample radiance [local tile][pass samples][64], 12-B slots
    size_t samples_bytes = 0;
    uint2 *spill = nullptr;    // global part of the cooperative traversal stacks [wavefront slot][spill_cap]
    size_t spill_bytes = 0;
    unsigned int *status = nullptr; // RTMI_STATUS_WORDS device words, see rtmi_types.hpp
    int slots = 0;                  // CUs x 16: resident wavefronts the render kernels are launched with
    bool has_alt = false;           // some BVH item carries an alternative tree
    bool all_alt = false;           // every BVH item does (and there is one): the workgroup-cooperative kernel can run
    bool needs_insd = false;        // DEFERRED items, list scans, nested media: the INSTL = 2 instantiations (rtmi_kernels.hpp)
    bool has_deferred = false;      // media that were children of a BVHNode (RTMI_ITEMFLAG_DEFERRED): the asynchronous
                                    // state-machine kernel does not carry them, RTMI_FLAG_ASYNC then runs the per-lane kernel
    uint32_t last_kernel = 0;       // RTMI_KERNEL_* of the last render enqueued on this handle (rtmi_stats.kernel)
    // scratch of the blocking host API (grow-only, so a host that renders frame after frame allocates once)
    rtmi_texel *texels = nullptr;
    size_t texel_count = 0;
    unsigned long long *d_sig = nullptr;
    size_t sig_count = 0;
    rtmi_texel *h_texels = nullptr;    // pinned host mirror of `texels` (hipHostMalloc: the D2H copy runs at link speed)
    size_t h_texel_count = 0;
    rtmi_texel *h_partial = nullptr;   // ... and of the partial images rtmi_partial_image fetches (RTMI_FLAG_PROGRESSIVE)
    size_t h_partial_count = 0;
    std::vector<unsigned long long> h_sig;
    // Thread model (rtmi.h): render calls on one handle serialise.  `mu` orders the host side (planning, scratch
    // (re)allocation, enqueue, and for the blocking calls the wait and the copy-out); `busy` chains the device side: a
    // render enqueued on ANY stream first waits for the previous render of this handle, whose kernels use the same
    // unit queue, status words and per-sample buffer.
    std::mutex mu;
    hipEvent_t busy = nullptr;
    bool busy_recorded = false;
    hipStream_t stream = nullptr;      // launches of the blocking API (non-blocking stream)
    hipStream_t copy_stream = nullptr; // progress polls while a launch runs
    uint64_t units_total = 0;          // work units of the last enqueued call (progress denominator)
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
};

extern "C" const char *rtmi_last_error(void) { return g_err.c_str(); }

#ifndef RTMI_BUILD_HASH
#define RTMI_BUILD_HASH "unknown"
#endif
extern "C" const char *rtmi_build_hash(void) { return RTMI_BUILD_HASH; }

// Per-sample buffers outlive their handle, one per device: a destroyed handle parks its buffer here and the next
// handle on that device takes it over when it is large enough.  Why: hipMalloc of tens of GB right after a hipFree of
// the same size stalls for SECONDS now and then on this stack (profiles/r04_experiments/first_call_probe.log: 0.3 ms
// nine times, then 4.6 s and 5.0 s for the same 25 GB; the r03 bench record shows one such stall as a 1.3 s first
// call), and a host that renders one image per handle (rtmi_render_multi, Camera::render — the reference's usage
// model is one render per process, tests/test.rs:802-838) would meet it again and again.  At most one buffer per device
// is parked; rtmi_release_cached() returns the memory.
namespace {
struct ParkedSamples { void *ptr = nullptr; size_t bytes = 0; };
std::mutex g_parked_mu;
std::map<int, ParkedSamples> g_parked;
// takes the parked buffer of `device` if it holds at least `need` bytes
bool parked_take(int device, size_t need, void **ptr, size_t *bytes) {
    std::lock_guard<std::mutex> lock(g_parked_mu);
    auto it = g_parked.find(device);
    if (it == g_parked.end() || !it->second.ptr || it->second.bytes < need) return false;
    *ptr = it->second.ptr; *bytes = it->second.bytes;
    g_parked.erase(it);
    return true;
}
size_t parked_bytes(int device) {
    std::lock_guard<std::mutex> lock(g_parked_mu);
    auto it = g_parked.find(device);
    return it == g_parked.end() ? 0 : it->second.bytes;
}
// frees what is parked on `device` (the current HIP device must be `device`)
void parked_drop(int device) {
    std::lock_guard<std::mutex> lock(g_parked_mu);
    auto it = g_parked.find(device);
    if (it == g_parked.end()) return;
    if (it->second.ptr) (void)hipFree(it->second.ptr);
    g_parked.erase(it);
}
// parks `ptr` (no kernel uses it any more) unless a larger one is parked already; the loser is freed
void parked_give(int device, void *ptr, size_t bytes) {
    std::lock_guard<std::mutex> lock(g_parked_mu);
    ParkedSamples &slot = g_parked[device];
    if (slot.ptr && slot.bytes >= bytes) { (void)hipFree(ptr); return; }
    if (slot.ptr) (void)hipFree(slot.ptr);
    slot.ptr = ptr; slot.bytes = bytes;
}
} // namespace

extern "C" void rtmi_release_cached(void) {
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    std::lock_guard<std::mutex> lock(g_parked_mu);
    for (auto &kv : g_parked)
        if (kv.second.ptr && hipSetDevice(kv.first) == hipSuccess) (void)hipFree(kv.second.ptr);
    g_parked.clear();
    if (have_cur) (void)hipSetDevice(cur);
}

extern "C" int rtmi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

template <typename T>
static int upload(rtmi_scene *s, const T *src, size_t n, const T **dst) {
    *dst = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    if (n) HIP_TRY(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = reinterpret_cast<const T *>(p);
    return RTMI_OK;
}

static int validate(const rtmi_scene_desc *d) {
    if (!d) return fail(RTMI_ERR_INVALID, "desc is NULL");
    if (d->abi_version != RTMI_ABI_VERSION) return fail(RTMI_ERR_INVALID, "abi_version mismatch");
    if (d->n_items == 0 || !d->items) return fail(RTMI_ERR_INVALID, "scene has no items");
    if (d->max_bvh_depth > RTMI_MAX_BVH_DEPTH)
        return fail(RTMI_ERR_UNSUPPORTED, "BVH deeper than RTMI_MAX_BVH_DEPTH");
    if (d->n_prims >= (1u << 28)) return fail(RTMI_ERR_UNSUPPORTED, "too many primitives");
    if ((d->n_prims && (!d->prim_a || !d->prim_b || !d->prim_meta)) || (d->n_nodes && !d->nodes) || (d->n_xforms && !d->xforms) ||
        (d->n_materials && !d->materials) || (d->n_textures && !d->textures) || (d->n_perlin && !d->perlin) ||
        (d->n_images && !d->images) || (d->image_bytes && !d->image_data))
        return fail(RTMI_ERR_INVALID, "a non-zero count comes with a NULL array");
    auto prim_ok = [&](int64_t i) { return i >= 0 && (uint64_t)i < d->n_prims; };
    for (uint32_t i = 0; i < d->n_prims; i++) {
        const rtmi_prim_meta &m = d->prim_meta[i];
        if (m.type < 0 || m.type > RTMI_PRIM_CUBE) return fail(RTMI_ERR_INVALID, "bad primitive type");
        if (m.material < 0 || (uint32_t)m.material >= d->n_materials)
            return fail(RTMI_ERR_INVALID, "primitive material out of range");
        const uint32_t xfc = (m.flags >> RTMI_PRIMFLAG_XF_COUNT_SHIFT) & RTMI_PRIM_XF_MAX, xff = m.flags >> RTMI_PRIMFLAG_XF_FIRST_SHIFT;
        if (xfc != 0u && ((uint64_t)xff + xfc > d->n_xforms)) return fail(RTMI_ERR_INVALID, "primitive transform range out of bounds");
        for (uint32_t k = 0; k < xfc; k++) // (gate and inner-medium records are no transforms: never inside a chain)
            if (d->xforms[xff + k].kind > RTMI_XF_ROTATE_Z) return fail(RTMI_ERR_INVALID, "a gate or inner-medium record inside a primitive's transform chain");
    }
    for (uint32_t i = 0; i < d->n_xforms; i++)
        if (d->xforms[i].kind < RTMI_XF_TRANSLATE || d->xforms[i].kind > RTMI_XF_INNER_MEDIUM) return fail(RTMI_ERR_INVALID, "bad transform kind");
    for (uint32_t i = 0; i < d->n_nodes; i++) {
        const int32_t ch[2] = {d->nodes[i].left, d->nodes[i].right};
        for (int c = 0; c < 2; c++) {
            if (ch[c] >= 0) {
                if ((uint32_t)ch[c] >= d->n_nodes) return fail(RTMI_ERR_INVALID, "BVH child out of range");
            } else {
                uint32_t type = ((uint32_t)ch[c] >> 28) & 7u, idx = (uint32_t)ch[c] & 0x0fffffffu;
                if (type > RTMI_PRIM_CUBE || !prim_ok(idx)) return fail(RTMI_ERR_INVALID, "BVH leaf out of range");
                if ((int)type != d->prim_meta[idx].type) return fail(RTMI_ERR_INVALID, "BVH leaf type mismatch");
            }
        }
    }
    // alternative-tree records: children in range, leaves typed like the primitive they name
    if (d->n_alt_nodes && !d->alt_nodes) return fail(RTMI_ERR_INVALID, "n_alt_nodes without alt_nodes");
    for (uint32_t i = 0; i < d->n_alt_nodes; i++)
        for (int c = 0; c < 4; c++) {
            const int32_t ch = d->alt_nodes[i].child[c];
            if (ch == RTMI_NO_CHILD) continue;
            if (ch >= 0) {
                if ((uint32_t)ch >= d->n_alt_nodes) return fail(RTMI_ERR_INVALID, "alternative tree child out of range");
            } else {
                const uint32_t type = ((uint32_t)ch >> 28) & 7u, idx = (uint32_t)ch & 0x0fffffffu;
                if (type > RTMI_PRIM_CUBE || !prim_ok(idx)) return fail(RTMI_ERR_INVALID, "alternative tree leaf out of range");
                if ((int)type != d->prim_meta[idx].type) return fail(RTMI_ERR_INVALID, "alternative tree leaf type mismatch");
            }
        }
    // Walk every tree from its item's root.  The kernels size their traversal stacks from the DECLARED depths
    // (fixed RTMI_MAX_BVH_DEPTH-entry LDS stack per lane; pool spill capacity), and a persistent wavefront that
    // follows a child cycle never ends: each node may be reached once (no cycles, no shared subtrees) and no
    // deeper than declared.  Depth of a root = 1, as the lowering counts it.
    std::vector<uint8_t> seen_bin(d->n_nodes, 0), seen_alt(d->n_alt_nodes, 0);
    std::vector<std::pair<uint32_t, uint32_t>> todo; // (node, depth)
    const auto walk = [&](bool alt, uint32_t root, uint32_t declared, const char *what) -> int {
        std::vector<uint8_t> &seen = alt ? seen_alt : seen_bin;
        todo.clear();
        todo.emplace_back(root, 1u);
        while (!todo.empty()) {
            const uint32_t n = todo.back().first, depth = todo.back().second;
            todo.pop_back();
            if (seen[n]) return fail(RTMI_ERR_INVALID, std::string(what) + ": a node is reached twice (cycle or shared subtree)");
            seen[n] = 1;
            if (depth > declared) return fail(RTMI_ERR_INVALID, std::string(what) + " is deeper than the declared depth");
            if (alt) {
                for (int c = 0; c < 4; c++) {
                    const int32_t ch = d->alt_nodes[n].child[c];
                    if (ch >= 0 && ch != RTMI_NO_CHILD) todo.emplace_back((uint32_t)ch, depth + 1u);
                }
            } else {
                const int32_t l = d->nodes[n].left, r = d->nodes[n].right;
                if (l >= 0) todo.emplace_back((uint32_t)l, depth + 1u);
                // right == left is legal: BVHNode::new over ONE element stores it on both sides (bvh.rs:44-45); the
                // kernels visit it once.  Any other repeated reference is caught by `seen`.
                if (r >= 0 && r != l) todo.emplace_back((uint32_t)r, depth + 1u);
            }
        }
        return RTMI_OK;
    };
    bool scan_open = false;
    for (uint32_t i = 0; i < d->n_items; i++) {
        const rtmi_item &it = d->items[i];
        if (it.kind == RTMI_ITEM_LIST) {
            if (it.count < 0) return fail(RTMI_ERR_INVALID, "item primitive count is negative");
            if (it.count > 0 && (!prim_ok(it.first) || !prim_ok((int64_t)it.first + it.count - 1)))
                return fail(RTMI_ERR_INVALID, "item primitive range out of bounds");
        } else if (it.kind == RTMI_ITEM_BVH) {
            if (it.first < 0 || (uint32_t)it.first >= d->n_nodes) return fail(RTMI_ERR_INVALID, "item BVH root out of range");
            if (int rc = walk(false, (uint32_t)it.first, d->max_bvh_depth, "BVH")) return rc;
            if (it.alt_first >= 0) {
                if ((uint32_t)it.alt_first >= d->n_alt_nodes || !d->prim_gate || !d->alt_nodes)
                    return fail(RTMI_ERR_INVALID, "item alternative tree out of range or prim_gate missing");
                if (d->alt_max_depth > 64u) return fail(RTMI_ERR_UNSUPPORTED, "alternative tree deeper than 64");
                if (int rc = walk(true, (uint32_t)it.alt_first, d->alt_max_depth, "alternative tree")) return rc;
            }
        } else {
            return fail(RTMI_ERR_INVALID, "bad item kind");
        }
        if (it.xform_count < 0 || it.xform_first < 0 || (uint32_t)(it.xform_first + it.xform_count) > d->n_xforms)
            return fail(RTMI_ERR_INVALID, "item transform range out of bounds");
        if ((it.flags & RTMI_ITEMFLAG_MEDIUM) &&
            (it.medium_material < 0 || (uint32_t)it.medium_material >= d->n_materials))
            return fail(RTMI_ERR_INVALID, "medium material out of range");
        if ((int32_t)((it.flags >> RTMI_ITEMFLAG_MEDIUM_OUTER_SHIFT) & 15u) > it.xform_count)
            return fail(RTMI_ERR_INVALID, "more outer medium transforms than the item has transforms");
        for (int32_t k = 0; k < it.xform_count; k++) // (the gate records are no transforms: never inside a chain)
            if (d->xforms[it.xform_first + k].kind > RTMI_XF_ROTATE_Z) return fail(RTMI_ERR_INVALID, "a gate record inside an item's transform chain");
        if (it.flags & RTMI_ITEMFLAG_NESTED_MEDIUM) { // a medium whose boundary is a medium: the inner density behind the chain (rtmi.h)
            const uint32_t at = (uint32_t)(it.xform_first + it.xform_count) + (((it.flags & RTMI_ITEMFLAG_DEFERRED) && it.kind == RTMI_ITEM_BVH) ? 2u : 0u);
            if (!(it.flags & RTMI_ITEMFLAG_MEDIUM) || at >= d->n_xforms || d->xforms[at].kind != RTMI_XF_INNER_MEDIUM)
                return fail(RTMI_ERR_INVALID, "a NESTED_MEDIUM item must be a MEDIUM with its RTMI_XF_INNER_MEDIUM record behind its transform chain");
        }
        { // list scans that were children of a BVHNode (rtmi.h): BEGIN on the first member, members only inside, one terminator
            const uint32_t ls = it.flags & (RTMI_ITEMFLAG_LISTSCAN_BEGIN | RTMI_ITEMFLAG_LISTSCAN_MEMBER | RTMI_ITEMFLAG_LISTSCAN_END);
            if (ls && !(it.flags & RTMI_ITEMFLAG_DEFERRED)) return fail(RTMI_ERR_INVALID, "a LISTSCAN item must be DEFERRED");
            if (ls & RTMI_ITEMFLAG_LISTSCAN_END) {
                if (ls != RTMI_ITEMFLAG_LISTSCAN_END || !scan_open || it.kind != RTMI_ITEM_LIST || it.count != 0 || it.first < 0 ||
                    (it.flags & (RTMI_ITEMFLAG_MEDIUM | RTMI_ITEMFLAG_SAVE_T0)))
                    return fail(RTMI_ERR_INVALID, "the terminator of a list scan is a LIST item of no primitives behind its members, `first` = its position in the tree");
                scan_open = false;
                continue;
            }
            if (ls & RTMI_ITEMFLAG_LISTSCAN_BEGIN) {
                if (scan_open || !(ls & RTMI_ITEMFLAG_LISTSCAN_MEMBER)) return fail(RTMI_ERR_INVALID, "LISTSCAN_BEGIN inside an open list scan, or not on a member");
                scan_open = true;
            }
            if (scan_open != ((ls & RTMI_ITEMFLAG_LISTSCAN_MEMBER) != 0u))
                return fail(RTMI_ERR_INVALID, "every item between LISTSCAN_BEGIN and the terminator is a LISTSCAN_MEMBER, and no other is");
            if ((ls & RTMI_ITEMFLAG_LISTSCAN_MEMBER) && it.kind == RTMI_ITEM_BVH && !(it.flags & RTMI_ITEMFLAG_MEDIUM))
                return fail(RTMI_ERR_UNSUPPORTED, "a BVH as a member of a list scan is supported as a medium's boundary only");
        }
        if (it.flags & RTMI_ITEMFLAG_DEFERRED) { // a medium or an instanced subtree that was a child of a BVHNode (rtmi.h)
            const int32_t G = (int32_t)((it.flags >> RTMI_ITEMFLAG_GATE_OUTER_SHIFT) & 15u);
            if (it.kind == RTMI_ITEM_LIST) {
                if (!(it.flags & (RTMI_ITEMFLAG_MEDIUM | RTMI_ITEMFLAG_LISTSCAN_MEMBER)) || it.count < 1 || !d->prim_gate)
                    return fail(RTMI_ERR_INVALID, "a DEFERRED item of kind LIST must be a MEDIUM or a member of a list scan, with at least one primitive, and prim_gate must be given");
            } else {
                if (G > it.xform_count || (uint32_t)(it.xform_first + it.xform_count) + 2u > d->n_xforms ||
                    d->xforms[it.xform_first + it.xform_count].kind != RTMI_XF_GATE_MIN || d->xforms[it.xform_first + it.xform_count + 1].kind != RTMI_XF_GATE_MAX)
                    return fail(RTMI_ERR_INVALID, "a DEFERRED item of kind BVH needs its two gate records behind its transform chain");
            }
            if ((it.flags & RTMI_ITEMFLAG_MEDIUM) && G > (int32_t)((it.flags >> RTMI_ITEMFLAG_MEDIUM_OUTER_SHIFT) & 15u))
                return fail(RTMI_ERR_INVALID, "a DEFERRED medium's enclosing transforms must be among those that wrap the medium");
        }
    }
    if (scan_open) return fail(RTMI_ERR_INVALID, "a list scan without its terminator");
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rtmi_material &m = d->materials[i];
        if (m.kind < 0 || m.kind > RTMI_MAT_ISOTROPIC) return fail(RTMI_ERR_INVALID, "bad material kind");
        if (m.kind != RTMI_MAT_DIELECTRIC && (m.tex < 0 || (uint32_t)m.tex >= d->n_textures))
            return fail(RTMI_ERR_INVALID, "material texture out of range");
    }
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const rtmi_texture &t = d->textures[i];
        switch (t.kind) {
        case RTMI_TEX_SOLID: break;
        case RTMI_TEX_CHECKER:
            if (t.i0 < 0 || t.i1 < 0 || (uint32_t)t.i0 >= d->n_textures || (uint32_t)t.i1 >= d->n_textures)
                return fail(RTMI_ERR_INVALID, "checker child out of range");
            break;
        case RTMI_TEX_NOISE:
            if (t.i0 < 0 || (uint32_t)t.i0 >= d->n_perlin) return fail(RTMI_ERR_INVALID, "perlin table out of range");
            break;
        case RTMI_TEX_IMAGE:
            if (t.i0 < 0 || (uint32_t)t.i0 >= d->n_images) return fail(RTMI_ERR_INVALID, "image out of range");
            break;
        default: return fail(RTMI_ERR_INVALID, "bad texture kind");
        }
    }
    { // checkers nest (texture.rs:28-48 is generic over its children); the device follows at most 16 levels, and a checker
      // that reaches itself would never end: longest checker chain below every texture, by 17 rounds of relaxation
        std::vector<uint8_t> depth(d->n_textures, 0);
        for (int round = 0; round <= 16; round++)
            for (uint32_t i = 0; i < d->n_textures; i++) {
                const rtmi_texture &t = d->textures[i];
                if (t.kind != RTMI_TEX_CHECKER) continue;
                const uint8_t below = depth[t.i0] > depth[t.i1] ? depth[t.i0] : depth[t.i1];
                depth[i] = (uint8_t)(below + 1);
                if (depth[i] > 16) return fail(RTMI_ERR_INVALID, "checker textures nested deeper than 16 levels, or a checker that reaches itself");
            }
    }
    for (uint32_t i = 0; i < d->n_images; i++) {
        const rtmi_image &im = d->images[i];
        if (im.nx == 0 || im.ny == 0 || im.offset + 3ull * im.nx * im.ny > d->image_bytes)
            return fail(RTMI_ERR_INVALID, "image outside image_data");
    }
    return RTMI_OK;
}

extern "C" int rtmi_scene_create(const rtmi_scene_desc *d, int device, rtmi_scene **out) {
    if (!out) return fail(RTMI_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int rc = validate(d);
    if (rc) return rc;
    int ndev = rtmi_device_count();
    if (ndev <= 0) return fail(RTMI_ERR_DEVICE, "no HIP device available (the rtmi path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(RTMI_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    rtmi_scene *s = new (std::nothrow) rtmi_scene();
    if (!s) return fail(RTMI_ERR_NOMEM, "out of host memory");
    s->device = device;
    s->meta = *d;
    const float4 *nodes4 = nullptr;
    {
        // device copy of the items: a ConstantMedium whose boundary is one static sphere (the common case:
        // tests/test.rs:471-483) carries that sphere in its unused root-box words, marked by a device-only flag bit,
        // so the kernel's fused two-root boundary query needs no dependent loads of the primitive's meta and planes
        std::vector<rtmi_item> items(d->items, d->items + d->n_items);
        for (rtmi_item &it : items) {
            it.flags &= (RTMI_ITEMFLAG_FLIP | RTMI_ITEMFLAG_MEDIUM | (15u << RTMI_ITEMFLAG_MEDIUM_OUTER_SHIFT) | RTMI_ITEMFLAG_SAVE_T0 |
                         RTMI_ITEMFLAG_DEFERRED | (15u << RTMI_ITEMFLAG_GATE_OUTER_SHIFT) | RTMI_ITEMFLAG_NESTED_MEDIUM |
                         RTMI_ITEMFLAG_LISTSCAN_BEGIN | RTMI_ITEMFLAG_LISTSCAN_MEMBER | RTMI_ITEMFLAG_LISTSCAN_END);
            if ((it.flags & RTMI_ITEMFLAG_MEDIUM) && !(it.flags & (RTMI_ITEMFLAG_DEFERRED | RTMI_ITEMFLAG_NESTED_MEDIUM)) && it.kind == RTMI_ITEM_LIST && it.count == 1 &&
                d->prim_meta[it.first].type == RTMI_PRIM_SPHERE &&
                ((d->prim_meta[it.first].flags >> RTMI_PRIMFLAG_XF_COUNT_SHIFT) & RTMI_PRIM_XF_MAX) == 0u) {
                // (a sphere with a transform chain of its own — ConstantMedium(HittableList[Traslate(Sphere)]) — takes the
                // general boundary query, which applies the chain; the fused query reads the raw centre)
                it.flags |= RTMI_ITEMFLAG_DEV_MEDIUM_SPHERE;
                memcpy(it.root_min, d->prim_a + (size_t)it.first * 4, 3 * sizeof(float));
                it.root_max[0] = d->prim_a[(size_t)it.first * 4 + 3];
            }
        }
        std::vector<DevItem> ditems(d->n_items);
        for (uint32_t i = 0; i < d->n_items; i++) {
            DevItem &D = ditems[i];
            memset(&D, 0, sizeof(D));
            D.it = items[i];
            if (items[i].xform_count > 0) D.x0 = d->xforms[items[i].xform_first];     // ranges checked by validate()
            if (items[i].xform_count > 1) D.x1 = d->xforms[items[i].xform_first + 1];
        }
        rc = upload(s, ditems.data(), d->n_items, &s->dev.items);
    }
    if (!rc) rc = upload(s, reinterpret_cast<const float4 *>(d->prim_a), d->n_prims, &s->dev.prim_a);
    if (!rc) rc = upload(s, reinterpret_cast<const float4 *>(d->prim_b), d->n_prims, &s->dev.prim_b);
    if (!rc) rc = upload(s, d->prim_meta, d->n_prims, &s->dev.meta);
    if (!rc && d->prim_gate) rc = upload(s, reinterpret_cast<const float4 *>(d->prim_gate), (size_t)d->n_prims * 2, &s->dev.gate);
    if (!rc) {
        // device copy of the nodes: the reserved words carry the child references in the 26-bit encoding of the
        // cooperative traversal's work pool (rtmi_bvh_coop.hpp), so a node visit does not re-encode them
        std::vector<rtmi_bvh_node> nodes(d->nodes, d->nodes + d->n_nodes);
        const auto enc = [](int32_t ref) -> int32_t {
            if (ref >= 0) return ref;
            const uint32_t u = (uint32_t)ref;
            return (int32_t)((1u << 25) | (((u >> 28) & 7u) << 22) | (u & 0x003fffffu));
        };
        for (rtmi_bvh_node &n : nodes) { n.pad[0] = enc(n.left); n.pad[1] = enc(n.right); }
        rc = upload(s, reinterpret_cast<const float4 *>(nodes.data()), (size_t)d->n_nodes * 4, &nodes4);
        if (!rc && d->n_alt_nodes && d->alt_nodes) { // 4-wide alternative trees: children stored pool-encoded
            std::vector<rtmi_bvh4_node> alt(d->alt_nodes, d->alt_nodes + d->n_alt_nodes);
            for (rtmi_bvh4_node &n : alt)
                for (int c = 0; c < 4; c++) {
                    if (n.child[c] == RTMI_NO_CHILD) { // empty slot: a box no ray can enter, whatever the host wrote
                        n.child[c] = (int32_t)0xffffffffu;
                        const float big = 3.40282346638528859811704183484516925e+38f;
                        n.minx[c] = n.miny[c] = n.minz[c] = big;
                        n.maxx[c] = n.maxy[c] = n.maxz[c] = -big;
                    }
                    else n.child[c] = enc(n.child[c]); // range-checked by validate()
                }
            if (!rc) rc = upload(s, reinterpret_cast<const float4 *>(alt.data()), (size_t)d->n_alt_nodes * 8, &s->dev.nodes4);
        }
    }
    if (!rc) { // leaf records: see DevScene
        std::vector<float> lr((size_t)d->n_prims * 20, 0.0f);
        for (uint32_t i = 0; i < d->n_prims; i++) {
            float *r = &lr[(size_t)i * 20];
            memcpy(r, d->prim_a + (size_t)i * 4, 16);
            memcpy(r + 4, d->prim_b + (size_t)i * 4, 16);
            memcpy(r + 8, &d->prim_meta[i], 16);
            if (d->prim_gate) memcpy(r + 12, d->prim_gate + (size_t)i * 8, 32);
        }
        rc = upload(s, reinterpret_cast<const float4 *>(lr.data()), (size_t)d->n_prims * 5, &s->dev.leaf_rec);
    }
    if (!rc) { // shading records: see DevScene
        const auto fill = [&](float *r, int32_t material) {
            const rtmi_material &m = d->materials[material];
            memcpy(r + 4, &m, 16);
            if (m.kind != RTMI_MAT_DIELECTRIC) memcpy(r + 8, &d->textures[m.tex], 32);
        };
        std::vector<float> sp((size_t)d->n_prims * 16, 0.0f), sm((size_t)d->n_materials * 16, 0.0f);
        for (uint32_t i = 0; i < d->n_prims; i++) {
            memcpy(&sp[(size_t)i * 16], d->prim_a + (size_t)i * 4, 16);
            fill(&sp[(size_t)i * 16], d->prim_meta[i].material);
        }
        for (uint32_t i = 0; i < d->n_materials; i++) fill(&sm[(size_t)i * 16], (int32_t)i);
        rc = upload(s, reinterpret_cast<const float4 *>(sp.data()), (size_t)d->n_prims * 4, &s->dev.shade_prim);
        if (!rc) rc = upload(s, reinterpret_cast<const float4 *>(sm.data()), (size_t)d->n_materials * 4, &s->dev.shade_mat);
    }
    if (!rc) rc = upload(s, d->xforms, d->n_xforms, &s->dev.xforms);
    if (!rc) rc = upload(s, d->materials, d->n_materials, &s->dev.mats);
    if (!rc) rc = upload(s, d->textures, d->n_textures, &s->dev.texs);
    if (!rc) rc = upload(s, d->perlin, d->n_perlin, &s->dev.perlin);
    if (!rc) rc = upload(s, d->images, d->n_images, &s->dev.images);
    if (!rc) rc = upload(s, d->image_data, (size_t)d->image_bytes, &s->dev.image_data);
    if (rc) {
        rtmi_scene_destroy(s);
        return rc;
    }
    s->dev.nodes = nodes4;
    for (uint32_t i = 0; i < d->n_items; i++)
        if (d->items[i].kind == RTMI_ITEM_BVH && d->items[i].alt_first >= 0) s->has_alt = true;
    s->all_alt = s->has_alt;
    for (uint32_t i = 0; i < d->n_items; i++)
        if (d->items[i].kind == RTMI_ITEM_BVH && d->items[i].alt_first < 0) s->all_alt = false;
    s->dev.n_items = d->n_items;
    s->dev.has_prim_xf = 0u;
    for (uint32_t i = 0; i < d->n_prims; i++)
        if ((d->prim_meta[i].flags >> RTMI_PRIMFLAG_XF_COUNT_SHIFT) & RTMI_PRIM_XF_MAX) s->dev.has_prim_xf = 1u;
    s->dev.has_medium_outer = 0u;
    for (uint32_t i = 0; i < d->n_items; i++)
        if (((d->items[i].flags >> RTMI_ITEMFLAG_MEDIUM_OUTER_SHIFT) & 15u) || (d->items[i].flags & (RTMI_ITEMFLAG_SAVE_T0 | RTMI_ITEMFLAG_DEFERRED | RTMI_ITEMFLAG_NESTED_MEDIUM)))
            s->dev.has_medium_outer = 1u; // (deferred media ride in the same instantiations as media inside transforms)
    for (uint32_t i = 0; i < d->n_items; i++)
        if (d->items[i].flags & (RTMI_ITEMFLAG_DEFERRED | RTMI_ITEMFLAG_NESTED_MEDIUM)) s->has_deferred = true;
    for (uint32_t i = 0; i < d->n_items; i++)
        if (d->items[i].flags & (RTMI_ITEMFLAG_SAVE_T0 | RTMI_ITEMFLAG_DEFERRED | RTMI_ITEMFLAG_NESTED_MEDIUM)) s->needs_insd = true;
    if (hipMalloc(reinterpret_cast<void **>(&s->status), RTMI_STATUS_WORDS * sizeof(unsigned int)) != hipSuccess ||
        hipMemset(s->status, 0, RTMI_STATUS_WORDS * sizeof(unsigned int)) != hipSuccess) {
        rtmi_scene_destroy(s);
        return fail(RTMI_ERR_DEVICE, "allocating the status word failed");
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
        s->slots = cus * 4 * 5; // 4 SIMDs x up to 5 wavefronts: capacity of the spill buffer; a launch uses cus * 4 * its waves per SIMD
    }
    for (int i = 0; i < 3; i++)
        if (hipEventCreate(&s->ev[i]) != hipSuccess) {
            rtmi_scene_destroy(s);
            return fail(RTMI_ERR_DEVICE, "hipEventCreate failed");
        }
    if (hipEventCreateWithFlags(&s->busy, hipEventDisableTiming) != hipSuccess) {
        rtmi_scene_destroy(s);
        return fail(RTMI_ERR_DEVICE, "hipEventCreate failed");
    }
    *out = s;
    return RTMI_OK;
}

extern "C" void rtmi_scene_destroy(rtmi_scene *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (void *p : s->allocs) (void)hipFree(p);
    if (s->partial) (void)hipFree(s->partial);
    if (s->samples) { // parked for the next handle on this device (see g_parked); no kernel may still write it
        if (s->busy_recorded) (void)hipEventSynchronize(s->busy);
        parked_give(s->device, s->samples, s->samples_bytes);
    }
    if (s->spill) (void)hipFree(s->spill);
    if (s->texels) (void)hipFree(s->texels);
    if (s->d_sig) (void)hipFree(s->d_sig);
    if (s->status) (void)hipFree(s->status);
    if (s->h_texels) (void)hipHostFree(s->h_texels);
    if (s->h_partial) (void)hipHostFree(s->h_partial);
    if (s->busy) (void)hipEventDestroy(s->busy);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
    for (int i = 0; i < 3; i++)
        if (s->ev[i]) (void)hipEventDestroy(s->ev[i]);
    delete s;
}

static inline uint32_t tiles_x_of(const rtmi_render_params *p) { return (p->nx + RTMI_TILE - 1) / RTMI_TILE; }
static inline uint32_t tiles_y_of(const rtmi_render_params *p) { return (p->ny + RTMI_TILE - 1) / RTMI_TILE; }
static inline uint32_t local_tiles_of(const rtmi_render_params *p, uint32_t rank) {
    const uint32_t T = tiles_x_of(p) * tiles_y_of(p);
    return rank < T ? (T - rank + p->tile_world - 1) / p->tile_world : 0;
}

extern "C" uint32_t rtmi_local_tiles(const rtmi_render_params *p) {
    if (!p || p->tile_world == 0) return 0;
    return local_tiles_of(p, p->tile_rank);
}

static int check_params(const rtmi_render_params *p) {
    if (!p) return fail(RTMI_ERR_INVALID, "params is NULL");
    if (p->nx == 0 || p->ny == 0 || p->ns == 0) return fail(RTMI_ERR_INVALID, "nx, ny and ns must be positive");
    if ((uint64_t)p->nx * p->ny > 0xffffffffull) return fail(RTMI_ERR_UNSUPPORTED, "image too large");
    if (p->ns >= (1u << 26)) return fail(RTMI_ERR_UNSUPPORTED, "ns must be below 2^26");
    if (p->tile_world == 0 || p->tile_rank >= p->tile_world) return fail(RTMI_ERR_INVALID, "bad tile_rank/tile_world");
    return RTMI_OK;
}

// Plan of one render call: unit size, samples per pass; (re)allocates the per-sample buffer and the f64 sums.
static int plan_and_reserve(rtmi_scene *s, const rtmi_render_params *p, uint32_t ntiles_local, uint32_t &chunk_spp,
                            uint32_t &pass_ns) {
    // ---- per-sample buffer and passes.  Every finished path stores its radiance (12 B) in
    // samples[local tile][sample of the pass][pixel]; the resolve kernel adds them in sample order.  With
    // 288 GB of HBM the whole sample range normally fits (headline: 25 GB); otherwise the range is rendered
    // in passes and the f64 sums are carried between them — the same additions in the same order.
    const size_t per_sample = (size_t)ntiles_local * 64 * RTMI_SAMPLE_SLOT_BYTES;
    size_t want = p->sample_buffer_bytes ? (size_t)p->sample_buffer_bytes : ((size_t)45 << 30);
    if (want > ((size_t)45 << 30)) want = (size_t)45 << 30; // slots are addressed with 32 bits (< 2^32 x 12 B = 48 GiB)
    uint64_t max_pass = want / per_sample;
    if (max_pass < 1) max_pass = 1;
    if (max_pass > p->ns) max_pass = p->ns;
    if (max_pass * per_sample > s->samples_bytes) {
        if (!p->sample_buffer_bytes) { // default budget: never more than 3/4 of what is free on the device
            size_t free_b = 0, total_b = 0;
            HIP_TRY(hipMemGetInfo(&free_b, &total_b));
            const size_t avail = (free_b + s->samples_bytes + parked_bytes(s->device)) / 4 * 3;
            if (max_pass * per_sample > avail) max_pass = avail / per_sample ? avail / per_sample : 1;
        }
        if (max_pass * per_sample > s->samples_bytes) {
            const size_t need = max_pass * per_sample;
            void *taken = nullptr;
            size_t taken_bytes = 0;
            if (parked_take(s->device, need, &taken, &taken_bytes)) { // a destroyed handle's buffer: no hipMalloc
                if (s->samples) parked_give(s->device, s->samples, s->samples_bytes); // (smaller than the one taken)
                s->samples = static_cast<Rad3 *>(taken);
                s->samples_bytes = taken_bytes;
            } else {
                if (s->samples) { HIP_TRY(hipFree(s->samples)); s->samples = nullptr; s->samples_bytes = 0; }
                parked_drop(s->device); // too small to serve this call: its memory may be what the allocation needs
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->samples), need));
                s->samples_bytes = need;
            }
        }
    }
    // unit = (tile, chunk of the sample range), the grain of the persistent wavefronts' queue.  Lanes
    // take (sample, pixel) items dynamically and a wavefront moves on to the next unit without
    // draining, so units can be small: the launch ends within about one heavy unit of the last
    // wavefront.  Small images still get a few units per wavefront slot.
    if (p->spp_chunks) {
        chunk_spp = (p->ns + p->spp_chunks - 1) / p->spp_chunks;
    } else {
        chunk_spp = 16u;
        const uint64_t want_units = (uint64_t)s->slots * 4u;
        if ((uint64_t)ntiles_local * ((p->ns + chunk_spp - 1) / chunk_spp) < want_units) {
            const uint64_t per_tile = (want_units + ntiles_local - 1) / ntiles_local;
            chunk_spp = (uint32_t)((p->ns + per_tile - 1) / per_tile);
        }
    }
    if (chunk_spp > max_pass) chunk_spp = (uint32_t)max_pass;
    if (chunk_spp < 1u) chunk_spp = 1u;
    // one pass when everything fits (its last chunk may be shorter); otherwise whole chunks per pass
    pass_ns = max_pass >= p->ns ? p->ns : (uint32_t)(max_pass / chunk_spp) * chunk_spp;

    const size_t need = (size_t)ntiles_local * 64 * 3 * sizeof(double); // f64 sums carried between passes
    if (need > s->partial_bytes) {
        if (s->partial) { HIP_TRY(hipFree(s->partial)); s->partial = nullptr; s->partial_bytes = 0; }
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->partial), need));
        s->partial_bytes = need;
    }
    return RTMI_OK;
}

extern "C" int rtmi_render_prepare(rtmi_scene *s, const rtmi_render_params *p) {
    if (!s) return fail(RTMI_ERR_INVALID, "NULL argument");
    int rc = check_params(p);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(s->mu);
    HIP_TRY(hipSetDevice(s->device));
    const uint32_t ntiles_local = local_tiles_of(p, p->tile_rank);
    if (ntiles_local == 0) return RTMI_OK;
    if (s->busy_recorded) HIP_TRY(hipEventSynchronize(s->busy)); // the buffer may be reallocated: no render may still use it
    uint32_t chunk_spp = 0, pass_ns = 0;
    return plan_and_reserve(s, p, ntiles_local, chunk_spp, pass_ns);
}

// (Re)allocates what a render with these parameters needs BEFORE the caller starts its event clock: an allocation of tens
// of GB can stall the host for seconds (see g_parked), and the kernel_ms of rtmi_stats are kernel time.  The caller holds s->mu.
static int reserve_before_clock(rtmi_scene *s, const rtmi_render_params *p) {
    const uint32_t ntiles_local = local_tiles_of(p, p->tile_rank);
    if (ntiles_local == 0) return RTMI_OK;
    if (s->busy_recorded) HIP_TRY(hipEventSynchronize(s->busy)); // the buffer may be reallocated: no render may still use it
    uint32_t chunk_spp = 0, pass_ns = 0;
    return plan_and_reserve(s, p, ntiles_local, chunk_spp, pass_ns);
}

// the enqueue itself; the caller holds s->mu
static int render_device_locked(rtmi_scene *s, const rtmi_camera *cam, const rtmi_render_params *p, void *d_texels,
                                void *stream_, rtmi_stats *stats) {
    if (!s || !cam || !d_texels) return fail(RTMI_ERR_INVALID, "NULL argument");
    int rc = check_params(p);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    // device side of the thread model: this render starts after the previous one of this handle has finished (a no-op
    // when both were given the same stream)
    if (s->busy_recorded) HIP_TRY(hipStreamWaitEvent(stream, s->busy, 0));
    struct BusyMark { // records the end of this call's work on every return path after the first enqueue
        rtmi_scene *s; hipStream_t st;
        ~BusyMark() { if (hipEventRecord(s->busy, st) == hipSuccess) s->busy_recorded = true; }
    } busy_mark{s, stream};

    DevParams P{};
    P.nx = p->nx; P.ny = p->ny; P.ns = p->ns; P.max_depth = p->max_depth; P.t_min = p->t_min;
    P.key0 = (uint32_t)p->seed; P.key1 = (uint32_t)(p->seed >> 32);
    P.tile_rank = p->tile_rank; P.tile_world = p->tile_world; P.tiles_x = tiles_x_of(p);
    P.ntiles_local = local_tiles_of(p, p->tile_rank);
    if (P.ntiles_local == 0) {
        if (stats) memset(stats, 0, sizeof(*stats));
        return RTMI_OK;
    }
    P.path_sig = reinterpret_cast<unsigned long long *>(p->path_sig);
    if (p->flags & RTMI_FLAG_PATH_SIG) {
        if (!p->path_sig) return fail(RTMI_ERR_INVALID, "RTMI_FLAG_PATH_SIG needs params.path_sig");
        HIP_TRY(hipMemsetAsync(P.path_sig, 0, (size_t)P.ntiles_local * 64 * sizeof(unsigned long long), stream));
    }

    uint32_t chunk_spp = 0, pass_ns = 0;
    rc = plan_and_reserve(s, p, P.ntiles_local, chunk_spp, pass_ns);
    if (rc) return rc;
    P.chunk_spp = chunk_spp;
    P.pass_stride = pass_ns;
    P.samples = s->samples;

    DevCamera C;
    C.origin = F3{cam->origin[0], cam->origin[1], cam->origin[2]};
    C.llc = F3{cam->lower_left_corner[0], cam->lower_left_corner[1], cam->lower_left_corner[2]};
    C.horizontal = F3{cam->horizontal[0], cam->horizontal[1], cam->horizontal[2]};
    C.vertical = F3{cam->vertical[0], cam->vertical[1], cam->vertical[2]};
    C.u = F3{cam->u[0], cam->u[1], cam->u[2]};
    C.v = F3{cam->v[0], cam->v[1], cam->v[2]};
    C.time0 = cam->time0; C.time1 = cam->time1; C.lens_radius = cam->lens_radius;

    if (stats) HIP_TRY(hipEventRecord(s->ev[0], stream));
    // pruned traversal needs the BVH boxes to contain their moving spheres at every ray time (camera.rs:65)
    const float cam_t_lo = cam->time0 < cam->time1 ? cam->time0 : cam->time1, cam_t_hi = cam->time0 < cam->time1 ? cam->time1 : cam->time0;
    const bool boxes_valid = cam_t_lo >= s->meta.bvh_time_lo && cam_t_hi <= s->meta.bvh_time_hi;
    const bool fast = (p->flags & RTMI_FLAG_FAST_CULL) != 0u && boxes_valid, sigf = (p->flags & RTMI_FLAG_PATH_SIG) != 0u;
    const dim3 block(64 * WAVES_PER_BLOCK);
    P.stack_depth = s->meta.max_bvh_depth + 1u;
    P.shade_threshold = p->shade_threshold ? (p->shade_threshold > 64u ? 64u : p->shade_threshold) : 40u; // tuned on C2..C5 (r02 sweep: 16..48)
    const size_t dyn_lds = (size_t)WAVES_PER_BLOCK * 2u * P.stack_depth * 64u * sizeof(uint32_t);
    const bool prof = (p->flags & RTMI_FLAG_PROFILE) != 0u, sync = (p->flags & RTMI_FLAG_SYNC) != 0u;
    if (prof) {
        if (!p->prof) return fail(RTMI_ERR_INVALID, "RTMI_FLAG_PROFILE needs params.prof");
        P.prof = reinterpret_cast<unsigned long long *>(p->prof);
        HIP_TRY(hipMemsetAsync(P.prof, 0, 2 * RTMI_PROF_SLOTS * sizeof(unsigned long long), stream));
    }
    // kernel selection: default = two-phase schedule, cooperative traversal when fast-cull is on
    // (it implements the fast-cull semantics); RTMI_FLAG_SYNC = per-lane traversal; RTMI_FLAG_ASYNC =
    // per-lane state machine (kept for comparison)
    const bool async = (p->flags & RTMI_FLAG_ASYNC) != 0u && !s->has_deferred;
    const bool coop_ok = s->meta.n_prims < (1u << 22) && s->meta.n_nodes < (1u << 25) && s->meta.n_alt_nodes < (1u << 25);
    const bool coop = fast && !sync && !async && coop_ok;
    P.status = s->status;
    P.queue = s->status + 1;
    P.sky = (p->flags & RTMI_FLAG_SKY) ? 1u : 0u;
    P.ext = ((p->flags & RTMI_FLAG_FACE_FORWARD) ? RTMI_EXT_FACE_FORWARD : 0u) | ((p->flags & RTMI_FLAG_UV_BOOK) ? RTMI_EXT_UV_BOOK : 0u) |
            ((p->flags & RTMI_FLAG_TEST_OVERFLOW) ? RTMI_EXT_TEST_OVERFLOW : 0u);
    // this call's overflow word, the unit counter and the finished-units word start at zero; the sticky word stays
    HIP_TRY(hipMemsetAsync(s->status, 0, 2 * sizeof(unsigned int), stream));
    HIP_TRY(hipMemsetAsync(s->status + 3, 0, 2 * sizeof(unsigned int), stream));
    s->units_total = 0;
    // LDS part of the traversal stack: 512 entries cover the deepest stack ever seen on the reference scenes
    // (447); deeper stacks continue in global memory (64 * (depth + 2) entries per wavefront, the bound of the
    // depth-first order), so the LDS footprint (7.7 KB per wavefront) does not depend on the tree depth
    const bool use_alt = s->dev.gate != nullptr && s->has_alt && !(p->flags & RTMI_FLAG_REF_TREE);
    P.use_alt = use_alt ? 1u : 0u;
    const uint32_t deepest = (use_alt && s->meta.alt_max_depth > s->meta.max_bvh_depth) ? s->meta.alt_max_depth : s->meta.max_bvh_depth;
    P.spill_cap = 64u * (deepest + 2u);
    if (use_alt) { // a 4-wide visit leaves up to three pending entries per level
        const uint32_t wide = 64u * (3u * s->meta.alt_max_depth + 2u);
        if (wide > P.spill_cap) P.spill_cap = wide;
    }
    // the lean kernel (no gates, no spill code) serves scenes without BVH items; everything else takes the extended
    // one with a 512-entry LDS part
    // (lean = scenes WITHOUT any BVH: its instantiation also carries the LDS word ring of the RNG, which pays exactly
    // there, see rtmi_rng.hpp)
    const bool ext = use_alt || s->meta.n_nodes != 0u || (p->flags & (1u << 11));
#ifndef RTMI_COOP_CAP
#define RTMI_COOP_CAP 512u
#endif
#ifndef RTMI_BLK_CAP /* entries of the workgroup's shared stack: 40 000 B of LDS per workgroup, four workgroups per CU */
#define RTMI_BLK_CAP 2432u
#endif
    P.coop_cap = ext ? RTMI_COOP_CAP : P.spill_cap;
    if (p->flags & (1u << 11)) P.coop_cap = 256u; // test knob: a pool this small spills all the time
    // persistent grid: as many wavefronts as the kernel instantiation keeps resident (4 SIMDs x its waves per SIMD)
    const uint32_t wps_req = (p->flags >> 8) & 7u; // experiment knob: requested waves per SIMD (0 = default)
    const bool inst = s->dev.has_prim_xf != 0u || s->dev.has_medium_outer != 0u; // the rare compositions: own instantiations
    const uint32_t wps_run = (coop && !prof && !sigf && !inst && (wps_req == 3u || wps_req == 5u)) ? wps_req : (coop && prof ? 3u : 4u);
    const uint64_t run_slots = (uint64_t)(s->slots / 20) * 4u * wps_run;
    if (coop) {
        const size_t spill_bytes = (size_t)s->slots * P.spill_cap * sizeof(uint2);
        if (spill_bytes > s->spill_bytes) {
            if (s->spill) { HIP_TRY(hipFree(s->spill)); s->spill = nullptr; s->spill_bytes = 0; }
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->spill), spill_bytes));
            s->spill_bytes = spill_bytes;
        }
    }
    P.spill = s->spill;
    // workgroup-cooperative traversal (rtmi_bvh_block.hpp): alternative trees only, no diagnostics build
    const bool bcoop = coop && (p->flags & RTMI_FLAG_BLOCK_COOP) != 0u && use_alt && s->all_alt && !prof && !inst &&
                       wps_req == 0u;
    const size_t bcoop_lds = (size_t)RTMI_BLK_LDS_WORDS(RTMI_BLK_CAP) * sizeof(uint32_t);
    // (test knob bit 11: a stack so small that rounds are throttled all the time — room for 64 visits when it is full)
    if (bcoop) P.coop_cap = (p->flags & (1u << 11)) ? 3u * RTMI_BLK_THREADS + 64u : RTMI_BLK_CAP;
    s->last_kernel = bcoop ? RTMI_KERNEL_BLOCK_COOP : coop ? RTMI_KERNEL_WAVE_COOP : async ? RTMI_KERNEL_ASYNC : RTMI_KERNEL_PERLANE;
    const size_t coop_lds = (size_t)WAVES_PER_BLOCK * (2u * P.coop_cap + 64u * 12u + 128u + RTMI_COOP_DUMMY_WORDS + (ext ? 0u : RTMI_RNG_RING_WORDS) +
                                                              ((inst && (s->needs_insd || ext || sigf)) ? RTMI_COOP_PARK_WORDS : 0u)) * sizeof(uint32_t);
    const uint32_t ntex = P.ntiles_local * 64u;
    uint32_t blocks_total = 0, chunks_total = 0;
    for (uint32_t s0 = 0; s0 < p->ns; s0 += pass_ns) { // one pass unless the sample buffer is smaller than ns samples
    P.pass_s0 = s0;
    P.pass_cnt = p->ns - s0 < pass_ns ? p->ns - s0 : pass_ns;
    P.nchunks = (P.pass_cnt + chunk_spp - 1) / chunk_spp;
    const uint64_t nitems = (uint64_t)P.ntiles_local * P.nchunks;
    if (nitems > 0x7fffffffull) return fail(RTMI_ERR_UNSUPPORTED, "too many (tile, chunk) items in one pass");
    // two-phase kernels: persistent wavefronts that take units from the queue; async kernel: one block per unit
    // (workgroup-cooperative kernel: resident workgroups of RTMI_BLK_WAVES wavefronts; every wavefront takes units)
    const uint64_t blk_slots = run_slots / RTMI_BLK_WAVES;
    const uint64_t nblocks = async ? (nitems + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK
                             : bcoop ? (nitems < blk_slots ? nitems : blk_slots)
                                     : (nitems < run_slots ? nitems : run_slots);
    const dim3 grid((uint32_t)nblocks);
    blocks_total += grid.x; chunks_total += P.nchunks;
    s->units_total += nitems;
#define RTMI_LAUNCH(KERN, F, S, PR, LDS) hipLaunchKernelGGL((KERN<F, S, PR>), grid, block, LDS, stream, s->dev, C, P)
#define RTMI_LAUNCH_COOP(S, PR, W, E, I)                                                                                 \
    do {                                                                                                                 \
        if (coop_lds > 48u * 1024u)                                                                                      \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&rtmi_render_coop<S, PR, W, E, I>),               \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)coop_lds));                     \
        hipLaunchKernelGGL((rtmi_render_coop<S, PR, W, E, I>), grid, block, coop_lds, stream, s->dev, C, P);             \
    } while (0)
    if (bcoop) {
        const dim3 blk(RTMI_BLK_THREADS);
        if (sigf) hipLaunchKernelGGL((rtmi_render_bcoop<true, false>), grid, blk, bcoop_lds, stream, s->dev, C, P);
        else hipLaunchKernelGGL((rtmi_render_bcoop<false, false>), grid, blk, bcoop_lds, stream, s->dev, C, P);
    } else if (coop) {
        const uint32_t wps = wps_req;
        if (inst) { // instanced primitives, media inside transforms: their own instantiations (no diagnostics builds)
            if (prof) return fail(RTMI_ERR_UNSUPPORTED, "the profiling build has no instantiation for instanced primitives / media inside transforms");
            // level 2 (DEFERRED items, list scans, nested media; its group state parked in LDS) also serves the scenes that
            // need level 1 only when they have trees: measured faster there than level 1's own instantiation (forced on
            // final_scene -5.2 % against -7.8 %, random_spheres -3.4 % / -3.1 %); without trees level 1 is the faster one
            // (cornell_box -9.5 % against -18 %) — profiles/r04_experiments/inst_levels_ab3_parked.log
            if (s->needs_insd || ext || sigf) {
                if (sigf) RTMI_LAUNCH_COOP(true, false, 4, true, 2);
                else if (ext) RTMI_LAUNCH_COOP(false, false, 4, true, 2);
                else RTMI_LAUNCH_COOP(false, false, 4, false, 2);
            } else {
                RTMI_LAUNCH_COOP(false, false, 4, false, 1);
            }
        }
        else if (prof) RTMI_LAUNCH_COOP(false, true, 3, true, 0);
        else if (sigf) RTMI_LAUNCH_COOP(true, false, 4, true, 0);
        else if (wps == 3) RTMI_LAUNCH_COOP(false, false, 3, true, 0);
        else if (wps == 5) RTMI_LAUNCH_COOP(false, false, 5, true, 0);
        else if (ext) RTMI_LAUNCH_COOP(false, false, 4, true, 0);
        else RTMI_LAUNCH_COOP(false, false, 4, false, 0);
    } else if (!async) {
        if (prof) { if (fast) RTMI_LAUNCH(rtmi_render_kernel, true, false, true, 0); else RTMI_LAUNCH(rtmi_render_kernel, false, false, true, 0); }
        else if (fast && sigf) RTMI_LAUNCH(rtmi_render_kernel, true, true, false, 0);
        else if (fast) RTMI_LAUNCH(rtmi_render_kernel, true, false, false, 0);
        else if (sigf) RTMI_LAUNCH(rtmi_render_kernel, false, true, false, 0);
        else RTMI_LAUNCH(rtmi_render_kernel, false, false, false, 0);
    } else {
        if (prof) { if (fast) RTMI_LAUNCH(rtmi_render_async, true, false, true, dyn_lds); else RTMI_LAUNCH(rtmi_render_async, false, false, true, dyn_lds); }
        else if (fast && sigf) RTMI_LAUNCH(rtmi_render_async, true, true, false, dyn_lds);
        else if (fast) RTMI_LAUNCH(rtmi_render_async, true, false, false, dyn_lds);
        else if (sigf) RTMI_LAUNCH(rtmi_render_async, false, true, false, dyn_lds);
        else RTMI_LAUNCH(rtmi_render_async, false, false, false, dyn_lds);
    }
#undef RTMI_LAUNCH
#undef RTMI_LAUNCH_COOP
    HIP_TRY(hipGetLastError());
    const bool last = s0 + P.pass_cnt >= p->ns;
    if (stats && last) HIP_TRY(hipEventRecord(s->ev[1], stream));
    hipLaunchKernelGGL(rtmi_resolve_kernel, dim3((ntex + 255) / 256), dim3(256), 0, stream, s->samples, s->partial,
                       reinterpret_cast<rtmi_texel *>(d_texels), P, s0 == 0 ? 1 : 0, last ? 1 : 0,
                       (p->flags & RTMI_FLAG_PROGRESSIVE) ? 1 : 0);
    hipLaunchKernelGGL(rtmi_pass_end_kernel, dim3(1), dim3(1), 0, stream, s->status, (unsigned int)nitems, P.pass_cnt, last ? 1 : 0);
    HIP_TRY(hipGetLastError());
    } // passes
    if (stats) {
        HIP_TRY(hipEventRecord(s->ev[2], stream));
        HIP_TRY(hipEventSynchronize(s->ev[2]));
        float ms_r = 0.f, ms_all = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms_r, s->ev[0], s->ev[1]));
        HIP_TRY(hipEventElapsedTime(&ms_all, s->ev[0], s->ev[2]));
        stats->render_ms = ms_r;
        stats->kernel_ms = ms_all;
        // samples actually traced: pixels inside the image that belong to local tiles
        uint64_t pix = 0;
        const uint32_t txn = tiles_x_of(p);
        for (uint32_t lt = 0; lt < P.ntiles_local; lt++) {
            const uint32_t t = lt * p->tile_world + p->tile_rank;
            const uint32_t ty = t / txn, tx = t % txn;
            const uint32_t w = (tx * RTMI_TILE + RTMI_TILE <= p->nx) ? RTMI_TILE : p->nx - tx * RTMI_TILE;
            const uint32_t h = (ty * RTMI_TILE + RTMI_TILE <= p->ny) ? RTMI_TILE : p->ny - ty * RTMI_TILE;
            pix += (uint64_t)w * h;
        }
        stats->samples = pix * p->ns;
        stats->tiles = P.ntiles_local; stats->chunks = chunks_total; stats->blocks = blocks_total;
        stats->kernel = s->last_kernel;
        unsigned int st = 0; // this call's overflow word (the kernels have finished: ev[2] was waited for)
        HIP_TRY(hipMemcpy(&st, s->status, sizeof(st), hipMemcpyDeviceToHost));
        if (st != 0) {
            HIP_TRY(hipMemset(s->status + 2, 0, sizeof(unsigned int))); // reported here: not again by rtmi_scene_status
            return fail(RTMI_ERR_DEVICE, "cooperative traversal pool overflow (results invalid, texels poisoned): use RTMI_FLAG_SYNC");
        }
    }
    return RTMI_OK;
}

extern "C" int rtmi_render_device(rtmi_scene *s, const rtmi_camera *cam, const rtmi_render_params *p, void *d_texels,
                                  void *stream_, rtmi_stats *stats) {
    if (!s) return fail(RTMI_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lock(s->mu);
    return render_device_locked(s, cam, p, d_texels, stream_, stats);
}

extern "C" int rtmi_scene_status(rtmi_scene *s, uint32_t *overflows) {
    if (!s) return fail(RTMI_ERR_INVALID, "scene is NULL");
    std::lock_guard<std::mutex> lock(s->mu);
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned int st = 0;
    HIP_TRY(hipMemcpy(&st, s->status + 2, sizeof(st), hipMemcpyDeviceToHost));
    if (overflows) *overflows = st;
    if (st != 0) {
        HIP_TRY(hipMemset(s->status + 2, 0, sizeof(unsigned int)));
        return fail(RTMI_ERR_DEVICE, "cooperative traversal pool overflow in an earlier render call (its texels are poisoned): use RTMI_FLAG_SYNC");
    }
    return RTMI_OK;
}

extern "C" int rtmi_untile(const rtmi_render_params *p, const rtmi_texel *g, float *out_linear, uint8_t *out_rgb8) {
    int rc = check_params(p);
    if (rc) return rc;
    if (!g) return fail(RTMI_ERR_INVALID, "gathered buffer is NULL");
    const uint32_t txn = tiles_x_of(p), tyn = tiles_y_of(p);
    const size_t stride = (size_t)local_tiles_of(p, 0) * 64; // every rank padded to rank 0's size
    for (uint32_t ty = 0; ty < tyn; ty++)
        for (uint32_t tx = 0; tx < txn; tx++) {
            const uint32_t t = ty * txn + tx;
            const uint32_t rank = t % p->tile_world, lt = t / p->tile_world;
            const rtmi_texel *src = g + rank * stride + (size_t)lt * 64;
            for (uint32_t ly = 0; ly < RTMI_TILE; ly++) {
                const uint32_t row = ty * RTMI_TILE + ly;
                if (row >= p->ny) break;
                for (uint32_t lx = 0; lx < RTMI_TILE; lx++) {
                    const uint32_t px = tx * RTMI_TILE + lx;
                    if (px >= p->nx) break;
                    const rtmi_texel &e = src[ly * RTMI_TILE + lx];
                    if (e.rgb8 & RTMI_TEXEL_POISON)
                        return fail(RTMI_ERR_DEVICE, "framebuffer holds poisoned texels (traversal pool overflow in the launch that wrote them)");
                    const size_t o = ((size_t)row * p->nx + px) * 3;
                    if (out_linear) { out_linear[o] = e.r; out_linear[o + 1] = e.g; out_linear[o + 2] = e.b; }
                    if (out_rgb8) {
                        out_rgb8[o] = (uint8_t)(e.rgb8 & 255u);
                        out_rgb8[o + 1] = (uint8_t)((e.rgb8 >> 8) & 255u);
                        out_rgb8[o + 2] = (uint8_t)((e.rgb8 >> 16) & 255u);
                    }
                }
            }
        }
    return RTMI_OK;
}

// ---- blocking host API ---------------------------------------------------------------------------------
static int ensure_streams(rtmi_scene *s) {
    if (!s->stream) HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    if (!s->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
    return RTMI_OK;
}
// Waits for `done_ev` of every listed scene; meanwhile (about every 50 ms) reads the devices' unit counters through
// their copy streams and reports progress to the caller's callback — what src/progressbar.rs:6-58 only pretends to.
static int wait_with_progress(rtmi_scene *const *scenes, hipEvent_t *done_ev, uint32_t n, const rtmi_render_params *p) {
    rtmi_progress_fn fn = reinterpret_cast<rtmi_progress_fn>(static_cast<uintptr_t>(p->progress_fn));
    void *user = reinterpret_cast<void *>(static_cast<uintptr_t>(p->progress_user));
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++) total += scenes[i]->units_total;
    bool cancelled = false;
    uint64_t shown = 0;
    if (fn) {
        for (;;) {
            bool all_done = true;
            uint64_t done = 0;
            for (uint32_t i = 0; i < n; i++) {
                rtmi_scene *s = scenes[i];
                HIP_TRY(hipSetDevice(s->device));
                const hipError_t q = hipEventQuery(done_ev[i]);
                if (q == hipErrorNotReady) {
                    all_done = false;
                    unsigned int w[RTMI_STATUS_WORDS] = {0, 0, 0, 0, 0};
                    HIP_TRY(hipMemcpyAsync(w, s->status, sizeof(w), hipMemcpyDeviceToHost, s->copy_stream));
                    HIP_TRY(hipStreamSynchronize(s->copy_stream));
                    const uint64_t d = (uint64_t)w[3] + w[1]; // finished passes + units handed out in the running one
                    done += d < s->units_total ? d : s->units_total;
                } else if (q == hipSuccess) {
                    done += s->units_total;
                } else {
                    return fail(RTMI_ERR_DEVICE, std::string("hipEventQuery: ") + hipGetErrorString(q));
                }
            }
            if (all_done) break;
            if (done < shown) done = shown; // the two words are read without a lock: keep the report monotone
            shown = done;
            if (!cancelled && fn(done, total, user) != 0) cancelled = true;
            struct timespec ts = {0, 50 * 1000 * 1000};
            nanosleep(&ts, nullptr);
        }
    }
    for (uint32_t i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(scenes[i]->device));
        HIP_TRY(hipEventSynchronize(done_ev[i]));
    }
    if (fn && !cancelled && fn(total, total, user) != 0) cancelled = true;
    return cancelled ? fail(RTMI_ERR_CANCELLED, "cancelled by the progress callback") : RTMI_OK;
}
// this call's overflow word of a scene whose kernels have finished
static int check_overflow(rtmi_scene *s) {
    unsigned int st = 0;
    HIP_TRY(hipMemcpy(&st, s->status, sizeof(st), hipMemcpyDeviceToHost));
    if (st != 0) {
        HIP_TRY(hipMemset(s->status + 2, 0, sizeof(unsigned int)));
        return fail(RTMI_ERR_DEVICE, "cooperative traversal pool overflow (results invalid, texels poisoned): use RTMI_FLAG_SYNC");
    }
    return RTMI_OK;
}
static void fill_stats(rtmi_scene *s, const rtmi_render_params *p, rtmi_stats *stats, float ms_render, float ms_all) {
    stats->render_ms = ms_render;
    stats->kernel_ms = ms_all;
    uint64_t pix = 0;
    const uint32_t txn = tiles_x_of(p), nl = local_tiles_of(p, p->tile_rank);
    for (uint32_t lt = 0; lt < nl; lt++) {
        const uint32_t t = lt * p->tile_world + p->tile_rank;
        const uint32_t ty = t / txn, tx = t % txn;
        const uint32_t w = (tx * RTMI_TILE + RTMI_TILE <= p->nx) ? RTMI_TILE : p->nx - tx * RTMI_TILE;
        const uint32_t h = (ty * RTMI_TILE + RTMI_TILE <= p->ny) ? RTMI_TILE : p->ny - ty * RTMI_TILE;
        pix += (uint64_t)w * h;
    }
    stats->samples = pix * p->ns;
    stats->tiles = nl; stats->chunks = 0; stats->blocks = 0; stats->kernel = s->last_kernel;
}

// pinned host mirror of a texel buffer (grow-only)
static int ensure_host_texels(rtmi_texel **h, size_t *have, size_t want) {
    if (want <= *have) return RTMI_OK;
    if (*h) { HIP_TRY(hipHostFree(*h)); *h = nullptr; *have = 0; }
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(h), want * sizeof(rtmi_texel), hipHostMallocDefault));
    *have = want;
    return RTMI_OK;
}

extern "C" int rtmi_render(rtmi_scene *s, const rtmi_camera *cam, const rtmi_render_params *p_in, float *out_linear,
                           uint8_t *out_rgb8, uint64_t *out_path_sig, rtmi_stats *stats) {
    if (!s) return fail(RTMI_ERR_INVALID, "scene is NULL");
    int rc = check_params(p_in);
    if (rc) return rc;
    if (p_in->tile_world != 1) return fail(RTMI_ERR_INVALID, "rtmi_render renders the whole image: tile_world must be 1");
    std::lock_guard<std::mutex> lock(s->mu); // the whole call: it works in the handle's texel / signature buffers
    HIP_TRY(hipSetDevice(s->device));
    if ((rc = ensure_streams(s))) return rc;
    rtmi_render_params p = *p_in;
    const size_t ntex = (size_t)rtmi_local_tiles(&p) * 64;
    if (ntex > s->texel_count) {
        if (s->texels) { HIP_TRY(hipFree(s->texels)); s->texels = nullptr; s->texel_count = 0; }
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->texels), ntex * sizeof(rtmi_texel)));
        s->texel_count = ntex;
    }
    if ((rc = ensure_host_texels(&s->h_texels, &s->h_texel_count, ntex))) return rc;
    p.flags &= ~RTMI_FLAG_PATH_SIG;
    p.path_sig = 0;
    if (out_path_sig) {
        if (ntex > s->sig_count) {
            if (s->d_sig) { HIP_TRY(hipFree(s->d_sig)); s->d_sig = nullptr; s->sig_count = 0; }
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_sig), ntex * sizeof(unsigned long long)));
            s->sig_count = ntex;
        }
        p.flags |= RTMI_FLAG_PATH_SIG;
        p.path_sig = reinterpret_cast<uint64_t>(s->d_sig);
    }
    if (p.progress_fn) {
        // asynchronous launch, then wait and report progress; stats from the scene's events afterwards
        if ((rc = reserve_before_clock(s, &p))) return rc;
        HIP_TRY(hipEventRecord(s->ev[0], s->stream));
        rc = render_device_locked(s, cam, &p, s->texels, s->stream, nullptr);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(s->ev[2], s->stream));
        rtmi_scene *one[1] = {s};
        rc = wait_with_progress(one, &s->ev[2], 1, &p);
        if (rc) return rc;
        if ((rc = check_overflow(s))) return rc;
        if (stats) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, s->ev[0], s->ev[2]));
            fill_stats(s, &p, stats, ms, ms);
        }
    } else {
        rtmi_stats local{};
        rc = render_device_locked(s, cam, &p, s->texels, s->stream, stats ? stats : &local);
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpy(s->h_texels, s->texels, ntex * sizeof(rtmi_texel), hipMemcpyDeviceToHost));
    if (out_path_sig) {
        s->h_sig.resize(ntex);
        HIP_TRY(hipMemcpy(s->h_sig.data(), s->d_sig, ntex * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        const uint32_t txn = tiles_x_of(&p);
        for (uint32_t row = 0; row < p.ny; row++)
            for (uint32_t px = 0; px < p.nx; px++) {
                const uint32_t t = (row / RTMI_TILE) * txn + px / RTMI_TILE;
                out_path_sig[(size_t)row * p.nx + px] = s->h_sig[(size_t)t * 64 + (row % RTMI_TILE) * RTMI_TILE + px % RTMI_TILE];
            }
    }
    return rtmi_untile(&p, s->h_texels, out_linear, out_rgb8);
}

// RTMI_FLAG_PROGRESSIVE: the framebuffer of the running rtmi_render call as it stands (called from its progress callback:
// the calling thread holds s->mu, so none is taken here)
extern "C" int rtmi_partial_image(rtmi_scene *s, const rtmi_render_params *p_in, float *out_linear, uint8_t *out_rgb8,
                                  uint32_t *spp_done) {
    if (!s || !spp_done) return fail(RTMI_ERR_INVALID, "NULL argument");
    int rc = check_params(p_in);
    if (rc) return rc;
    if (p_in->tile_world != 1) return fail(RTMI_ERR_INVALID, "rtmi_partial_image serves rtmi_render: tile_world must be 1");
    *spp_done = 0;
    rtmi_render_params p = *p_in;
    const size_t ntex = (size_t)rtmi_local_tiles(&p) * 64;
    if (!s->texels || ntex > s->texel_count || !s->copy_stream) return fail(RTMI_ERR_INVALID, "no rtmi_render call of this size is running on the handle");
    HIP_TRY(hipSetDevice(s->device));
    if ((rc = ensure_host_texels(&s->h_partial, &s->h_partial_count, ntex))) return rc;
    unsigned int w[RTMI_STATUS_WORDS] = {0, 0, 0, 0, 0}, w2[RTMI_STATUS_WORDS] = {0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(w, s->status, sizeof(w), hipMemcpyDeviceToHost, s->copy_stream));
    HIP_TRY(hipStreamSynchronize(s->copy_stream));
    if (w[4] == 0u) return RTMI_OK; // no pass has finished: nothing to show yet
    for (int attempt = 0; attempt < 2; attempt++) { // a consistent snapshot unless passes end faster than the copy
        HIP_TRY(hipMemcpyAsync(s->h_partial, s->texels, ntex * sizeof(rtmi_texel), hipMemcpyDeviceToHost, s->copy_stream));
        HIP_TRY(hipMemcpyAsync(w2, s->status, sizeof(w2), hipMemcpyDeviceToHost, s->copy_stream));
        HIP_TRY(hipStreamSynchronize(s->copy_stream));
        if (w2[4] == w[4]) break;
        w[4] = w2[4];
    }
    *spp_done = w[4] < p.ns ? w[4] : p.ns;
    return rtmi_untile(&p, s->h_partial, out_linear, out_rgb8);
}

// ---- several GPUs of this process behind one handle: scene replicated, tiles t % n, one gather on devices[0] -----
// RCCL is bound lazily (dlopen) so that single-GPU users of librtmi.so do not load it; inside a PyTorch process the
// soname resolves to the copy torch already mapped.
namespace {
struct RcclApi {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Gather)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr; // rccl.h ncclGather
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("dlopen(librccl.so.1): ") + (dlerror() ? dlerror() : "not found"); return false; }
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        Gather = reinterpret_cast<decltype(Gather)>(dlsym(lib, "ncclGather"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Gather || !GetErrorString) {
            err = "librccl lacks one of ncclCommInitAll/ncclCommDestroy/ncclGroupStart/ncclGroupEnd/ncclGather/ncclGetErrorString";
            dlclose(lib); lib = nullptr;
            return false;
        }
        return true;
    }
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;
// Communicator sets are expensive to make (ncclCommInitAll) and RCCL allows ONE thread at a time to enqueue on a
// communicator: a set belongs to exactly one rtmi_multi handle while that handle lives (checked out at create,
// handed back at destroy) and is kept for the next handle on the same device list afterwards, so that one-shot
// calls (rtmi_render_multi) do not pay the initialisation per image and two live handles on one device list never
// share communicators.  Sets live for the process lifetime.
struct CommSet {
    std::vector<void *> comms;
    bool busy = false;
};
std::map<std::vector<int>, std::vector<CommSet *>> g_comm_pool;
// RTMI_FORCE_RCCL=1 (environment, read at every rtmi_multi_create): also a ONE-entry device list gets a communicator,
// so that the whole collective path — dlopen, ncclCommInitAll, grouped ncclGather on the render stream — runs on a
// single GPU (tests, and bench.py's abi_multi leg at N = 1).  Without it a one-device handle needs no exchange.
bool force_rccl() {
    const char *e = getenv("RTMI_FORCE_RCCL");
    return e && *e && *e != '0';
}
} // namespace
#define RCCL_TRY(expr)                                                                                        \
    do {                                                                                                      \
        int r_ = (expr);                                                                                      \
        if (r_ != 0) return fail(RTMI_ERR_DEVICE, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));     \
    } while (0)

// The persistent multi-device handle.  Everything a render needs between calls lives here; rtmi_multi_render only
// enqueues kernels and the one gather, waits, copies the gathered framebuffer out and un-tiles it.
struct rtmi_multi {
    std::vector<int> devices;
    bool distinct = true;
    std::vector<rtmi_scene *> scenes;    // one per listed device (a device listed twice holds two scenes)
    std::vector<rtmi_texel *> texels;    // per device: tile-packed local framebuffer (grow-only)
    size_t stride = 0;                   // texels per rank in those buffers and in `gathered`
    rtmi_texel *gathered = nullptr;      // on devices[0]: n x stride texels, rank-major
    rtmi_texel *h_gathered = nullptr;    // its pinned host mirror
    size_t h_count = 0;
    CommSet *commset = nullptr;          // RCCL communicators (distinct devices; n > 1 or RTMI_FORCE_RCCL), checked out at create
    std::mutex mu;                       // render calls on one handle serialise (rtmi.h, thread model)
};

extern "C" void rtmi_multi_destroy(rtmi_multi *m) {
    if (!m) return;
    for (size_t i = 0; i < m->texels.size(); i++)
        if (m->texels[i]) { (void)hipSetDevice(m->devices[i]); (void)hipFree(m->texels[i]); }
    if (m->gathered) { (void)hipSetDevice(m->devices[0]); (void)hipFree(m->gathered); }
    if (m->h_gathered) (void)hipHostFree(m->h_gathered);
    for (rtmi_scene *s : m->scenes) rtmi_scene_destroy(s);
    if (m->commset) { // back to the pool: the next handle on this device list takes it over
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        m->commset->busy = false;
    }
    delete m;
}

extern "C" int rtmi_multi_create(const rtmi_scene_desc *desc, const int *devices, uint32_t n, rtmi_multi **out) {
    if (!out) return fail(RTMI_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!desc || !devices || n == 0) return fail(RTMI_ERR_INVALID, "NULL argument or empty device list");
    int rc = validate(desc);
    if (rc) return rc;
    const int ndev = rtmi_device_count();
    if (ndev <= 0) return fail(RTMI_ERR_DEVICE, "no HIP device available (the rtmi path has no CPU fallback)");
    rtmi_multi *m = new (std::nothrow) rtmi_multi();
    if (!m) return fail(RTMI_ERR_NOMEM, "out of host memory");
    struct Guard { rtmi_multi *m; ~Guard() { if (m) rtmi_multi_destroy(m); } } guard{m};
    for (uint32_t i = 0; i < n; i++) {
        if (devices[i] < 0 || devices[i] >= ndev) return fail(RTMI_ERR_INVALID, "device index out of range");
        for (uint32_t k = 0; k < i; k++) m->distinct = m->distinct && devices[k] != devices[i];
    }
    m->devices.assign(devices, devices + n);
    m->scenes.assign(n, nullptr);
    m->texels.assign(n, nullptr);
    // uploads to all devices at once: one host thread per device (hipMalloc / hipMemcpy of one device do not wait for
    // another's; the derived records — leaf, shading, pool-encoded nodes — are built per thread as well)
    std::vector<int> rcs(n, RTMI_OK);
    std::vector<std::string> errs(n);
    {
        std::vector<std::thread> th;
        for (uint32_t i = 0; i < n; i++)
            th.emplace_back([&, i]() {
                rcs[i] = rtmi_scene_create(desc, devices[i], &m->scenes[i]);
                if (rcs[i] == RTMI_OK) {
                    (void)hipSetDevice(devices[i]);
                    rcs[i] = ensure_streams(m->scenes[i]);
                }
                if (rcs[i] != RTMI_OK) errs[i] = g_err; // g_err is thread-local
            });
        for (std::thread &t : th) t.join();
    }
    for (uint32_t i = 0; i < n; i++)
        if (rcs[i] != RTMI_OK) return fail(rcs[i], "device " + std::to_string(devices[i]) + ": " + errs[i]);
    if (m->distinct && (n > 1 || force_rccl())) { // communicators at create, not in the first render
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        if (!g_rccl.load()) return fail(RTMI_ERR_DEVICE, g_rccl.err);
        std::vector<CommSet *> &pool = g_comm_pool[m->devices];
        for (CommSet *c : pool)
            if (!c->busy) { m->commset = c; break; }
        if (!m->commset) {
            CommSet *c = new CommSet();
            c->comms.assign(n, nullptr);
            int r = g_rccl.CommInitAll(c->comms.data(), (int)n, devices);
            if (r != 0) { delete c; return fail(RTMI_ERR_DEVICE, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r)); }
            pool.push_back(c);
            m->commset = c;
        }
        m->commset->busy = true;
    }
    guard.m = nullptr;
    *out = m;
    return RTMI_OK;
}

extern "C" int rtmi_multi_collective(const rtmi_multi *m) {
    if (!m) return -1;
    return m->commset ? RTMI_COLLECTIVE_RCCL : (m->devices.size() > 1 ? RTMI_COLLECTIVE_PEER_COPY : RTMI_COLLECTIVE_NONE);
}

// whole-image parameters -> per-device parameters; validates what the multi-device entry points accept
static int multi_params(const rtmi_multi *m, const rtmi_render_params *p_in, std::vector<rtmi_render_params> &params) {
    int rc = check_params(p_in);
    if (rc) return rc;
    if (p_in->tile_world != 1 || p_in->tile_rank != 0)
        return fail(RTMI_ERR_INVALID, "multi-device render calls render the whole image: tile_rank/tile_world must be 0/1");
    if (p_in->flags & (RTMI_FLAG_PATH_SIG | RTMI_FLAG_PROFILE)) return fail(RTMI_ERR_INVALID, "PATH_SIG / PROFILE are single-device diagnostics");
    const uint32_t n = (uint32_t)m->devices.size();
    params.assign(n, *p_in);
    for (uint32_t i = 0; i < n; i++) { params[i].tile_world = n; params[i].tile_rank = i; }
    return RTMI_OK;
}
// framebuffers of a render of this size (grow-only); every rank padded to rank 0's size (the largest)
static int multi_reserve_texels(rtmi_multi *m, const rtmi_render_params *p0) {
    const uint32_t n = (uint32_t)m->devices.size();
    const size_t stride = (size_t)local_tiles_of(p0, 0) * 64;
    if (stride > m->stride) {
        m->stride = 0; // nothing is valid until every allocation below has succeeded
        for (uint32_t i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(m->devices[i]));
            if (m->texels[i]) { HIP_TRY(hipFree(m->texels[i])); m->texels[i] = nullptr; }
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->texels[i]), stride * sizeof(rtmi_texel)));
            // ranks with fewer tiles than rank 0 never write their padding: zero it once
            HIP_TRY(hipMemset(m->texels[i], 0, stride * sizeof(rtmi_texel)));
        }
        HIP_TRY(hipSetDevice(m->devices[0]));
        if (m->gathered) { HIP_TRY(hipFree(m->gathered)); m->gathered = nullptr; }
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->gathered), (size_t)n * stride * sizeof(rtmi_texel)));
        m->stride = stride;
    }
    return ensure_host_texels(&m->h_gathered, &m->h_count, (size_t)n * m->stride);
}

extern "C" int rtmi_multi_prepare(rtmi_multi *m, const rtmi_render_params *p_in) {
    if (!m) return fail(RTMI_ERR_INVALID, "NULL argument");
    std::vector<rtmi_render_params> params;
    int rc = multi_params(m, p_in, params);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(m->mu);
    if ((rc = multi_reserve_texels(m, &params[0]))) return rc;
    for (size_t i = 0; i < m->scenes.size(); i++)
        if ((rc = rtmi_render_prepare(m->scenes[i], &params[i]))) return rc;
    return RTMI_OK;
}

extern "C" int rtmi_multi_render(rtmi_multi *m, const rtmi_camera *cam, const rtmi_render_params *p_in, float *out_linear,
                                 uint8_t *out_rgb8, rtmi_stats *stats) {
    if (!m || !cam) return fail(RTMI_ERR_INVALID, "NULL argument");
    std::vector<rtmi_render_params> params;
    int rc = multi_params(m, p_in, params);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(m->mu);
    const uint32_t n = (uint32_t)m->devices.size();
    if ((rc = multi_reserve_texels(m, &params[0]))) return rc;
    // rank-major layout of the gathered buffer uses THIS call's stride (the buffers may be larger from an earlier call)
    const size_t stride = (size_t)local_tiles_of(&params[0], 0) * 64;
    std::vector<hipEvent_t> done(n);
    // every device renders its tiles on its own stream, all concurrently
    for (uint32_t i = 0; i < n; i++) {
        rtmi_scene *s = m->scenes[i];
        std::lock_guard<std::mutex> slock(s->mu);
        HIP_TRY(hipSetDevice(m->devices[i]));
        if ((rc = reserve_before_clock(s, &params[i]))) return rc;
        HIP_TRY(hipEventRecord(s->ev[0], s->stream));
        if ((rc = render_device_locked(s, cam, &params[i], m->texels[i], s->stream, nullptr))) return rc;
        HIP_TRY(hipEventRecord(s->ev[1], s->stream));
        done[i] = s->ev[2];
    }
    // the ONE exchange of the path: tile-packed framebuffers -> devices[0]
    if (m->commset) {
        RCCL_TRY(g_rccl.GroupStart());
        for (uint32_t i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(m->devices[i]));
            // (recvbuff is read on the root only; the others pass a valid device pointer rather than NULL for argument checkers)
            RCCL_TRY(g_rccl.Gather(m->texels[i], i == 0 ? m->gathered : m->texels[i], stride * sizeof(rtmi_texel), /*ncclInt8*/ 0, 0, m->commset->comms[i], m->scenes[i]->stream));
        }
        RCCL_TRY(g_rccl.GroupEnd());
    } else {
        for (uint32_t i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(m->devices[i]));
            HIP_TRY(hipMemcpyPeerAsync(m->gathered + (size_t)i * stride, m->devices[0], m->texels[i], m->devices[i], stride * sizeof(rtmi_texel), m->scenes[i]->stream));
        }
    }
    for (uint32_t i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(m->devices[i]));
        HIP_TRY(hipEventRecord(m->scenes[i]->ev[2], m->scenes[i]->stream));
    }
    if ((rc = wait_with_progress(m->scenes.data(), done.data(), n, p_in))) return rc;
    if (stats) memset(stats, 0, sizeof(*stats));
    for (uint32_t i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(m->devices[i]));
        if ((rc = check_overflow(m->scenes[i]))) return rc;
        if (stats) {
            float ms_r = 0.f, ms_all = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms_r, m->scenes[i]->ev[0], m->scenes[i]->ev[1]));
            HIP_TRY(hipEventElapsedTime(&ms_all, m->scenes[i]->ev[0], m->scenes[i]->ev[2]));
            rtmi_stats one{};
            fill_stats(m->scenes[i], &params[i], &one, ms_r, ms_all);
            stats->samples += one.samples; stats->tiles += one.tiles; stats->kernel = one.kernel;
            if (one.render_ms > stats->render_ms) stats->render_ms = one.render_ms;
            if (one.kernel_ms > stats->kernel_ms) stats->kernel_ms = one.kernel_ms;
        }
    }
    HIP_TRY(hipSetDevice(m->devices[0]));
    HIP_TRY(hipMemcpy(m->h_gathered, m->gathered, (size_t)n * stride * sizeof(rtmi_texel), hipMemcpyDeviceToHost));
    return rtmi_untile(&params[0], m->h_gathered, out_linear, out_rgb8);
}

// one-shot form: create + render + destroy (pays uploads, allocations and the communicator look-up on every call)
extern "C" int rtmi_render_multi(const rtmi_scene_desc *desc, const int *devices, uint32_t n, const rtmi_camera *cam,
                                 const rtmi_render_params *p_in, float *out_linear, uint8_t *out_rgb8, rtmi_stats *stats) {
    if (!desc || !devices || !cam || n == 0) return fail(RTMI_ERR_INVALID, "NULL argument or empty device list");
    int rc = check_params(p_in);
    if (rc) return rc;
    if (p_in->tile_world != 1 || p_in->tile_rank != 0)
        return fail(RTMI_ERR_INVALID, "rtmi_render_multi renders the whole image: tile_rank/tile_world must be 0/1");
    if (p_in->flags & (RTMI_FLAG_PATH_SIG | RTMI_FLAG_PROFILE)) return fail(RTMI_ERR_INVALID, "PATH_SIG / PROFILE are single-device diagnostics");
    rtmi_multi *m = nullptr;
    if ((rc = rtmi_multi_create(desc, devices, n, &m))) return rc;
    rc = rtmi_multi_render(m, cam, p_in, out_linear, out_rgb8, stats);
    const std::string keep = g_err; // destroy must not lose the message
    rtmi_multi_destroy(m);
    if (rc) g_err = keep;
    return rc;
}

// P3 writer — tests/test.rs:59,79
extern "C" size_t rtmi_ppm_p3(uint32_t nx, uint32_t ny, const uint8_t *rgb8, char *buf, size_t cap) {
    const size_t need = 40 + (size_t)nx * ny * 12;
    if (!buf || cap < need || !rgb8) return need;
    size_t n = (size_t)snprintf(buf, cap, "P3\n%u %u\n255\n", nx, ny);
    static const char digits[] = "0123456789";
    const size_t npx = (size_t)nx * ny;
    char *w = buf + n;
    for (size_t i = 0; i < npx * 3; i++) {
        const unsigned v = rgb8[i];
        if (v >= 100) { *w++ = digits[v / 100]; *w++ = digits[(v / 10) % 10]; *w++ = digits[v % 10]; }
        else if (v >= 10) { *w++ = digits[v / 10]; *w++ = digits[v % 10]; }
        else { *w++ = digits[v]; }
        *w++ = (i % 3 == 2) ? '\n' : ' ';
    }
    return (size_t)(w - buf);
}

// streaming writer: P3 text identical to rtmi_ppm_p3, or binary P6
extern "C" int rtmi_write_ppm(const char *path, uint32_t nx, uint32_t ny, const uint8_t *rgb8, int format) {
    if (!path || !rgb8) return fail(RTMI_ERR_INVALID, "rtmi_write_ppm: path or rgb8 is NULL");
    if (format != 3 && format != 6) return fail(RTMI_ERR_INVALID, "rtmi_write_ppm: format must be 3 (P3 text) or 6 (P6 binary)");
    FILE *f = fopen(path, "wb");
    if (!f) return fail(RTMI_ERR_INVALID, "rtmi_write_ppm: cannot open the output file");
    bool ok = fprintf(f, "P%d\n%u %u\n255\n", format, nx, ny) > 0;
    const size_t total = (size_t)nx * ny * 3;
    if (format == 6) {
        ok = ok && fwrite(rgb8, 1, total, f) == total;
    } else {
        static const char digits[] = "0123456789";
        std::vector<char> chunk((size_t)1 << 20);
        const size_t per_chunk = chunk.size() / 4; // <= 4 characters per value
        for (size_t i0 = 0; ok && i0 < total; i0 += per_chunk) {
            const size_t i1 = i0 + per_chunk < total ? i0 + per_chunk : total;
            char *w = chunk.data();
            for (size_t i = i0; i < i1; i++) {
                const unsigned v = rgb8[i];
                if (v >= 100) { *w++ = digits[v / 100]; *w++ = digits[(v / 10) % 10]; *w++ = digits[v % 10]; }
                else if (v >= 10) { *w++ = digits[v / 10]; *w++ = digits[v % 10]; }
                else { *w++ = digits[v]; }
                *w++ = (i % 3 == 2) ? '\n' : ' ';
            }
            const size_t n = (size_t)(w - chunk.data());
            ok = fwrite(chunk.data(), 1, n, f) == n;
        }
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? RTMI_OK : fail(RTMI_ERR_INVALID, "rtmi_write_ppm: write failed");
}

// ---- probes (parity tests call these through the C ABI) ------------------------------
extern "C" int rtmi_probe_math(int op, const float *x, const float *y, float *out, uint32_t n) {
    if (rtmi_device_count() <= 0) return fail(RTMI_ERR_DEVICE, "no HIP device available");
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dx), n * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dy), n * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dout), n * 4));
    HIP_TRY(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dy, y ? y : x, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rtmi_math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, op, dx, dy, dout, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dout);
    return RTMI_OK;
}
extern "C" int rtmi_probe_xform(const rtmi_xform *xforms, uint32_t count, const float *a, const float *b, float *out, uint32_t n) {
    if (rtmi_device_count() <= 0) return fail(RTMI_ERR_DEVICE, "no HIP device available");
    rtmi_xform *dx = nullptr;
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dx), (count ? count : 1) * sizeof(rtmi_xform)));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&da), n * 12));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&db), n * 12));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dout), n * 13 * 4));
    if (count) HIP_TRY(hipMemcpy(dx, xforms, count * sizeof(rtmi_xform), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(da, a, n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, b, n * 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rtmi_xform_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dx, (int)count, da, db, dout, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, n * 13 * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return RTMI_OK;
}
extern "C" int rtmi_probe_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out, uint32_t n) {
    if (rtmi_device_count() <= 0) return fail(RTMI_ERR_DEVICE, "no HIP device available");
    uint32_t *dc = nullptr, *dk = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dc), n * 16));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dk), n * 8));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dout), n * 16));
    HIP_TRY(hipMemcpy(dc, ctr, n * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dk, key, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rtmi_philox_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dc, dk, dout, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, n * 16, hipMemcpyDeviceToHost));
    (void)hipFree(dc); (void)hipFree(dk); (void)hipFree(dout);
    return RTMI_OK;
}
