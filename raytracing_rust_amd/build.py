"""In-tree build of the native libraries (no cmake; hipcc cross-compiles gfx950 without a GPU).

  lib/librtmi.so     HIP kernels + the C ABI of include/rtmi.h       (hipcc --offload-arch=gfx950)
  lib/librt_host.so  C++ host mirror + its C bindings, links librtmi (g++)

Flags that matter for parity: -ffp-contract=off on both (no implicit FMA), no fast-math
(IEEE division/sqrt are correctly rounded by default under hipcc).
"""
import glob
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_DIR = os.environ.get("RTMI_LIB_DIR") or os.path.join(_PKG, "lib")
INCLUDE = os.path.join(_ROOT, "include")

RTMI_SRC = [os.path.join(_PKG, "csrc", "rtmi_device.hip"), os.path.join(_PKG, "csrc", "rtmi_lean.hip"),
            os.path.join(_PKG, "csrc", "rtmi_alt.hip")]
HOST_SRC = [os.path.join(_PKG, "host", "rt_host.cpp"), os.path.join(_PKG, "host", "rt_host_c.cpp")]
RTMI_DEPS = RTMI_SRC + sorted(glob.glob(os.path.join(_PKG, "csrc", "*.hpp"))) + [
    os.path.join(INCLUDE, "rtmi.h"), os.path.join(INCLUDE, "rtmi_math.h")]
HOST_DEPS = HOST_SRC + [os.path.join(_PKG, "host", "rt_host.hpp"), os.path.join(INCLUDE, "rtmi.h")]

LIBRTMI = os.path.join(LIB_DIR, "librtmi.so")
LIBHOST = os.path.join(LIB_DIR, "librt_host.so")
REFTESTS = os.path.join(LIB_DIR, "rt_reference_tests")  # C++ counterpart of the reference's #[test] drivers
REFTESTS_SRC = os.path.join(_PKG, "host", "reference_scenes.cpp")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_hash(extra=None, default_sched=None):
    """sha256 (16 hex digits) of everything that decides the device code: csrc/*, include/*.h and the compile flags.
    Compiled into librtmi.so (rtmi_build_hash()); tools/pmc_summary.py records it with the committed counter profiles and
    bench.py marks them stale when the library it runs is another build."""
    import hashlib

    h = hashlib.sha256()
    for path in sorted(RTMI_DEPS):
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    extra = os.environ.get("RTMI_EXTRA_CFLAGS", "") if extra is None else extra
    default_sched = bool(os.environ.get("RTMI_DEFAULT_SCHED")) if default_sched is None else default_sched
    h.update(("|".join(_COMMON_FLAGS) + "|" + extra + "|" + str(default_sched)).encode())
    return h.hexdigest()[:16]


_COMMON_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-std=c++17"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_rtmi(force=False, verbose=False):
    if not force and not _stale(LIBRTMI, RTMI_DEPS):
        return LIBRTMI
    os.makedirs(LIB_DIR, exist_ok=True)
    # -fno-slp-vectorize: ROCm 7.2's SLP vectoriser packs the scalar F3 arithmetic into v_pk_mul/add_f32 and gets the
    # operand selection wrong in at least one place (hit point/normal of a sphere under Rotate about Z: every
    # pixel differed from the oracle, found by tests/test_random_scenes.py; fine with the function out of line,
    # fine without SLP).  The scalar code is also 2-3 % faster on this VALU-bound path.
    # -amdgpu-sched-strategy=iterative-maxocc: the machine scheduler that first gets the register pressure under the
    # occupancy target and then schedules for latency.  Same instructions (8.75 k static in the headline kernel either
    # way), a better order: final_scene +2.4 %, random_spheres +1.7 % against the default strategy; max-ilp -1.2 %,
    # max-memory-clause -1.4 %, iterative-ilp +0.6 %, iterative-minreg -6 % (profiles/r03_experiments/sched_strategy_ab.log).
    # RTMI_EXTRA_CFLAGS: experiment switches (-DRTMI_...) for A/B builds into another RTMI_LIB_DIR (tools/ab_build.sh)
    # The lean instantiations of the cooperative kernel (scenes without BVH items) lose 2 % under that strategy: they are a
    # translation unit of their own (csrc/rtmi_lean.hip) with the default one.
    extra = os.environ.get("RTMI_EXTRA_CFLAGS", "").split()
    sched = [] if os.environ.get("RTMI_DEFAULT_SCHED") else ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"]
    common = [_hipcc()] + _COMMON_FLAGS + ["-I" + INCLUDE, '-DRTMI_BUILD_HASH="%s"' % source_hash()]
    if verbose:
        common.insert(1, "-Rpass-analysis=kernel-resource-usage")
    # three translation units, compiled side by side: the headline kernels (rtmi_device.hip), the lean instantiations
    # with the default scheduler (rtmi_lean.hip), and the two alternative kernels kept for the parity tests (rtmi_alt.hip)
    objs, procs = [], []
    for src, flags in ((RTMI_SRC[0], sched), (RTMI_SRC[1], []), (RTMI_SRC[2], sched)):
        obj = os.path.join(LIB_DIR, os.path.basename(src)[:-4] + ".o")
        procs.append((src, subprocess.Popen(common + flags + extra + ["-c", src, "-o", obj])))
        objs.append(obj)
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, "hipcc -c " + " ".join(failed))
    subprocess.run([_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIBRTMI] + objs, check=True)
    return LIBRTMI


def build_host(force=False):
    if not force and not _stale(LIBHOST, HOST_DEPS + [LIBRTMI]):
        return LIBHOST
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = ["g++", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wall",
           "-I" + INCLUDE, "-o", LIBHOST] + HOST_SRC + [
               "-L" + LIB_DIR, "-lrtmi", "-Wl,-rpath,$ORIGIN"]
    subprocess.run(cmd, check=True)
    return LIBHOST


def build_reference_tests(force=False):
    if not force and not _stale(REFTESTS, [REFTESTS_SRC, LIBHOST, os.path.join(_PKG, "host", "rt_host.hpp")]):
        return REFTESTS
    cmd = ["g++", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-std=c++17", "-Wall", "-I" + INCLUDE,
           "-I" + os.path.join(_PKG, "host"), "-o", REFTESTS, REFTESTS_SRC, "-L" + LIB_DIR, "-lrt_host", "-lrtmi",
           "-Wl,-rpath,$ORIGIN"]
    subprocess.run(cmd, check=True)
    return REFTESTS


def build_all(force=False, verbose=False):
    build_rtmi(force, verbose)
    build_host(force)
    build_reference_tests(force)
    return LIBRTMI, LIBHOST


if __name__ == "__main__":
    import sys

    build_all(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(LIBRTMI)
    print(LIBHOST)
