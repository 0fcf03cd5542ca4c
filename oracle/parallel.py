"""The fp32 oracle on several host cores — TEST INFRASTRUCTURE (tests/ only; the product never imports oracle/).

A parity test at ~10^6 paths per scene needs seconds of oracle time per core; the rows of an image are independent
(the random streams are keyed by pixel and sample), so every worker renders its own bands of rows with its own
Oracle instance and the bands are pasted together.  Workers are fresh interpreters (`python -m oracle.parallel`,
numpy + the oracle only): a forked child of a process that has initialised HIP would inherit its device handles.
Reference loop restated by the oracle: tests/test.rs:62-79, src/color.rs:6-23."""
import importlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(spec_path, out_path):
    spec = json.load(open(spec_path))
    for p in spec["path"]:
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle.oracle import Oracle

    mod = importlib.import_module(spec["scenes_mod"])
    orc = Oracle(spec.get("precision", "f32"))
    nx, ny, ns = spec["nx"], spec["ny"], spec["ns"]
    cam, world = mod.build(orc, spec["name"], nx, ny, seed=spec["scene_seed"])
    res = {}
    for r0, r1 in spec["bands"]:
        out = orc.render(cam, world, nx, ny, ns, seed=spec["seed"], flags=spec["oflags"], rows=(r0, r1))
        res["lin_%d_%d" % (r0, r1)] = out["linear"][r0:r1]
        res["rgb_%d_%d" % (r0, r1)] = out["rgb"][r0:r1]
        res["sig_%d_%d" % (r0, r1)] = out["sig"][r0:r1]
        res["mean_%d_%d" % (r0, r1)] = out["mean"][r0:r1]
    np.savez(out_path, **res)
    orc.free_all()


def render_parallel(scenes_mod, name, nx, ny, ns, seed, oflags, scene_seed=1, workers=None, band=4, timeout=900,
                    precision="f32", rows=None):
    """Oracle image (precision "f32": the device's arithmetic contract; "f64": the literal restatement of the
    reference's own arithmetic) of scene `name` built by module `scenes_mod` (its build(api, name, nx, ny, seed=)),
    rendered in bands of `band` rows dealt round-robin to `workers` processes.  rows: only these rows (each a band of
    its own; the others stay zero).  Returns dict(linear, rgb, sig, mean) like Oracle.render."""
    if workers is None:
        try:
            workers = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            workers = os.cpu_count() or 1
        workers = max(1, min(workers, 16))
    bands = [(r, min(ny, r + band)) for r in range(0, ny, band)] if rows is None else [(int(r), int(r) + 1) for r in rows]
    workers = min(workers, len(bands))
    lin = np.zeros((ny, nx, 3), np.float32)
    rgb = np.zeros((ny, nx, 3), np.int32)
    sig = np.zeros((ny, nx), np.uint64)
    mean = np.zeros((ny, nx, 3), np.float64)
    with tempfile.TemporaryDirectory(prefix="orc_par_") as tmp:
        procs = []
        for w in range(workers):
            spec = {"path": [_ROOT, os.path.join(_ROOT, "tests")], "scenes_mod": scenes_mod, "name": name, "nx": nx, "ny": ny,
                    "ns": ns, "seed": seed, "oflags": oflags, "scene_seed": scene_seed, "bands": bands[w::workers],
                    "precision": precision}
            sp, op = os.path.join(tmp, "spec%d.json" % w), os.path.join(tmp, "out%d.npz" % w)
            json.dump(spec, open(sp, "w"))
            env = dict(os.environ, PYTHONPATH=_ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
            procs.append((subprocess.Popen([sys.executable, "-m", "oracle.parallel", sp, op], cwd=_ROOT, env=env), op))
        try:
            for p, op in procs:
                if p.wait(timeout=timeout) != 0:
                    raise RuntimeError("oracle worker exited with code %d" % p.returncode)
                z = np.load(op)
                for key in z.files:
                    kind, r0, r1 = key.split("_")
                    {"lin": lin, "rgb": rgb, "sig": sig, "mean": mean}[kind][int(r0):int(r1)] = z[key]
        finally:
            for p, _ in procs:  # exactly the children started above
                if p.poll() is None:
                    p.kill()
                    p.wait()
    return {"linear": lin, "rgb": rgb, "sig": sig, "mean": mean}


if __name__ == "__main__":
    _worker(sys.argv[1], sys.argv[2])
