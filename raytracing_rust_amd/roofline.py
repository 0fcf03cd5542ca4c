"""Algorithmic work per sample (one camera path) from operation counts.

SURVEY.md §8(d) fixes the cost table (fp32 device layout): bytes and flops per counted
operation.  Counts per sample come from the instrumented CPU oracle (bench.py's
cpu_baseline leg) — the product never counts.  Peaks are from
/opt/skills/guides/MI355X_MICROARCH.md (HBM3E 8.0 TB/s spec; FP32 vector 157.3 TFLOP/s).
"""
HBM_PEAK_GBS = 8000.0
FP32_VALU_PEAK_TFLOPS = 157.3

# counter name -> (bytes, flops) per counted operation
COST = {
    "aabb": (32, 18),
    "sphere": (16, 23),
    "msphere": (36, 32),
    "rect": (24, 8),
    "sphere_accept": (0, 17 + 2),   # + 1 sqrt + 1 div
    "rect_accept": (0, 12),
    "xform": (32, 24),
    "medium_draw": (8, 10 + 1),     # + 1 log
    "mat_fetch": (16, 0),
    "tex_solid": (12, 0),
    "tex_checker": (24, 8 + 3),     # + 3 sin
    "tex_noise": (1344, 1050 + 1),
    "tex_image": (3, 0),
    "sc_lambert": (0, 12),
    "sc_isotropic": (0, 12),
    "sc_metal": (0, 20),
    "sc_dielectric": (0, 45 + 2),
    "sphere_trials": (0, 9),
    "disk_trials": (0, 6),
    "samples": (0, 30),             # camera ray
}
FRAMEBUFFER_BYTES_PER_PIXEL = 12   # 12 B / spp per sample
PHILOX_INTOPS_PER_BLOCK = 70       # integer VALU, reported separately


def per_sample(counters, ns):
    """counters: dict of totals (oracle.COUNTER_NAMES) over `samples` camera paths."""
    n = float(counters["samples"])
    if n <= 0:
        raise ValueError("no samples counted")
    by = sum(COST[k][0] * counters.get(k, 0) for k in COST) / n + FRAMEBUFFER_BYTES_PER_PIXEL / float(ns)
    fl = sum(COST[k][1] * counters.get(k, 0) for k in COST) / n
    return {
        "bytes": by,
        "flops": fl,
        "philox_intops": PHILOX_INTOPS_PER_BLOCK * counters.get("draws", 0) / 4.0 / n,
        "queries": counters.get("queries", 0) / n,
        "draws": counters.get("draws", 0) / n,
    }


# chip constants for the SQ counter fractions (MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 @ 2.4 GHz; a wave64 VALU
# instruction issues over 2 cycles)
N_SIMD = 1024
CLOCK_HZ = 2.4e9
VALU_ISSUE_CYCLES = 2.0


def sq_fractions(prof):
    """prof: one workload entry of profiles/pmc_summary.json (tools/pmc_summary.py).  Returns the measured fractions
    of the PROFILED launch of the dominant kernel (all <= 1 by construction):
      issue_frac = wave64 VALU instructions / (SIMDs x clock / 2 cycles x launch time)
      lane_util  = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): active lanes per issued VALU instruction
      wait_frac  = SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of a wavefront's life parked at s_waitcnt / barriers"""
    sq = (prof or {}).get("sq")
    if not sq or not sq.get("launch_ns"):
        return None
    t = sq["launch_ns"] * 1e-9
    out = {"profiled_launch_ms": round(sq["launch_ns"] * 1e-6, 3)}
    if sq.get("SQ_INSTS_VALU"):
        out["issue_frac"] = round(sq["SQ_INSTS_VALU"] / (N_SIMD * CLOCK_HZ / VALU_ISSUE_CYCLES * t), 4)
    if sq.get("SQ_ACTIVE_INST_VALU"):
        out["lane_util"] = round(sq.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * sq["SQ_ACTIVE_INST_VALU"]), 4)
    if sq.get("SQ_WAVE_CYCLES"):
        out["wait_frac"] = round(sq.get("SQ_WAIT_ANY", 0.0) / sq["SQ_WAVE_CYCLES"], 4)
    return out
