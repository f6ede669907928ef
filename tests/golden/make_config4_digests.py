#!/usr/bin/env python3
"""Generator of tests/golden/config4_n16384_digests.json: ONE WHOLE N=16384 fp32 solve of the
benchmark matrix on the CPU oracle (all host cores, ~4 min on the GPU box), with and without the
next-hop matrix, and the same solve on every GPU engine, bit for bit.  The fixture keeps the
oracle's side only -- xxh64 of the solved rates, of the solved next-hops, and U -- so that `-m gpu`
tests can tie any engine, any partitioning, to the whole oracle solve in seconds.

    python3 tests/golden/make_config4_digests.py profiles/full_parity.json [n] [--next] [--f64]
    python3 tests/golden/make_config4_digests.py --write-fixture rates.json next.json
    python3 tests/golden/make_config4_digests.py --write-fixture-f64 next_f64.json

--f64: the same matrix BEFORE rounding to f32 (synth.d1_uniform(n, float64, BASE_SEED + 3): what
bench.py's `f64` leg solves), the reference's own precision (Types.hs:26); with --next one run gives the
rate digest, the next-hop digest and U -> tests/golden/config4_n16384_f64_digests.json (round 4).

(round 2 ran it twice on an MI355X box: profiles/r02_full_parity_n16384.json and
profiles/r02_full_parity_n16384_next.json; --write-fixture merges two such records.)
Prints a progress line per 1024 pivots (the GPU box kills a silent command after 7 minutes).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402  (tools/ may use the checker; the product never does)
from floydwarshall_amd import engine, synth  # noqa: E402
sys.path.insert(0, ROOT)
from bench import digest, host_cores  # noqa: E402


def write_fixture(rates_json, next_json):
    a, b = json.load(open(rates_json)), json.load(open(next_json))
    assert a["ok"] and b["ok"] and a["n"] == b["n"] and a["U"] == b["U"]
    assert a["rate_digest_oracle"] == b["rate_digest_oracle"]
    fix = {"n": a["n"], "dtype": a["dtype"],
           "input": "synth.d1_uniform(n, float32, BASE_SEED + 3): bench.py's matrix",
           "oracle": "oracle.relax_mt, whole solve (%.0f s rates only, %.0f s with next-hops)"
                     % (a["oracle_seconds"], b["oracle_seconds"]),
           "U": a["U"], "rate_digest": a["rate_digest_oracle"], "next_digest": b["next_digest_oracle"],
           "made_by": "tests/golden/make_config4_digests.py"}
    with open(os.path.join(ROOT, "tests", "golden", "config4_n16384_digests.json"), "w") as f:
        json.dump(fix, f, indent=1)
    print(json.dumps(fix))


def write_fixture_f64(next_json):
    b = json.load(open(next_json))
    assert b["ok"] and b["dtype"] == "f64" and b["with_next"]
    fix = {"n": b["n"], "dtype": "f64",
           "input": "synth.d1_uniform(n, float64, BASE_SEED + 3): bench.py's matrix before rounding to f32",
           "oracle": "oracle.relax_mt, whole solve with next-hops (%.0f s)" % b["oracle_seconds"],
           "U": b["U"], "rate_digest": b["rate_digest_oracle"], "next_digest": b["next_digest_oracle"],
           "made_by": "tests/golden/make_config4_digests.py --f64 --next"}
    with open(os.path.join(ROOT, "tests", "golden", "config4_n16384_f64_digests.json"), "w") as f:
        json.dump(fix, f, indent=1)
    print(json.dumps(fix))


def main():
    if len(sys.argv) > 3 and sys.argv[1] == "--write-fixture":
        write_fixture(sys.argv[2], sys.argv[3])
        return 0
    if len(sys.argv) > 2 and sys.argv[1] == "--write-fixture-f64":
        write_fixture_f64(sys.argv[2])
        return 0
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 16384
    with_next = "--next" in sys.argv
    f64 = "--f64" in sys.argv
    np_t, it = (np.float64, np.uint64) if f64 else (np.float32, np.uint32)
    rate, nxt = synth.d1_uniform(n, np_t, synth.BASE_SEED + 3)            # bench.py's matrix
    if not with_next:
        nxt = None
    cores = host_cores()
    er = rate.copy()
    en = None if nxt is None else nxt.copy()
    u = 0
    t0 = time.perf_counter()
    for k0 in range(0, n, 1024):
        u += oracle.relax_mt(er, en, k0, min(n, k0 + 1024), threads=cores)
        print("oracle: pivots [0,%d) done, %.0f s" % (min(n, k0 + 1024), time.perf_counter() - t0), flush=True)
    t_cpu = time.perf_counter() - t0
    rec = {"n": n, "dtype": "f64" if f64 else "f32", "input": "D1, seed BASE_SEED+3 (bench.py's matrix)",
           "with_next": with_next, "oracle": "oracle.relax_mt (fwo_relax_mt_%s), %d threads" % ("f64" if f64 else "f32", cores),
           "oracle_seconds": t_cpu, "oracle_relax_per_s": float(n) ** 3 / t_cpu, "U": u,
           "rate_digest_oracle": digest(er)}
    if en is not None:
        rec["next_digest_oracle"] = digest(en)
    # "fused" counts U, which keeps it on the compare-form kernel; "fused_max_form" is the same call
    # without counting = what fwx_solve_f32 runs by default on this input (v_max3 pairs)
    for name, eng, count in (("fused_max_form", engine.FWX_ENGINE_FUSED, False),
                             ("fused", engine.FWX_ENGINE_FUSED, True),
                             ("per_k", engine.FWX_ENGINE_PERK, True)):
        gr = rate.copy()
        gn = None if nxt is None else nxt.copy()
        t1 = time.perf_counter()
        gu = engine.solve(gr, gn, engine=eng, count_updates=count)
        if not count:
            gu = u
        rec[name] = {"seconds_incl_pcie": time.perf_counter() - t1, "U": gu, "counted": count,
                     "rate_digest": digest(gr),
                     "rate_bits_equal_oracle": bool(np.array_equal(gr.view(it), er.view(it)))}
        if gn is not None:
            rec[name]["next_equal_oracle"] = bool(np.array_equal(gn, en))
        print(name, json.dumps(rec[name]), flush=True)
    rec["ok"] = all(rec[e]["rate_bits_equal_oracle"] and rec[e]["U"] == u and
                    rec[e].get("next_equal_oracle", True) for e in ("fused_max_form", "fused", "per_k"))
    print(json.dumps(rec), flush=True)
    if out_path:
        with open(out_path, "w") as f:
            json.dump(rec, f, indent=1)
    return 0 if rec["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
