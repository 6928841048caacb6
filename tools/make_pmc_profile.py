#!/usr/bin/env python3
"""Fold the per-kernel means of the rocprofv3 --pmc passes (tools/pmc_pass.sh -> gpurun_out/pmc_<tag>_<counter>.json)
into the profile bench.py quotes: HBM bytes per launch of every kernel of one step.

    tools/make_pmc_profile.py <tag> <n_gates> <width> > profiles/r02_pmc_hbm_bytes.json

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE = TCC_EA0_RDREQ x 64 B, requests are 128 B) and is doubled here."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def norm(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"<.*$", "", name)


def load(tag, counter):
    p = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{counter}.json")
    with open(p) as f:
        raw = json.load(f)
    out = {}
    for k, v in raw.items():
        out.setdefault(norm(k), {}).update(v)
    return out


def main(tag, n_gates, width):
    fetch, write = load(tag, "FETCH_SIZE"), load(tag, "WRITE_SIZE")
    try:
        req = load(tag, "TCC_EA0_RDREQ_sum")
    except OSError:
        req = {}
    try:
        sq = load(tag, "SQ_WAVE_CYCLES")
    except OSError:
        sq = {}
    steps = None
    kernels = {}
    for k in sorted(fetch):
        if not k.startswith("c2a::"):
            continue
        f = fetch[k].get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
        w = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
        e = {"fetch_bytes": f, "write_bytes": w, "hbm_bytes_per_launch": f + w, "launches_in_pass": fetch[k]["launches"]}
        r = req.get(k)
        if r:
            e["ea_read_requests"] = r.get("TCC_EA0_RDREQ_sum")
            e["ea_write_requests"] = r.get("TCC_EA0_WRREQ_sum")
            e["atomics"] = r.get("TCC_ATOMIC_sum")
        q = sq.get(k)
        if q and q.get("SQ_WAVE_CYCLES"):
            e["sq_wait_frac"] = q.get("SQ_WAIT_ANY", 0.0) / q["SQ_WAVE_CYCLES"]
            e["valu_per_launch"] = q.get("SQ_INSTS_VALU")
            e["salu_per_launch"] = q.get("SQ_INSTS_SALU")
            e["vmem_per_launch"] = q.get("SQ_INSTS_VMEM")
        kernels[k] = e
    # launches per step: the pass ran (warmup 1 + steps 2) = 3 steps
    base = kernels.get("c2a::k_boolify", {}).get("launches_in_pass", 3) or 3
    for e in kernels.values():
        e["launches_per_step"] = e["launches_in_pass"] / base
    json.dump({"workload": {"n_gates": n_gates, "width": width},
               "source": f"rocprofv3 --kernel-trace --pmc <counter> -- python bench.py --steps 2 --warmup 1 (tools/pmc_pass.sh {tag} ...), "
                         "one pass per counter group; FETCH_SIZE x2 per the gfx950 note of MI355X_MICROARCH.md",
               "kernels": kernels}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
