#!/usr/bin/env python3
"""Fold the per-kernel means of the rocprofv3 --pmc passes (tools/pmc_pass.sh -> gpurun_out/pmc_<tag>_<counter>.json)
into the profile bench.py quotes: HBM bytes per launch of every kernel of one step.

    tools/make_pmc_profile.py <tag> <n_gates> <width> > profiles/r02_pmc_hbm_bytes.json

FETCH_SIZE / WRITE_SIZE are in KiB.  The byte model is CALIBRATED (profiles/r06_pmc_calibration.txt: tools/ubench/gather.hip, known
access counts per pattern under the same counters): on gfx950 every read request to the fabric is a 128-byte line — streaming,
scattered 4 / 8 / 16-byte loads and 512-byte records alike (TCC_EA0_RDREQ_128B = TCC_EA0_RDREQ) — and FETCH_SIZE tallies 64 B per
request, so it is doubled here for EVERY kernel (= 128 x TCC_EA0_RDREQ); WRITE_SIZE is exact as reported (64 B per full request, 32 B
per scattered store); an atomic shows as a 32-byte write and its read half is not counted (`atomics` x 32 B more than reported)."""
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
                         "one pass per counter group; byte model calibrated in profiles/r06_pmc_calibration.txt (reads = FETCH_SIZE x 2 = 128 B x TCC_EA0_RDREQ in every "
                         "pattern; WRITE_SIZE exact; an atomic = 32 B written, its read half uncounted)",
               "kernels": kernels}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
