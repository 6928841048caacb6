#!/bin/bash
# usage (on the GPU box): tools/pmc_calibrate.sh > gpurun_out/pmc_calibration.txt — tools/ubench/gather (known access counts per pattern) under
# rocprofv3 --pmc, one pass per counter group (TCC has four slots: FETCH_SIZE takes three, WRITE_SIZE two); per kernel and counter
# the value PER ACCESS, next to the bytes the pattern needs.  Also: what the box's rocprofv3 calls the counters.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
echo "# counters this rocprofv3 lists (TCC_EA0 / SIZE):"; rocprofv3 -L 2>/dev/null | grep -o -E "(TCC_EA0_(RD|WR)REQ[A-Za-z0-9_]*|TCC_EA0_ATOMIC[A-Za-z0-9_]*|TCC_ATOMIC[A-Za-z0-9_]*|FETCH_SIZE|WRITE_SIZE|TCC_REQ[A-Za-z0-9_]*|TCC_HIT[A-Za-z0-9_]*|TCC_MISS[A-Za-z0-9_]*)" | sort -u | tr '\n' ' '; echo
$R/tools/ubench/gather > /tmp/gather_plain.txt 2>&1; cat /tmp/gather_plain.txt
pass=0
for group in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum" \
             "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_DRAM_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum TCC_ATOMIC_sum" \
             "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  pass=$((pass+1)); d=/tmp/cal_$pass; rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $d -- $R/tools/ubench/gather > $d.log 2>&1
  python3 $R/tools/pmc_by_kernel.py $d > /tmp/cal_$pass.json 2>/dev/null || echo "{}" > /tmp/cal_$pass.json
done
python3 - <<'PY'
import json, re
known = {}
for l in open("/tmp/gather_plain.txt"):
    m = re.match(r"(\S+)\s+(\d+) accesses x\s+(\d+) B", l)
    if m: known[m.group(1)] = (int(m.group(2)), int(m.group(3)))
vals = {}
for p in range(1, 8):
    try: d = json.load(open(f"/tmp/cal_{p}.json"))
    except Exception: d = {}
    for k, v in d.items():
        name = re.sub(r"^void ", "", k); base = re.sub(r"<.*", "", name)
        if base in ("gather", "scatter"):
            t = re.search(r"<(.*)>", name).group(1)
            base += "16" if "4" in t and "vector" in t.lower() or "uint4" in t else ("8" if "long" in t else "4")
        vals.setdefault(base, {}).update({c: x for c, x in v.items() if c != "launches"})
cols = [("FETCH_SIZE", 1024, "FETCH_SIZE B"), ("TCC_EA0_RDREQ_sum", 1, "RDREQ"), ("TCC_EA0_RDREQ_32B_sum", 1, "RD_32B"), ("TCC_EA0_RDREQ_64B_sum", 1, "RD_64B"), ("TCC_EA0_RDREQ_128B_sum", 1, "RD_128B"),
        ("TCC_EA0_RDREQ_DRAM_sum", 1, "RD_DRAM"), ("TCC_EA0_RDREQ_DRAM_32B_sum", 1, "RD_DRAM_32B"),
        ("WRITE_SIZE", 1024, "WRITE_SIZE B"), ("TCC_EA0_WRREQ_sum", 1, "WRREQ"), ("TCC_EA0_WRREQ_64B_sum", 1, "WR_64B"), ("TCC_EA0_WRREQ_WRITE_DRAM_sum", 1, "WR_DRAM"), ("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", 1, "WR_DRAM_32B"),
        ("TCC_EA0_ATOMIC_sum", 1, "EA_ATOMIC"), ("TCC_EA0_WRREQ_ATOMIC_DRAM_sum", 1, "AT_DRAM"), ("TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum", 1, "AT_DRAM_32B"), ("TCC_ATOMIC_sum", 1, "TCC_ATOMIC"),
        ("TCC_REQ_sum", 1, "TCC_REQ"), ("TCC_HIT_sum", 1, "HIT"), ("TCC_MISS_sum", 1, "MISS")]
print("# per ACCESS (one lane's word, or one wave's 512-byte line): counter values; FETCH_SIZE / WRITE_SIZE are KiB in rocprofv3 -> shown as bytes")
print(f"{'pattern':16s} {'useful B':>8s} | " + " ".join(f"{h:>12s}" for _, _, h in cols))
for name, (n, b) in known.items():
    v = vals.get(name, {})
    print(f"{name:16s} {b:8d} | " + " ".join(f"{(v[c] * s / n):12.3f}" if c in v else f"{'-':>12s}" for c, s, _ in cols))
print("# derived read bytes per access = 32 RD_32B + 64 RD_64B + 128 RD_128B; write bytes = 64 WR_64B + 32 (WRREQ - WR_64B)")
for name, (n, b) in known.items():
    v = vals.get(name, {})
    try:
        rd = (32 * v.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * v.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * v.get("TCC_EA0_RDREQ_128B_sum", 0)) / n
        wr = (64 * v.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (v.get("TCC_EA0_WRREQ_sum", 0) - v.get("TCC_EA0_WRREQ_64B_sum", 0))) / n
        f2 = v.get("FETCH_SIZE", 0) * 1024 / n
        print(f"{name:16s} useful {b:4d} B | read {rd:8.2f} B (FETCH_SIZE says {f2:7.2f}: x {rd / f2 if f2 > 0.5 else float('nan'):.2f}) | written {wr:8.2f} B (WRITE_SIZE says {v.get('WRITE_SIZE', 0) * 1024 / n:7.2f})")
    except Exception as e:
        print(name, "n/a", e)
PY
for p in 1 2 3 4 5 6 7; do grep -i -m2 "error\|invalid\|unknown" /tmp/cal_$p.log; done
