#!/bin/bash
# The random-access ceiling of the seeding kernels (scripts/ubench/rand32.hip: chains of dependent random 32-byte reads) at the index
# footprints of the bench configs -> <out>.txt (the tool's lines) and <out>.json (best rate per table size).
# Usage: scripts/ubench_rand32.sh <out_prefix> [MB ...]
OUT=$1; shift
SIZES=${@:-16 64 192 600 4000 9000}
R=$GRAFT_REPO_ROOT
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/scripts/ubench/rand32.hip -o /tmp/rand32 || exit 1
/tmp/rand32 $SIZES > $OUT.txt 2>&1
python3 - $OUT.txt > $OUT.json <<'PY'
import json, re, sys
best = {}
for ln in open(sys.argv[1]):
    m = re.match(r"table\s+(\d+) MB\s+waves/SIMD (\d+)\s+ILP (\d+) :\s+([\d.]+) G reads/s", ln)
    if m:
        mb, rate = int(m.group(1)), float(m.group(4))
        if rate > best.get(mb, {}).get("g_reads_per_s", 0):
            best[mb] = {"g_reads_per_s": rate, "waves_per_simd": int(m.group(2)), "ilp": int(m.group(3))}
print(json.dumps({"what": "scripts/ubench/rand32.hip: dependent random 32-byte reads, best of the (waves/SIMD, ILP) settings per table size",
                  "table_mb": {str(k): v for k, v in sorted(best.items())}}, indent=1))
PY
cat $OUT.json
