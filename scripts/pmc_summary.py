"""Summarises the rocprofv3 passes written by scripts/profile_round.sh:
   python scripts/pmc_summary.py <out_dir> <reads_per_launch> > profiles/rNN_pmc_summary.json
Per kernel: mean counter values over the full-size launches (the largest launches of each kernel).  FETCH_SIZE /
WRITE_SIZE are reported by rocprofv3 in KB."""
import collections, csv, glob, json, sys

out_dir, reads = sys.argv[1], float(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out_dir + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if "rocprim" in name or "rocclr" in name or "hipcub" in name:
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
kern = {}
for k, d in acc.items():
    kern[k] = {}
    for c, v in d.items():
        big = [x for x in v if x >= 0.5 * max(v)] or v       # full-size launches only
        kern[k][c] = sum(big) / len(big)
def g(k, c):
    return kern.get(k, {}).get(c, 0.0)
seed = ["k_seed12", "k_seed3"]
fetch = sum(g(k, "FETCH_SIZE") for k in seed) * 1024.0 / reads
write = sum(g(k, "WRITE_SIZE") for k in seed) * 1024.0 / reads
res = {
    "command": "rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --no-cpu-baseline --verify 0 --steps 2 (one pass per counter set; scripts/profile_round.sh)",
    "note": "per-launch means over the full-size launches (3 workers x reads/3 each); FETCH_SIZE/WRITE_SIZE in KB as rocprofv3 reports them; on gfx950 "
            "FETCH_SIZE is known to under-report wide coalesced streams by 2x (MI355X_MICROARCH.md) -- the seeding kernels issue 32-byte random reads, "
            "for which the counter is uncalibrated, so the raw value is reported",
    "reads_per_launch": reads,
    "seed_fetch_bytes_per_read": fetch, "seed_write_bytes_per_read": write,
    "seed_l2_hit_rate": (g("k_seed12", "TCC_HIT_sum") / g("k_seed12", "TCC_REQ_sum")) if g("k_seed12", "TCC_REQ_sum") else None,
    "seed_wait_frac_of_wave_cycles": (g("k_seed12", "SQ_WAIT_ANY") / g("k_seed12", "SQ_WAVE_CYCLES")) if g("k_seed12", "SQ_WAVE_CYCLES") else None,
    "seed_valu_lane_utilisation": (g("k_seed12", "SQ_THREAD_CYCLES_VALU") / (64.0 * g("k_seed12", "SQ_ACTIVE_INST_VALU"))) if g("k_seed12", "SQ_ACTIVE_INST_VALU") else None,
    "extend_valu_lane_utilisation": (g("k_extend_reg<160>", "SQ_THREAD_CYCLES_VALU") / (64.0 * g("k_extend_reg<160>", "SQ_ACTIVE_INST_VALU")))
                                    if g("k_extend_reg<160>", "SQ_ACTIVE_INST_VALU") else None,
    "kernels": kern,
}
print(json.dumps(res, indent=1))
