#!/bin/bash
# C5 (FermiAssembler window pipeline) profile: rocprofv3 --kernel-trace --stats of the bench command, then separate --pmc passes
# (kernel-trace only) for the HBM traffic of the k-mer counting kernel.
# Usage: scripts/profile_c5.sh <out_dir> [bench args]  ->  <out_dir>/{c5_kernel_stats.csv, c5_bench_under_rocprof.json, c5_pmc_summary.json}
OUT=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --config C5 --no-cpu-baseline --verify 0 --steps 2 --warmup 1 $*"
timeout -s KILL 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BENCH > $OUT/c5_under_rocprof.log 2>&1
grep '^{"metric"' $OUT/c5_under_rocprof.log > $OUT/c5_bench_under_rocprof.json
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/c5_kernel_stats.csv
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  timeout -s KILL 900 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1
  f=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1)
  if [ ! -s "$f" ]; then echo "profile_c5.sh: pass $i left no counter_collection.csv"; tail -5 $OUT/pmc$i.log; fi
done
python3 - $OUT <<'PY' > $OUT/c5_pmc_summary.json
import collections, csv, glob, json, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        name = name.replace("slxw::", "").replace("slx::", "")
        if name.startswith(("k_fml", "k_asm", "k_xseg", "k_ext", "k_regs_wave_long", "k_cig_band", "k_seed12m", "k_chain_coop")):
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
bench = json.loads(open(out + "/c5_bench_under_rocprof.json").read().strip().splitlines()[-1])
kern = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
launches = {k: max(len(v) for v in d.values()) for k, d in acc.items()}
cnt = collections.defaultdict(float)          # the counting kernels together (k_fml_count is the fallback for small batches)
for name in ("k_fml_bin", "k_fml_part", "k_fml_count", "k_fml_pack", "k_fml_starts"):
    for c, v in kern.get(name, {}).items():
        cnt[c] += v
n_reads = bench["config"]["windows_per_gpu"] * bench["config"]["reads_per_window"]
# the realignment half's extension kernels: VALUBusy = SQ_ACTIVE_INST_VALU x 2 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), SUMS over each kernel's launches
# (scripts/pmc_summary.py has the units); lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)
tot = {k: {c: sum(v) for c, v in d.items()} for k, d in acc.items()}
def busy(ks):
    inst = sum(tot[k].get("SQ_ACTIVE_INST_VALU", 0.0) for k in ks); gui = sum(tot[k].get("GRBM_GUI_ACTIVE", 0.0) for k in ks)
    return inst * 2.0 / (1024 * gui / 8) if gui else None
def lanes(ks):
    inst = sum(tot[k].get("SQ_ACTIVE_INST_VALU", 0.0) for k in ks); thr = sum(tot[k].get("SQ_THREAD_CYCLES_VALU", 0.0) for k in ks)
    return thr / (64.0 * inst) if inst else None
ext = [k for k in tot if k.startswith(("k_xseg", "k_ext"))]
res = {
    "command": "bench.py --config C5 --steps 2 --warmup 1 (separate rocprofv3 --pmc passes, --kernel-trace only)",
    "reads_per_launch": n_reads,
    "units": "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them, x 1024 here; FETCH_SIZE is known to under-report wide coalesced streams 2x on gfx950 "
             "(MI355X_MICROARCH.md) -- the counting kernels' traffic is 8-byte scattered item writes and uncontended slot inserts, for which no calibration exists, so the raw value is given",
    "count_fetch_bytes_per_launch": cnt.get("FETCH_SIZE", 0.0) * 1024.0 or None,
    "count_write_bytes_per_launch": cnt.get("WRITE_SIZE", 0.0) * 1024.0 or None,
    "count_fetch_plus_write_bytes_per_launch": (cnt.get("FETCH_SIZE", 0.0) + cnt.get("WRITE_SIZE", 0.0)) * 1024.0 or None,
    "count_algorithmic_bytes_per_launch": 32.0 * (bench.get("roofline_count") or bench["roofline"])["kmers_per_launch"] + 2.0 * (bench.get("roofline_count") or bench["roofline"])["bases_per_launch"],
    "ext_kernels": sorted(ext),
    "ext_valu_busy": busy(ext), "ext_valu_lane_utilisation": lanes(ext),
    "ext_valu_busy_by_kernel": {k: busy([k]) for k in sorted(ext)},
    "valu_busy_by_kernel": {k: busy([k]) for k in sorted(tot) if not k.startswith(("k_fml", "k_asm"))},
    "valu_busy_formula": "SQ_ACTIVE_INST_VALU x 2 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), summed over the kernel's launches; a kernel that is one block on an otherwise idle chip shows ~1/1024 per busy SIMD",
    "kernels": {k: dict(v, launches_seen=launches[k]) for k, v in sorted(kern.items())},
}
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/stats $OUT/pmc[0-9] $OUT/pmc[0-9].log
ls -la $OUT
