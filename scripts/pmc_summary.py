"""Summarises the rocprofv3 passes written by scripts/profile_round.sh:
   python scripts/pmc_summary.py <out_dir> <config> > profiles/rNN_pmc_summary.json
   python scripts/pmc_summary.py <summary.json> <config>      recomputes the derived figures (stage groups) from the per-kernel counters of an earlier summary
Per kernel: mean counter values over the full-size launches (the largest launches of each kernel).

Units, as found on this gfx950 / ROCm 7.2 (cross-checked between counters of the same launches):
  FETCH_SIZE / WRITE_SIZE   KB as rocprofv3 reports them.  FETCH_SIZE under-reports wide coalesced streams by 2x on gfx950
                            (MI355X_MICROARCH.md); the seeding kernels issue 32-byte random reads, for which the counter is
                            uncalibrated, so the raw value is reported.
  SQ_ACTIVE_INST_VALU       one count per wave64 VALU instruction issued (it equals SQ_INSTS_VALU to within 3 % on every kernel
                            here); the guide lists it in "quad-cycles".  A 32-bit wave64 VALU op occupies a SIMD-32 for 2 cycles
                            (guide: v_fma_f32 wave64 = 2 cyc).
  GRBM_GUI_ACTIVE           cycles, SUMMED over the 8 XCDs (GRBM_GUI_ACTIVE / 8 / kernel wall time = 2.3-2.4 GHz);
                            SQ_BUSY_CYCLES is summed over the 32 shader engines (= 4 x GRBM_GUI_ACTIVE).
  VALUBusy (derived here)   SQ_ACTIVE_INST_VALU x 2 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the fraction of SIMD issue cycles this
                            kernel's VALU instructions filled while it was resident.  With three workers' kernels overlapping on
                            the GPU the same cycles also carry the other kernels' instructions, so the values of concurrently
                            running kernels add up.  (The gfx94x formula of rocprof, x4 instead of x2, gives values above 100 % here.)
  lane utilisation          SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): active lanes per VALU instruction.
"""
import collections, csv, glob, json, sys

out_dir, config = sys.argv[1], sys.argv[2]
N_SIMD, N_XCD = 1024, 8
acc = collections.defaultdict(lambda: collections.defaultdict(list))
prior = json.load(open(out_dir)) if out_dir.endswith(".json") else None
for f in ([] if prior else glob.glob(out_dir + "/pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if "rocprim" in name or "rocclr" in name or "hipcub" in name:
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
kern, n_launch = {}, {}
for k, d in acc.items():
    kern[k] = {}
    for c, v in d.items():
        big = [x for x in v if x >= 0.5 * max(v)] or v       # full-size launches only
        kern[k][c] = sum(big) / len(big)
        n_launch[k] = max(n_launch.get(k, 0), len(big))
if prior:
    kern = prior["kernels"]


def g(k, c):
    return kern.get(k, {}).get(c, 0.0)


def pick(prefix):
    return [k for k in kern if k.startswith(prefix)]


def valu_busy(k):
    gui = g(k, "GRBM_GUI_ACTIVE")
    return g(k, "SQ_ACTIVE_INST_VALU") * 2.0 / (N_SIMD * gui / N_XCD) if gui else None


def lane_util(k):
    a = g(k, "SQ_ACTIVE_INST_VALU")
    return g(k, "SQ_THREAD_CYCLES_VALU") / (64.0 * a) if a else None


try:
    if prior:
        bench, reads = None, prior["reads_per_launch"]
    else:
        bench = json.loads(open(out_dir + "/bench_under_rocprof.json").read().strip().splitlines()[-1])
        reads = bench["reads_per_seed_launch"]
except Exception:
    bench, reads = None, {"C2": 10e6 / 3, "C3": 50e6 / 6}.get(config, 1.0)
seed12 = ([k for k in pick("k_seed12") if ", 1>" in k] or pick("k_seed12") or ["k_seed12m"])[0]      # pass 1 (the bulk); pass 2 is k_seed12m<.., 2>
seed12_p2 = ([k for k in pick("k_seed12") if ", 2>" in k] or [None])[0]
seed_all = pick("k_seed")                     # k_seed12m, k_seed3m, k_seed_epi
fetch = sum(g(k, "FETCH_SIZE") for k in seed_all) * 1024.0 / reads
write = sum(g(k, "WRITE_SIZE") for k in seed_all) * 1024.0 / reads
ext = [k for k in kern if k.split("<")[0] in ("k_extend_cand", "k_ext_first", "k_first_lanes", "k_first_bin_count", "k_first_bin_scatter", "k_ext_replay", "k_extend_reg", "k_first_prep", "k_first_diag", "k_ext_lanes", "k_cand_lane_prep")]
ext_inst = sum(g(k, "SQ_ACTIVE_INST_VALU") for k in ext)
ext_gui = sum(g(k, "GRBM_GUI_ACTIVE") for k in ext)
def wait_frac(k):
    w = g(k, "SQ_WAVE_CYCLES")
    return g(k, "SQ_WAIT_ANY") / w if w else None


def group(names):
    """instruction-weighted VALUBusy / lane utilisation / waiting share over the kernels of a stage (full-size launches)"""
    ks = [k for k in kern if k.split("<")[0] in names]
    inst = sum(g(k, "SQ_ACTIVE_INST_VALU") for k in ks)
    gui = sum(g(k, "GRBM_GUI_ACTIVE") for k in ks)
    thr = sum(g(k, "SQ_THREAD_CYCLES_VALU") for k in ks)
    wav = sum(g(k, "SQ_WAVE_CYCLES") for k in ks)
    return {"kernels": ks, "valu_busy": (inst * 2.0 / (N_SIMD * gui / N_XCD)) if gui else None, "valu_lane_utilisation": thr / (64.0 * inst) if inst else None,
            "wait_frac_of_wave_cycles": sum(g(k, "SQ_WAIT_ANY") for k in ks) / wav if wav else None,
            "fetch_bytes_per_read": sum(g(k, "FETCH_SIZE") for k in ks) * 1024.0 / reads, "write_bytes_per_read": sum(g(k, "WRITE_SIZE") for k in ks) * 1024.0 / reads,
            "by_kernel": {k: {"valu_busy": valu_busy(k), "valu_lane_utilisation": lane_util(k), "wait_frac_of_wave_cycles": wait_frac(k)} for k in ks}}


res = {
    "config": config,
    "command": "rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --config %s --no-cpu-baseline --no-extras --verify 0 --steps 2 --warmup 1 "
               "(one pass per counter set; scripts/profile_round.sh)" % config,
    "note": "per-launch means over the full-size launches; see the module docstring of scripts/pmc_summary.py for units and formulas",
    "valu_busy_formula": "SQ_ACTIVE_INST_VALU x 2 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)",
    "reads_per_launch": reads,
    "seed_fetch_bytes_per_read": fetch, "seed_write_bytes_per_read": write,
    "seed_l2_hit_rate": (g(seed12, "TCC_HIT_sum") / g(seed12, "TCC_REQ_sum")) if g(seed12, "TCC_REQ_sum") else None,
    "seed_wait_frac_of_wave_cycles": (g(seed12, "SQ_WAIT_ANY") / g(seed12, "SQ_WAVE_CYCLES")) if g(seed12, "SQ_WAVE_CYCLES") else None,
    "seed_valu_lane_utilisation": lane_util(seed12), "seed_valu_busy": valu_busy(seed12), "seed_lds_instructions": g(seed12, "SQ_INSTS_LDS"),
    "seed_pass2_valu_lane_utilisation": lane_util(seed12_p2) if seed12_p2 else None,
    # extension family: instruction-weighted over its kernels' full-size launches
    "ext_valu_busy": (ext_inst * 2.0 / (N_SIMD * ext_gui / N_XCD)) if ext_gui else None,
    "ext_valu_busy_by_kernel": {k: valu_busy(k) for k in ext},
    "ext_valu_lane_utilisation_by_kernel": {k: lane_util(k) for k in ext},
    # the other stages of a step (bench.py roofline_chain / roofline_fin)
    "chain": group(("k_chain", "k_chain_coop", "k_part_flags", "k_part_scatter", "k_heavy_keys")),
    "regions": group(("k_regs1", "k_regs", "k_regs_wave", "k_part_flags_nreg")),
    "hits": group(("k_hits", "k_hits_sam")),
    "cigar": group(("k_cig_fast", "k_cig_fast_coop", "k_cig_lanes", "k_cig_dp")),
    "kernels": kern,
}
print(json.dumps(res, indent=1))
