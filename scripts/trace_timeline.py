"""Coarse timeline of a rocprofv3 --kernel-trace CSV: python scripts/trace_timeline.py <kernel_trace.csv> [bin_ms]
Per queue, which kernel covers most of each time bin over the last (timed) step, plus per-kernel stretched durations."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
bin_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
ev = []
for r in rows:
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "")
    nm = nm.split("<")[0] if not nm.startswith("k_regs_wave") else nm[:22]
    if "rocprim" in nm or "hipcub" in nm: nm = "cub"
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm, r.get("Queue_Id", "0")))
ev.sort()
# the last step: everything after the last long gap following a k_compact? simpler: take the last 45 % of seeding launches
seeds = [e for e in ev if e[2].startswith("k_seed12")]
n_last = max(1, len(seeds) // 2)
t0 = seeds[-n_last][0]
ev = [e for e in ev if e[0] >= t0]
t1 = max(e[1] for e in ev)
print("window %.1f ms, %d kernels" % ((t1 - t0) / 1e6, len(ev)))
queues = sorted(set(e[3] for e in ev))
nb = int((t1 - t0) / 1e6 / bin_ms) + 1
for q in queues:
    line = []
    for b in range(nb):
        lo, hi = t0 + b * bin_ms * 1e6, t0 + (b + 1) * bin_ms * 1e6
        cov = collections.Counter()
        for s, e, nm, qq in ev:
            if qq == q and e > lo and s < hi: cov[nm] += min(e, hi) - max(s, lo)
        line.append(cov.most_common(1)[0][0] if cov else ".")
    print("queue", q)
    prev = None; run = 0; out = []
    for x in line + [None]:
        if x == prev: run += 1
        else:
            if prev is not None: out.append("%s x%d" % (prev, run))
            prev, run = x, 1
    print("   " + " | ".join(out))
tot = collections.Counter(); cnt = collections.Counter()
for s, e, nm, q in ev: tot[nm] += e - s; cnt[nm] += 1
for nm, t in tot.most_common(18): print("%-28s n %4d  sum %8.1f ms  avg %7.2f ms" % (nm, cnt[nm], t / 1e6, t / 1e6 / cnt[nm]))
# idle time per queue and its longest gaps
for q in queues:
    qe = sorted((s, e, nm) for s, e, nm, qq in ev if qq == q)
    cov = 0; gaps = []
    end = qe[0][0]; last = qe[0][2]
    for s, e, nm in qe:
        if s > end:
            gaps.append(((s - end) / 1e6, last, nm, (end - t0) / 1e6))
        if e > end:
            cov += e - max(s, end); end = e; last = nm
    print("queue %s: busy %.1f ms of %.1f; gaps > 2 ms: %s" % (q, cov / 1e6, (t1 - t0) / 1e6,
          "; ".join("%.1f ms at %.0f (%s -> %s)" % g for g in sorted(gaps, reverse=True)[:12] if g[0] > 2)))
