"""prints the per-stream kernel timeline of the last bench step in a trace written by scripts/trace_step.sh"""
import sys
rows = [l.rstrip("\n").rsplit(",", 4) for l in open(sys.argv[1])]
rows = [(n, int(s), int(e), q, st) for n, s, e, q, st in rows]
enc = sorted([r for r in rows if r[0].startswith("k_encode")], key=lambda r: r[1])
nw = len(set(r[4] for r in enc))
start = enc[-nw][1]
sel = [r for r in rows if r[1] >= start - 1e6]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for st in sorted(set(r[4] for r in sel)):
    print("== stream", st)
    for r in sorted([x for x in sel if x[4] == st], key=lambda x: x[1]):
        d = (r[2] - r[1]) / 1e6
        if d > thr:
            print("  %-48s %8.1f -> %8.1f  (%6.1f ms)" % (r[0], (r[1] - start) / 1e6, (r[2] - start) / 1e6, d))
