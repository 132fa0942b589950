"""Shared helpers of the FermiAssembler / BFC tests: seeded read simulation from the reference's own fixture genome (tests/golden/tiny.fa)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
COMP = bytes.maketrans(b"ACGTacgtNn", b"TGCAtgcaNn")


def read_fa(path):
    d, name = {}, None
    for line in open(path):
        line = line.strip()
        if line.startswith(">"):
            name = line[1:].split()[0]
            d[name] = []
        elif name:
            d[name].append(line)
    return {k: "".join(v).upper().encode() for k, v in d.items()}


def revcomp(s):
    return s.translate(COMP)[::-1]


def sim_window(genome, n, length=150, err=0.01, seed=1, n_frac=0.0, lower_frac=0.0, qual=True, ragged=False):
    """n reads of `genome` (bytes) with substitution errors; an erroneous base gets a low quality 80 % of the time, a correct one 5 %.
    Returns (seqs, quals or None, truth) as lists of bytes."""
    rng = np.random.default_rng(seed)
    seqs, quals, truth = [], [], []
    for _ in range(n):
        L = length if not ragged else int(rng.integers(max(20, length // 3), length + 1))
        L = min(L, len(genome))
        s = int(rng.integers(0, len(genome) - L + 1))
        t = genome[s:s + L]
        if rng.random() < 0.5:
            t = revcomp(t)
        b = bytearray(t)
        q = bytearray(b"I" * L)
        e = rng.random(L)
        for j in np.nonzero(e < err)[0]:
            b[j] = b"ACGT"[(b"ACGT".index(b[j]) + int(rng.integers(1, 4))) % 4]
            q[j] = ord("#") if rng.random() < 0.8 else ord("I")
        for j in np.nonzero((e >= err) & (e < err + 0.05))[0]:
            q[j] = ord("#")
        if n_frac > 0:
            for j in np.nonzero(rng.random(L) < n_frac)[0]:
                b[j] = ord("N"); q[j] = ord("!")
        if lower_frac > 0 and rng.random() < lower_frac:
            b = bytearray(bytes(b).lower())
        seqs.append(bytes(b)); quals.append(bytes(q)); truth.append(t)
    return seqs, (quals if qual else None), truth


def fixture_genome():
    return read_fa(os.path.join(GOLDEN, "tiny.fa"))


def fixture_reads():
    import os
    out = []
    for name in ("sim1_bcr.head3000.fq", "sim2_bcr.head3000.fq"):
        L = open(os.path.join(GOLDEN, name)).read().split("\n")
        out.append(([L[i + 1].encode() for i in range(0, len(L) - 3, 4)], [L[i + 3].encode() for i in range(0, len(L) - 3, 4)]))
    return out[0][0] + out[1][0], out[0][1] + out[1][1]


def het_genome(g, seed):
    """a second haplotype: a SNP every ~400 bp and a few small indels (bubbles and open bubbles in the graph)"""
    rng = np.random.default_rng(seed)
    b = bytearray(g)
    for p in sorted(rng.integers(200, len(g) - 200, len(g) // 400), reverse=True):
        r = rng.random()
        if r < 0.8:
            b[p] = b"ACGT"[(b"ACGT".index(b[p]) + 1) % 4]
        elif r < 0.9:
            del b[p:p + int(rng.integers(1, 4))]
        else:
            b[p:p] = b"ACGT"[int(rng.integers(0, 4)):][:1] * int(rng.integers(1, 4))
    return bytes(b)


def asm_windows(genome):
    w = []
    w.append(sim_window(genome["bcr"][20000:50000], 8000, seed=7)[:2])                         # clean 40x window
    g2 = genome["abl"][50000:62000]
    a = sim_window(g2, 2000, seed=41)
    b = sim_window(het_genome(g2, 5), 2000, seed=42)
    w.append((a[0] + b[0], a[1] + b[1]))                                                         # two haplotypes
    rep = genome["tp53"][3000:3600]
    g3 = genome["tp53"][0:9000] + rep + genome["tp53"][9000:15000] + revcomp(rep) + genome["tp53"][15000:19000]
    w.append(sim_window(g3, 5000, seed=43, err=0.005)[:2])                                     # a 600 bp repeat in three copies, one inverted
    w.append(fixture_reads())                                                                   # the reference's own fixture reads: 17x, bcr/abl fusion
    w.append(sim_window(genome["myc"][0:3000], 150, seed=44)[:2])                              # too thin to assemble
    return w
