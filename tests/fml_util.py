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
