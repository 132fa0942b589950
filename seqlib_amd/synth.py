"""Synthetic references and reads for tests and bench.py (no network: everything is generated).

Recipe = SURVEY.md 8(d) / BASELINE.md 3: reference with fixed seed 20240601, 50 % GC, 2 % of its length
made of copies of random 300-6 000 bp segments at 0-3 % divergence (both orientations), 0.5 % low-complexity
tracts (period 1-6), no N.  Reads: uniform start, strand 50/50, wgsim-like errors mirroring the reference's
own recipe (/root/reference/tests/data/wgsim.sh:27): 0.2 %/base substitutions + 0.1 % mutations of which
15 % are indels with geometric extension p = 0.3.  Generators are numpy PCG64 streams (the survey names
xoshiro256**; any fixed-seed generator serves -- what matters is that every rank and the oracle see the
same bytes).  Reads are produced in independent blocks of BLOCK reads so that any block can be made alone.
"""
import numpy as np

GENOME_SEED = 20240601
BLOCK = 1 << 17            # reads per independently seeded block
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.array([3, 2, 1, 0], dtype=np.uint8)

CONFIGS = {                 # BASELINE.json configs
    "C1": dict(name="plumbing_1kb", length=1000, read_len=100, read_seed=43),
    "C2": dict(name="ecoli_syn", length=4641652, read_len=150, read_seed=44),
    "C3": dict(name="chr20_syn", length=64444167, read_len=150, read_seed=45),
}


def make_genome(length, seed=GENOME_SEED):
    """-> uint8 codes 0..3"""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = rng.integers(0, 4, size=length, dtype=np.uint8)
    # repeats: 2 % of the length
    budget = int(0.02 * length)
    while budget > 300 and length > 20000:
        L = int(rng.integers(300, min(6000, length // 4) + 1))
        src = int(rng.integers(0, length - L))
        dst = int(rng.integers(0, length - L))
        seg = g[src:src + L].copy()
        div = rng.uniform(0.0, 0.03)
        mut = rng.random(L) < div
        seg[mut] = (seg[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
        if rng.random() < 0.5:
            seg = COMP[seg[::-1]]
        g[dst:dst + L] = seg
        budget -= L
    # low complexity: 0.5 % of the length
    budget = int(0.005 * length)
    while budget > 20 and length > 20000:
        L = int(rng.integers(20, 301))
        period = int(rng.integers(1, 7))
        unit = rng.integers(0, 4, size=period, dtype=np.uint8)
        dst = int(rng.integers(0, length - L))
        g[dst:dst + L] = np.resize(unit, L)
        budget -= L
    return g


def genome_ascii(g):
    return ACGT[g].tobytes().decode()


def make_reads_block(genome, block_idx, n, read_len, seed):
    """Reads of block `block_idx` (deterministic, independent of other blocks) -> (uint8 ASCII [n, read_len], start, strand)"""
    rng = np.random.Generator(np.random.PCG64([seed, block_idx]))
    G = len(genome)
    slack = 32
    frag = read_len + slack
    start = rng.integers(0, G - frag, size=n)
    strand = rng.integers(0, 2, size=n, dtype=np.uint8)
    seq = np.lib.stride_tricks.sliding_window_view(genome, frag)[start]      # [n, frag] copy
    # mutations (0.1 %/base): 85 % substitutions, 15 % indels with geometric extension
    n_mut = int(rng.binomial(n * frag, 0.001))
    mpos = rng.integers(0, n * frag, size=n_mut)
    indel = rng.random(n_mut) < 0.15
    flat = seq.reshape(-1)
    sp = mpos[~indel]
    flat[sp] = (flat[sp] + rng.integers(1, 4, size=len(sp), dtype=np.uint8)) & 3
    ip = np.sort(mpos[indel])[::-1]                                         # right to left inside each read
    for p in ip:
        r, c = divmod(int(p), frag)
        ext = int(rng.geometric(0.7))
        row = seq[r]
        if rng.random() < 0.5:                                              # deletion from the read
            row[c:frag - ext] = row[c + ext:].copy()
        else:                                                               # insertion into the read
            ins = rng.integers(0, 4, size=ext, dtype=np.uint8)
            tail = row[c:frag - ext].copy()
            row[c:c + ext] = ins[:max(0, min(ext, frag - c))]
            row[c + ext:] = tail[:max(0, frag - c - ext)]
    seq = seq[:, :read_len].copy()
    rv = strand == 1
    seq[rv] = COMP[seq[rv][:, ::-1]]
    # sequencing errors (0.2 %/base)
    n_err = int(rng.binomial(n * read_len, 0.002))
    ep = rng.integers(0, n * read_len, size=n_err)
    flat = seq.reshape(-1)
    flat[ep] = (flat[ep] + rng.integers(1, 4, size=n_err, dtype=np.uint8)) & 3
    return ACGT[seq], start, strand


def make_reads(genome, n_reads, read_len, seed, first_block=0):
    """-> uint8 ASCII array [n_reads, read_len]; block b covers reads [b*BLOCK, (b+1)*BLOCK)"""
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    done, b = 0, first_block
    while done < n_reads:
        m = min(BLOCK, n_reads - done)
        blk, _, _ = make_reads_block(genome, b, BLOCK, read_len, seed)
        out[done:done + m] = blk[:m]
        done += m
        b += 1
    return out


def offsets_for(n_reads, read_len):
    return (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len))
