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

GRCH38 = [("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555), ("chr5", 181538259), ("chr6", 170805979),
          ("chr7", 159345973), ("chr8", 145138636), ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309),
          ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189), ("chr16", 90338345), ("chr17", 83257441), ("chr18", 80373285),
          ("chr19", 58617616), ("chr20", 64444167), ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415)]

CONFIGS = {                 # BASELINE.json configs
    "C1": dict(name="plumbing_1kb", length=1000, read_len=100, read_seed=43),
    "C2": dict(name="ecoli_syn", length=4641652, read_len=150, read_seed=44),
    "C3": dict(name="chr20_syn", length=64444167, read_len=150, read_seed=45, pairs=True),
    # full GRCh38-sized reference: 24 contigs with the primary assembly's lengths (3.09 Gbp, 6.2 G BWT symbols)
    "C4": dict(name="grch38_syn", length=sum(l for _, l in GRCH38), read_len=150, read_seed=46, contigs=[(n + "_syn", l) for n, l in GRCH38]),
    # the first 14 contigs of it (2.30 Gbp, 4.6 G BWT symbols: past 2^32, so the u64 index path): what the C4 parity test and the
    # C4-scale bench line run on one GPU
    "C4h": dict(name="grch38_syn_chr1-14", length=sum(l for _, l in GRCH38[:14]), read_len=150, read_seed=46, contigs=[(n + "_syn", l) for n, l in GRCH38[:14]]),
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


def genome_ascii_bytes(g):
    """ASCII as bytes (no str round trip: a 2 Gbp contig would be copied twice more)"""
    return ACGT[g].tobytes()


def make_reference(cfg):
    """-> [(contig name, uint8 codes)]: one contig named after the config, or cfg["contigs"] = [(name, length)] with one
    generator stream per contig (seed = GENOME_SEED + contig ordinal) so that any prefix of the list can be made alone"""
    if "contigs" not in cfg:
        return [(cfg["name"], make_genome(cfg["length"]))]
    return [(nm, make_genome(L, seed=GENOME_SEED + i)) for i, (nm, L) in enumerate(cfg["contigs"])]


def _mutate(seq, n, frag, read_len, strand, rng):
    """wgsim-like edits of `seq` [n, frag] in place -> ASCII [n, read_len] (shared by every read generator below)"""
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
    return ACGT[seq]


def make_pair_block(genome, block_idx, n, read_len, seed):
    """Paired reads of block `block_idx` as SURVEY 8d defines them for C3 -- the API has no paired-end mode, so a pair is two
    independent single-end reads: mates 2k, 2k+1 come from one fragment of 300 +- 30 bp, opposite strands (which mate is
    forward alternates at random).  -> (uint8 ASCII [n, read_len], start, strand), n even."""
    rng = np.random.Generator(np.random.PCG64([seed, block_idx, 2]))
    G = len(genome)
    slack = 32
    frag = read_len + slack
    npair = n // 2
    isz = np.clip(np.rint(rng.normal(300.0, 30.0, size=npair)).astype(np.int64), read_len, 450)
    fstart = rng.integers(0, G - 450 - frag, size=npair)
    flip = rng.integers(0, 2, size=npair, dtype=np.uint8)
    start = np.empty(n, dtype=np.int64)
    strand = np.empty(n, dtype=np.uint8)
    # the forward mate starts at the fragment's start; the reverse mate is the reverse complement of the window that ENDS at the
    # fragment's end (its window starts slack bases earlier so that deletions still leave read_len bases after trimming)
    start[0::2] = np.where(flip == 0, fstart, fstart + isz - read_len)
    start[1::2] = np.where(flip == 0, fstart + isz - read_len, fstart)
    strand[0::2] = flip
    strand[1::2] = 1 - flip
    seq = np.lib.stride_tricks.sliding_window_view(genome, frag)[start]
    return _mutate(seq, n, frag, read_len, strand, rng), start, strand


def make_multi_block(genomes, cum, block_idx, n, read_len, seed):
    """Single-end reads over several contigs: uniform start on a contig picked with probability proportional to its length
    (SURVEY 8d).  genomes = list of code arrays, cum = cumulative lengths (len + 1 entries)."""
    rng = np.random.Generator(np.random.PCG64([seed, block_idx, 3]))
    slack = 32
    frag = read_len + slack
    pos = rng.integers(0, int(cum[-1]), size=n)
    cid = np.searchsorted(cum, pos, side="right") - 1
    strand = rng.integers(0, 2, size=n, dtype=np.uint8)
    seq = np.empty((n, frag), dtype=np.uint8)
    start = np.empty(n, dtype=np.int64)
    for c in np.unique(cid):
        sel = np.nonzero(cid == c)[0]
        g = genomes[int(c)]
        st = np.minimum(pos[sel] - int(cum[c]), len(g) - frag)          # keep the window inside its contig
        start[sel] = st
        seq[sel] = np.lib.stride_tricks.sliding_window_view(g, frag)[st]
    return _mutate(seq, n, frag, read_len, strand, rng), start, strand


def make_config_block(cfg, refs, block_idx, n=None):
    """block `block_idx` of the read set of a BASELINE config -> uint8 ASCII [n, read_len]"""
    n = BLOCK if n is None else n
    if "contigs" in cfg:
        genomes = [g for _, g in refs]
        cum = np.concatenate([[0], np.cumsum([len(g) for g in genomes])]).astype(np.int64)
        return make_multi_block(genomes, cum, block_idx, n, cfg["read_len"], cfg["read_seed"])[0]
    if cfg.get("pairs"):
        return make_pair_block(refs[0][1], block_idx, n, cfg["read_len"], cfg["read_seed"])[0]
    return make_reads_block(refs[0][1], block_idx, n, cfg["read_len"], cfg["read_seed"])[0]


def make_config_reads(cfg, refs, n_reads, first_block=0):
    """the first n_reads reads (from block first_block on) of a config's read set"""
    out = np.empty((n_reads, cfg["read_len"]), dtype=np.uint8)
    done, b = 0, first_block
    while done < n_reads:
        m = min(BLOCK, n_reads - done)
        out[done:done + m] = make_config_block(cfg, refs, b, BLOCK if n_reads >= BLOCK else max(2, (n_reads + 1) // 2 * 2))[:m]
        done += m
        b += 1
    return out


def make_reads_block(genome, block_idx, n, read_len, seed):
    """Reads of block `block_idx` (deterministic, independent of other blocks) -> (uint8 ASCII [n, read_len], start, strand)"""
    rng = np.random.Generator(np.random.PCG64([seed, block_idx]))
    G = len(genome)
    slack = 32
    frag = read_len + slack
    start = rng.integers(0, G - frag, size=n)
    strand = rng.integers(0, 2, size=n, dtype=np.uint8)
    seq = np.lib.stride_tricks.sliding_window_view(genome, frag)[start]      # [n, frag] copy
    return _mutate(seq, n, frag, read_len, strand, rng), start, strand


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
