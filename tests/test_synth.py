"""The synthetic workload generator is deterministic and block-independent (every rank and the oracle must see the same bytes)."""
import numpy as np


def test_blocks_are_deterministic_and_independent():
    from seqlib_amd import synth
    g = synth.make_genome(200000)
    assert np.array_equal(g, synth.make_genome(200000)) and g.max() <= 3
    a, s, st = synth.make_reads_block(g, 3, 4096, 150, 44)
    b, _, _ = synth.make_reads_block(g, 3, 4096, 150, 44)
    c, _, _ = synth.make_reads_block(g, 4, 4096, 150, 44)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.shape == (4096, 150) and set(np.unique(a)) <= set(b"ACGT")
    # most reads are near-exact copies of the genome at their start (forward) / reverse complement (reverse)
    codes = np.searchsorted(synth.ACGT, a)
    ok = 0
    for i in range(300):
        r = codes[i] if st[i] == 0 else synth.COMP[codes[i][::-1]]
        ok += (r != g[s[i]:s[i] + 150]).sum() <= 6
    assert ok > 270
