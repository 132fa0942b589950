"""The C++ drop-in headers (include/SeqLib/*.h): compiled with g++ against libseqlib_amd.so and driven the way
a SeqLib user would (tests/cpp/seqlib_api_test.cpp).  CPU part mirrors the reference's own
tests/test_BamRecord.cpp:9-66 and the setter / accessor checks of seq_test/seq_test.cpp:798-883; the GPU part
checks per-read alignSequence and batch alignSequences against the committed golden records."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "seqlib_amd", "libseqlib_amd.so")):
        g.build()
    out = str(tmp_path_factory.mktemp("cpp") / "seqlib_api_test")
    lib = os.path.join(ROOT, "seqlib_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "seqlib_api_test.cpp"), "-o", out, "-L" + lib, "-lseqlib_amd",
                           "-Wl,-rpath," + lib, "-lz", "-lpthread"])
    return out


def test_cpp_headers_cpu(exe, golden_dir, tmp_path):
    r = subprocess.run([exe, "cpu", os.path.join(golden_dir, "tiny.fa"), str(tmp_path / "rt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cpu checks OK" in r.stdout


@pytest.mark.gpu
def test_cpp_align_matches_golden(exe, golden_dir):
    n1, n2 = 40, 960
    r = subprocess.run([exe, "gpu", os.path.join(golden_dir, "tiny.fa"), os.path.join(golden_dir, "sim1_bcr.head3000.fq"), str(n1), str(n2)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [l.split("\t") for l in r.stdout.strip().split("\n")]
    exp = [l.rstrip("\n").split("\t") for l in open(os.path.join(golden_dir, "sim1_head3000.records.tsv")) if int(l.split("\t")[0]) < n1 + n2]
    assert [g[:10] for g in got] == exp
    # record materialisation: forward-strand records carry the read itself; reverse ones the reference's A<->T-only reversal
    from oracle import orc
    _, seqs = orc.read_fastq(os.path.join(golden_dir, "sim1_bcr.head3000.fq"), n1 + n2)
    for g in got:
        s = seqs[int(g[0])]
        if int(g[2]) & 16:
            assert g[10] == s[::-1].translate(str.maketrans("AT", "TA"))
        else:
            assert g[10] == s
    assert "1 record(s), qname name" in r.stderr


@pytest.mark.gpu
def test_cpp_one_aligner_shared_by_threads(exe, golden_dir):
    """one const BWAAligner called from 6 host threads at once (the reference's alignSequence is const and re-entrant):
    every read comes out as in a serial pass"""
    r = subprocess.run([exe, "threads", os.path.join(golden_dir, "tiny.fa"), os.path.join(golden_dir, "sim1_bcr.head3000.fq"), "400", "6"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches=0" in r.stdout


@pytest.mark.gpu
def test_cpp_multi_device_handle_and_chunked_pipeline(exe, golden_dir):
    """SeqLib::BWAAligner over a group handle (every visible GPU, each listed twice: slx_aligner_create with n_dev > 1 shards the
    batch by contiguous read ranges) with the batch cut into 257-read chunks, so that the pack / align / build pipeline of
    alignSequences runs many rounds: every BamRecord byte-identical to the one-device, one-chunk pass"""
    r = subprocess.run([exe, "multidev", os.path.join(golden_dir, "tiny.fa"), os.path.join(golden_dir, "sim1_bcr.head3000.fq"), "3000", "2"],
                       env=dict(os.environ, SEQLIB_AMD_SLAB_MIN_READS="1"),          # records out of shared slabs, as large batches build them
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches=0" in r.stdout


@pytest.mark.gpu
def test_cpp_bwa_mem_records(exe, golden_dir):
    """BWAAligner::UseBwaMemRecords: the records `bwa mem` would print (opt->T, 0x800, mapq cap, XS:i, XA:Z through the reference's
    own branch at src/BWAAligner.cpp:240, SA:Z; an unmapped record for a read without one) -- tag strings as the oracle's
    restatement of mem_gen_alt / mem_aln2sam builds them"""
    from oracle import orc
    n = 3000
    fq = os.path.join(golden_dir, "sim2_bcr.head3000.fq")
    r = subprocess.run([exe, "bwamem", os.path.join(golden_dir, "tiny.fa"), fq, str(n)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, SEQLIB_AMD_SLAB_MIN_READS="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    got = {}
    for l in r.stdout.strip().split("\n"):
        f = l.split("\t")
        got.setdefault(int(f[0]), []).append(f[2:])
    oidx = orc.Index.load(os.path.join(golden_dir, "tiny.fa"))
    _, seqs = orc.read_fastq(fq, n)
    seqs.append("ACGTACGTACGTTGCATGCATGCAAACCGGTT")
    n_sa = n_xa = n_md = 0
    for i, sq in enumerate(seqs):
        recs = [e for e in orc.align_sequence_sam(orc.default_opt(), oidx, sq, ordinal=i) if e["XS"] >= 0]
        if not recs:                                   # bwa prints an unmapped record
            assert got[i] == [["4", "-1", "-1", "0", "*", "0", "-1", "0", "0", "*", "*", "*"]], (i, got[i])
            continue
        exp = [[str(e["flag"]), str(e["rid"]), str(e["pos"]), str(e["mapq"]), orc.cigar_str(e["cigar"]), str(e["AS"]), str(e["NM"]), str(e["NA"]),
                str(e["XS"]), e["XA"] or "*", e["SA"] or "*", e["MD"] or "*"] for e in recs]
        assert got[i] == exp, (i, got[i], exp)
        n_md += sum(1 for e in recs if e["MD"] and not e["MD"].isdigit())
        n_sa += any(e["SA"] for e in recs)
        n_xa += any(e["XA"] for e in recs)
    assert n_sa >= 1 and n_md > 500 and len(got) == len(seqs)          # (wgsim's reads carry mismatches and indels: MD strings with letters and ^)


# ---------------------------------------------------------------------------------------------- FastqReader / BamWriter
def _kseq_like(text):
    """Independent statement of the record grammar FastqReader follows (bwa kseq.h, see include/SeqLib/FastqReader.h),
    on an in-memory byte string, for well-formed input."""
    recs, lines, i = [], text.split(b"\n"), 0
    strip = lambda l: l[:-1] if len(l) > 1 and l.endswith(b"\r") else l
    while i < len(lines):
        if not lines[i][:1] in (b">", b"@"):
            i += 1
            continue
        head = lines[i][1:]
        i += 1
        m = len(head)
        for k, ch in enumerate(head):
            if chr(ch).isspace():
                m = k
                break
        name, com = head[:m], strip(head[m + 1:]) if m < len(head) else b""
        seq = b""
        while i < len(lines) and lines[i][:1] not in (b">", b"@", b"+"):
            seq = strip(seq + lines[i]) if lines[i] else seq
            i += 1
        qual = b""
        if i < len(lines) and lines[i][:1] == b"+":
            i += 1
            while i < len(lines) and len(qual) < len(seq):
                qual = strip(qual + lines[i])
                i += 1
        recs.append((name, com, seq, qual))
    return recs


def _dump(exe, path):
    r = subprocess.run([exe, "fastq", path], capture_output=True)
    assert r.returncode == 0, r.stderr
    return [tuple(l.split(b"\t")) for l in r.stdout.split(b"\n")[:-1]]


def test_fastq_reader(exe, golden_dir, tmp_path):
    import gzip
    fq = os.path.join(golden_dir, "sim1_bcr.head3000.fq")
    raw = open(fq, "rb").read()
    exp = _kseq_like(raw)
    assert len(exp) == 3000 and all(len(s) == len(q) == 150 for _, _, s, q in exp)
    assert _dump(exe, fq) == exp
    gz = str(tmp_path / "r.fq.gz")
    with gzip.open(gz, "wb") as f:
        f.write(raw)
    assert _dump(exe, gz) == exp                                   # zlib stream, as gzopen in the reference
    # FASTA, multi-line, read through the same parser (no quality: Qual keeps its previous value = empty here)
    fa = os.path.join(golden_dir, "tiny.fa")
    got = _dump(exe, fa)
    exp_fa = _kseq_like(open(fa, "rb").read())
    assert got == exp_fa and [g[0] for g in got] == [b"bcr", b"abl", b"tp53", b"myc"] and len(got[1][2]) == 178633
    # odd but legal inputs: comments, CRLF, blank lines, leading junk, multi-line FASTQ, '@' opening a quality line
    tricky = (b"junk before the first record\n"
              b"@r1 first comment here\nACGT\nAC\n+\nII@I\nII\n"
              b"@r2\r\nGGCC\r\n+r2\r\n@@@@\r\n"
              b"\n\n>fa1\tdesc with space\nAAAA\n\nCCCC\n"
              b"@r3 c3\nTTTT\n+\nIIII")
    p = str(tmp_path / "tricky.fq")
    open(p, "wb").write(tricky)
    got = _dump(exe, p)
    # the reference assigns a field only when kseq has a buffer for it; kseq clears lengths, not buffers, so after r2 the
    # FASTA record fa1 gets Qual == "" (buffer exists, length 0) rather than keeping r2's qualities
    assert got == [(b"r1", b"first comment here", b"ACGTAC", b"II@III"), (b"r2", b"", b"GGCC", b"@@@@"),
                   (b"fa1", b"desc with space", b"AAAACCCC", b""),
                   (b"r3", b"c3", b"TTTT", b"IIII")]
    # truncated quality => the record is refused and the stream ends (kseq_read() == -2)
    p2 = str(tmp_path / "trunc.fq")
    open(p2, "wb").write(b"@ok\nACGT\n+\nIIII\n@bad\nACGT\n+\nII\n")
    assert _dump(exe, p2) == [(b"ok", b"", b"ACGT", b"IIII")]
    # missing file: Open() == false
    assert subprocess.run([exe, "fastq", str(tmp_path / "nope.fq")], capture_output=True).returncode == 3


def _parse_bam(path):
    import gzip, struct
    raw = open(path, "rb").read()
    assert raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))     # BGZF EOF block
    # every BGZF member declares its own size in the BC extra field
    off, n_blocks = 0, 0
    while off < len(raw):
        assert raw[off:off + 4] == b"\x1f\x8b\x08\x04" and raw[off + 12:off + 16] == b"BC\x02\x00"
        off += struct.unpack_from("<H", raw, off + 16)[0] + 1
        n_blocks += 1
    assert off == len(raw)
    d = gzip.decompress(raw)
    assert d[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", d, 4)
    text = d[8:8 + l_text].decode()
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", d, p); p += 4
    refs = []
    for _ in range(n_ref):
        l, = struct.unpack_from("<i", d, p); p += 4
        nm = d[p:p + l - 1].decode(); p += l
        ln, = struct.unpack_from("<i", d, p); p += 4
        refs.append((nm, ln))
    recs = []
    while p < len(d):
        bs, tid, pos, l_name, mapq, bin_, n_cig, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiiBBHHHiiii", d, p)
        q = p + 36
        name = d[q:q + l_name - 1].decode(); q += l_name
        cig = struct.unpack_from("<%dI" % n_cig, d, q); q += 4 * n_cig
        sq = d[q:q + (l_seq + 1) // 2]; q += (l_seq + 1) // 2
        seq = "".join("=ACMGRSVTWYHKDBN"[(sq[i >> 1] >> (0 if i & 1 else 4)) & 15] for i in range(l_seq))
        ql = d[q:q + l_seq]; q += l_seq
        qual = "*" if (l_seq == 0 or ql[0] == 0xff) else "".join(chr(c + 33) for c in ql)
        tags, end = [], p + 4 + bs
        while q < end:
            tg, ty = d[q:q + 2].decode(), chr(d[q + 2]); q += 3
            if ty == "i":
                tags.append("%s:i:%d" % (tg, struct.unpack_from("<i", d, q)[0])); q += 4
            elif ty == "Z":
                e = d.index(b"\0", q)
                tags.append("%s:Z:%s" % (tg, d[q:e].decode())); q = e + 1
            else:
                raise AssertionError("unexpected aux type " + ty)
        assert q == end
        cigar = "".join("%d%s" % (c >> 4, "MIDNSHP=XB"[c & 15]) for c in cig) or "*"
        recs.append(dict(name=name, flag=flag, tid=tid, pos=pos, mapq=mapq, cigar=cigar, mtid=mtid, mpos=mpos, tlen=tlen, seq=seq or "*",
                         qual=qual, tags=tags, bin=bin_, cig=cig))
        p = end
    return text, refs, recs, n_blocks


def _reg2bin(beg, end):
    end -= 1
    for sh, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> sh == end >> sh:
            return off + (beg >> sh)
    return 0


def _sam_of(rec, refs):
    rn = lambda t: "*" if t < 0 else refs[t][0]
    rnext = "*" if rec["mtid"] < 0 else ("=" if rec["mtid"] == rec["tid"] else rn(rec["mtid"]))
    return "\t".join([rec["name"], str(rec["flag"]), rn(rec["tid"]), str(rec["pos"] + 1), str(rec["mapq"]), rec["cigar"], rnext, str(rec["mpos"] + 1),
                      str(rec["tlen"]), rec["seq"], rec["qual"]] + rec["tags"])


def test_bam_writer(exe, tmp_path):
    n = 3001
    pre = str(tmp_path / "w")
    r = subprocess.run([exe, "writer", pre, str(n)], capture_output=True, text=True)
    assert r.returncode == 0 and "writer checks OK" in r.stdout, r.stdout + r.stderr
    hdr = "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chrA\tLN:1000\n@SQ\tSN:chrB\tLN:50000000\n@PG\tID:test\n"
    sam = open(pre + ".sam").read()
    assert sam.startswith(hdr)
    lines = sam[len(hdr):].split("\n")[:-1]
    text, refs, recs, n_blocks = _parse_bam(pre + ".bam")
    assert text == hdr and refs == [("chrA", 1000), ("chrB", 50000000)]
    assert n_blocks > 4                                             # header block, several record blocks, EOF
    assert len(lines) == n and len(recs) == n + 1                   # the record with tid 7 cannot be named in SAM
    assert [_sam_of(x, refs) for x in recs[:n]] == lines            # the two encodings carry the same records
    # spot values written by tests/cpp/seqlib_api_test.cpp:writer_checks
    f = lines[0].split("\t")
    assert f[:9] == ["r0", "0", "chrA", "1", "60", "40M", "*", "0", "0"] and f[10] == "*" and f[11:] == ["NA:i:0", "NM:i:0", "AS:i:40", "XA:Z:chrA,+0,40M,0;"]
    f = lines[1].split("\t")
    assert f[:9] == ["r1", "17", "chrB", "16412", "13", "5S31M2D3I2H", "=", "16712", "-350"] and len(f[9]) == 41 == len(f[10])
    assert lines[2].split("\t")[5:8] == ["10M100N32M", "chrA", "8"]
    assert lines[3].split("\t")[:9] == ["r3", "4", "*", "0", "0", "*", "*", "0", "0"]
    for x in recs:                                                  # bin = reg2bin(pos, end) per the BAM specification
        span = sum(c >> 4 for c in x["cig"] if (c & 15) in (0, 2, 3, 7, 8))
        assert x["bin"] == _reg2bin(x["pos"], x["pos"] + (span or 1)), x


@pytest.mark.gpu
def test_cpp_fastq_to_sam_pipeline(exe, golden_dir, tmp_path):
    """FastqReader -> BWAAligner::alignSequences -> BamWriter, as a SeqLib user strings them together; the SAM/BAM
    content is checked against the committed golden records of the reference algorithm."""
    n = 1500
    pre = str(tmp_path / "aln")
    fq = os.path.join(golden_dir, "sim1_bcr.head3000.fq")
    r = subprocess.run([exe, "pipeline", os.path.join(golden_dir, "tiny.fa"), fq, str(n), pre], capture_output=True, text=True,
                       env=dict(os.environ, SEQLIB_AMD_SLAB_MIN_READS="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    sam = open(pre + ".sam").read().split("\n")[:-1]
    head, body = [l for l in sam if l.startswith("@")], [l for l in sam if not l.startswith("@")]
    assert [h.split("\t")[1] for h in head if h.startswith("@SQ")] == ["SN:bcr", "SN:abl", "SN:tp53", "SN:myc"]
    exp = [l.rstrip("\n").split("\t") for l in open(os.path.join(golden_dir, "sim1_head3000.records.tsv")) if int(l.split("\t")[0]) < n]
    from oracle import orc
    names, seqs = orc.read_fastq(fq, n)
    chrom = ["bcr", "abl", "tp53", "myc"]
    assert len(body) == len(exp)
    for l, e in zip(body, exp):
        f = l.split("\t")
        rd = int(e[0])
        assert f[0] == names[rd] and f[1] == e[2] and f[2] == chrom[int(e[3])] and int(f[3]) == int(e[4]) + 1 and f[4] == e[5] and f[5] == e[6]
        assert f[6:9] == ["*", "0", "0"] and f[10] == "*"
        assert f[11:] == ["NA:i:" + e[9], "NM:i:" + e[8], "AS:i:" + e[7]]
    text, refs, recs, _ = _parse_bam(pre + ".bam")
    assert [r_[0] for r_ in refs] == chrom and text == "\n".join(head) + "\n"
    assert [_sam_of(x, refs) for x in recs] == body


@pytest.mark.gpu
@pytest.mark.parametrize("hardclip,slabs", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_cpp_record_blobs_match_oracle(exe, golden_dir, tmp_path, hardclip, slabs):
    """(slabs = 1: the batch path's records carved out of shared slabs -- include/SeqLib/BamRecord.h, detail::Slab -- as every batch of 8 192 reads or more builds them;
    forced here on a small batch.)
    The BamRecord data block byte for byte (src/BWAAligner.cpp:151-248): qname, CIGAR with op 3 rewritten to S / H, the hard-clip trimming
    of the sequence (:164-177), the case-sensitive 4-bit packing (a lower-case base packs as 15, :214-231), N bases, the A<->T-only reversal of
    reverse-strand reads, qual[0] = 0xff, tags NA, NM, AS in that order -- against the block the oracle restates (orc_hit.data), through the
    batch call and the per-read call.  The quality bytes after the first are uninitialised in the reference: masked."""
    from oracle import orc
    names, refs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    _, s1 = orc.read_fastq(os.path.join(golden_dir, "sim1_bcr.head3000.fq"), 120)
    comp = str.maketrans("ACGT", "TGCA")
    reads = list(s1[:100])
    reads.append(refs[0][5000:5100] + refs[1][9000:9070])                      # chimeric: soft / hard clips on both records
    reads.append((refs[1][20000:20090] + refs[2][700:790]).translate(comp)[::-1])          # the same on the reverse strand
    reads.append(refs[0][7000:7150].lower())                                   # lower case: aligns like upper case, packs as N
    reads.append(refs[0][8000:8060] + "NNNN" + refs[0][8064:8150])             # N bases
    reads.append(refs[2][100:250][:75] + "acgtn" + refs[2][100:250][80:])     # mixed case inside
    reads.append(refs[3][300:340])                                             # short
    reads.append("ACGTACGTAC")                                                 # no alignment: no record
    contig = list(refs[1][2000:72000])                                         # a 70 kb contig (the pipeline with 64-bit packed positions) with an insertion and
    contig[30000:30000] = list("ACGTTGCATT")                                   # a junction to another locus: a long CIGAR, a supplementary-like second record
    reads.append("".join(contig) + refs[0][90000:93000])
    names_r = ["q%d" % i for i in range(len(reads))]
    path = tmp_path / "reads.tsv"
    path.write_text("".join("%s\t%s\n" % (n, r) for n, r in zip(names_r, reads)))
    r = subprocess.run([exe, "blob", os.path.join(golden_dir, "tiny.fa"), str(path), str(hardclip), "0"], capture_output=True, text=True,
                       env=dict(os.environ, SEQLIB_AMD_SLAB_MIN_READS="1" if slabs else "1000000000"))
    assert r.returncode == 0, r.stderr
    got = {"B": {}, "S": {}}
    for line in r.stdout.strip().split("\n"):
        tag, i, k, tid, pos, flag, mapq, l_data, hexd = line.split("\t")
        got[tag].setdefault(int(i), []).append((int(tid), int(pos), int(flag), int(mapq), int(l_data), bytes.fromhex(hexd)))
    idx = orc.Index.load(os.path.join(golden_dir, "tiny.fa"))
    opt = orc.default_opt()
    n_clip = n_rec = 0

    def masked(d, l_qname, n_cig, l_qseq):
        q0 = l_qname + 4 * n_cig + (l_qseq + 1) // 2
        return d[:q0 + 1] + bytes(max(l_qseq - 1, 0)) + d[q0 + l_qseq:]

    for tag, base in (("B", 0), ("S", len(reads))):          # the per-read calls follow the batch: read i of them is call number len(reads) + i of the process
        for i, sq in enumerate(reads):
            exp = orc.align_sequence(opt, idx, sq, name=names_r[i], hardclip=bool(hardclip), ordinal=base + i)
            g = got[tag].get(i, [])
            assert len(g) == len(exp), (tag, i, len(g), len(exp))
            for a, e in zip(g, exp):
                assert a[:4] == (e["rid"], e["pos"], e["flag"], e["mapq"]) and a[4] == len(e["data"]), (tag, i)
                assert masked(a[5], e["l_qname"], len(e["cigar"]), e["l_qseq"]) == masked(e["data"], e["l_qname"], len(e["cigar"]), e["l_qseq"]), (tag, i)
                n_rec += 1
                n_clip += any((w & 0xf) in (4, 5) for w in e["cigar"])
    assert n_rec > 200 and n_clip >= 4
