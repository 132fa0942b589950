// Drives the C++ drop-in headers (include/SeqLib/*.h) the way a SeqLib user would.
//   seqlib_api_test cpu <index_prefix> <tmp_prefix>
//       host-only checks: Cigar/CigarField (mirrors /root/reference/tests/test_BamRecord.cpp:9-66),
//       setter exceptions and index accessors (mirrors /root/reference/seq_test/seq_test.cpp:798-828,
//       863-883), LoadIndex -> WriteIndex round trip.  No GPU call.
//   seqlib_api_test gpu <index_prefix> <fastq> <n_single> <n_batch>
//       aligns the first n_single reads one alignSequence call at a time and the next n_batch reads through
//       alignSequences, printing "read# rec# flag rid pos mapq CIGAR AS NM NA seq" per record.
//   seqlib_api_test fastq <file>
//       FastqReader: one "name<TAB>comment<TAB>seq<TAB>qual" line per record, reusing one UnalignedSequence as a
//       `while (r.GetNextSequence(s))` loop does.  No GPU call.
//   seqlib_api_test writer <tmp_prefix> <n_records>
//       BamWriter on hand-built records: <tmp_prefix>.sam and <tmp_prefix>.bam.  No GPU call.
//   seqlib_api_test threads <index_prefix> <fastq> <n> <T>
//       ONE BWAAligner shared by T host threads, each calling the const alignSequence on its own reads (what the reference allows,
//       SeqLib/BWAAligner.h:51-63): every read must come out as in a serial pass.  A read's lrand48 ordinal depends on the order
//       the threads arrive in, so reads whose records depend on the tie-breaking draw (found by two serial passes with different
//       ordinals) are left out of the comparison.
//   seqlib_api_test pipeline <index_prefix> <fastq> <n> <out_prefix>
//       FastqReader -> alignSequences -> BamWriter (SAM and BAM), the path a SeqLib user strings together.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <thread>
#include <atomic>
#include "SeqLib/BWAAligner.h"
#include "SeqLib/BamWriter.h"
#include "SeqLib/FastqReader.h"

using namespace SeqLib;

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)
#define CHECK_THROWS(expr, ex) do { bool ok_ = false; try { expr; } catch (const ex &) { ok_ = true; } catch (...) {} \
    if (!ok_) { std::fprintf(stderr, "expected " #ex " from %s (line %d)\n", #expr, __LINE__); return 1; } } while (0)

static int cpu_checks(const std::string &prefix, const std::string &tmp)
{
    // --- Cigar / CigarField
    CigarField cf('M', 35);
    std::ostringstream ss; ss << cf;
    CHECK(ss.str() == "35M");
    CHECK(cf.Type() == 'M' && cf.Length() == 35 && cf.ConsumesQuery() && cf.ConsumesReference());
    CHECK_THROWS(CigarField('Z', 3), std::invalid_argument);
    Cigar cig("5M2I3D4S");
    CHECK(cig.size() == 4);
    CHECK(cig[0].Type() == 'M' && cig[0].Length() == 5 && cig[1].Type() == 'I' && cig[2].Type() == 'D' && cig[3].Type() == 'S');
    CHECK(cig.NumQueryConsumed() == 11 && cig.NumReferenceConsumed() == 8);
    Cigar c2("5M2I3D4S"), c3("5M2I3D5S");
    CHECK(cig == c2 && cig != c3);
    std::ostringstream s2; s2 << cig;
    CHECK(s2.str() == "5M2I3D4S");
    // --- the batch path's record memory (BamRecord.h, detail::Slab) and the 4-bit packing of a record's sequence (BWAAligner.h, detail::pack_seq4), no GPU needed
    {
        // every length 0..200 on both strands, every kind of letter: the block form (SSE2) == the reference's switch, letter by letter (src/BWAAligner.cpp:208-231)
        unsigned long long lcg = 12345;
        auto rnd = [&]() { lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL; return (unsigned)(lcg >> 33); };
        for (int sl = 0; sl <= 200; ++sl)
            for (int rev = 0; rev < 2; ++rev)
                for (int rep = 0; rep < 4; ++rep) {
                    std::string q((size_t)sl, 'A');
                    for (auto &c : q) c = "ACGTNacgtnRY-"[rnd() % 100 < 90 ? rnd() & 3 : rnd() % 13];
                    std::vector<uint8_t> a((size_t)((sl + 1) >> 1) + 1, 0xAB), e((size_t)((sl + 1) >> 1) + 1, 0);
                    e.back() = 0xAB;
                    detail::pack_seq4(reinterpret_cast<const uint8_t *>(q.data()), sl, rev != 0, a.data());
                    detail::pack_seq4_scalar(reinterpret_cast<const uint8_t *>(q.data()), 0, sl, rev != 0, e.data());
                    CHECK(a == e);          // (and the byte behind the sequence is untouched)
                }
        // slabs: allocations of shells and blobs hold the block; it goes when the last one does, whatever the order; a record that outgrows its blob moves out of it
        std::vector<std::shared_ptr<bam1_t>> keep;
        {
            detail::SlabWriter w;
            w.hint_bytes = 1 << 16;
            for (int i = 0; i < 20000; ++i) {
                detail::Slab *slab = w.ensure(300 + 384);
                auto box = std::allocate_shared<Bam1Box>(detail::SlabAlloc<Bam1Box>(slab));
                box->b.data = static_cast<uint8_t *>(slab->take(300));
                slab->retain(); box->slab = slab;
                box->b.mempolicy = BAM_USER_OWNS_DATA;
                box->b.m_data = 300; box->b.l_data = 290;
                std::memset(box->b.data, i & 0xff, 290);
                if (i % 7 == 0) keep.emplace_back(box, &box->b);          // the others die at once, inside a slab that is still being filled
            }
        }
        for (size_t i = 0; i < keep.size(); ++i) {
            bam1_t *b = keep[i].get();
            CHECK(b->data[0] == (uint8_t)((i * 7) & 0xff) && b->data[289] == b->data[0]);
            if (i % 3 == 0) {          // 290 + 3 + 40 > 300: sam_realloc_bam_data's rule for caller-owned data
                const uint8_t *old = b->data;
                char z[40] = "0123456789012345678901234567890123456";
                CHECK(bam_aux_append(b, "ZZ", 'Z', 40, reinterpret_cast<const uint8_t *>(z)) == 0);
                CHECK(b->data != old && !(b->mempolicy & BAM_USER_OWNS_DATA) && b->data[0] == (uint8_t)((i * 7) & 0xff) && b->l_data == 333);
            }
        }
        for (size_t i = 0; i < keep.size(); i += 2) keep[i].reset();
        keep.clear();
        detail::Slab::trim_pool();
    }
    // --- BamRecord tags / accessors on a hand-built record
    {
        BamRecord r;
        bam1_t *b = r.raw();
        const char *nm = "q1";
        b->core.l_qname = 3; b->core.n_cigar = 1; b->core.l_qseq = 4;
        b->l_data = 3 + 4 + 2 + 4;
        b->data = (uint8_t *)std::calloc((size_t)b->l_data, 1);
        std::memcpy(b->data, nm, 3);
        uint32_t w = bam_cigar_gen(4, BAM_CMATCH); std::memcpy(b->data + 3, &w, 4);
        b->data[7] = 0x12; b->data[8] = 0x48;     // ACGT
        r.AddIntTag("NM", 7); r.AddZTag("XA", "hello"); r.AddIntTag("AS", -3); r.AddZTag("XA", "again");
        int32_t v = 0; std::string z;
        CHECK(r.GetIntTag("NM", v) && v == 7);
        CHECK(r.GetIntTag("AS", v) && v == -3);
        CHECK(r.GetZTag("XA", z) && z == "again");
        CHECK(!r.GetIntTag("ZZ", v));
        CHECK(r.Qname() == "q1" && r.Sequence() == "ACGT" && r.CigarString() == "4M" && r.Length() == 4);
        CHECK(r.GetCigar() == Cigar("4M"));
    }
    // --- index accessors and errors
    auto idx = std::make_shared<BWAIndex>();
    CHECK(idx->IsEmpty() && idx->NumSequences() == 0 && idx->printSamHeader().empty());
    CHECK_THROWS(idx->ChrIDToName(1), std::runtime_error);
    CHECK_THROWS(idx->WriteIndex(tmp), std::runtime_error);
    CHECK_THROWS(idx->LoadIndex("/nonexistent/prefix"), std::runtime_error);
    idx->LoadIndex(prefix);
    CHECK(!idx->IsEmpty() && idx->NumSequences() == 4);
    CHECK(idx->ChrIDToName(0) == "bcr" && idx->ChrIDToName(3) == "myc");
    CHECK_THROWS(idx->ChrIDToName(4), std::out_of_range);
    CHECK_THROWS(idx->ChrIDToName(-1), std::out_of_range);
    BamHeader hh = idx->HeaderFromIndex();
    CHECK(hh.NumSequences() == 4 && hh.IDtoName(1) == "abl" && hh.GetSequenceLength(1) == 178633);
    // --- BamHeader's own surface (/root/reference/SeqLib/BamHeader.h:36-107, src/BamHeader.cpp:12-140)
    {
        CHECK(hh.IsOpen() && !hh.isEmpty() && hh.Name2ID("abl") == 1 && hh.Name2ID("nope") == -1);
        CHECK(hh.GetSequenceLength("abl") == 178633 && hh.GetSequenceLength("nope") == -1 && hh.GetSequenceLength(4) == -1);
        const HeaderSequenceVector hsv = hh.GetHeaderSequenceVector();
        CHECK(hsv.size() == 4 && hsv[0].Name == "bcr" && hsv[3].Name == "myc" && hsv[1].Length == 178633u);
        BamHeader h2(hsv);                                   // names and lengths -> "@HD VN:1.4" + one @SQ line each
        CHECK(h2.IsOpen() && h2.NumSequences() == 4 && h2.IDtoName(3) == "myc" && h2.GetSequenceLength("bcr") == hh.GetSequenceLength(0));
        CHECK(h2.AsString().rfind("@HD\tVN:1.4\n@SQ\tSN:bcr\tLN:", 0) == 0);
        BamHeader h3(h2.AsString());                         // ... and that text parses back to the same dictionary
        CHECK(h3.NumSequences() == 4 && h3.Name2ID("myc") == 3 && h3.GetSequenceLength(2) == h2.GetSequenceLength(2));
        BamHeader none;                                      // uninitialised: no sequences, IDtoName says so
        CHECK(!none.IsOpen() && none.isEmpty() && none.NumSequences() == 0 && none.GetSequenceLength(0) == -1 && none.Name2ID("bcr") == -1);
        CHECK_THROWS(none.IDtoName(0), std::out_of_range);
        CHECK_THROWS(hh.IDtoName(-1), std::invalid_argument);
        CHECK_THROWS(hh.IDtoName(4), std::out_of_range);
        BamHeader only_hd("@HD\tVN:1.6\n");                  // a header without @SQ lines is open and holds no sequences
        CHECK(only_hd.IsOpen() && !only_hd.isEmpty() && only_hd.NumSequences() == 0);
        BamHeader twice("@SQ\tSN:x\tLN:5\n@SQ\tSN:x\tLN:7\n");  // of two sequences of one name the first keeps it
        CHECK(twice.NumSequences() == 2 && twice.Name2ID("x") == 0 && twice.GetSequenceLength("x") == 5);
    }
    std::ostringstream s3; s3 << *idx;
    CHECK(s3.str() == "[BWAIndex] #seqs=4 pac_len=354751 holes=0");
    idx->WriteIndex(tmp);
    for (const char *ext : {".bwt", ".sa", ".pac", ".ann", ".amb"}) {
        std::ifstream a(prefix + ext, std::ios::binary), b(tmp + ext, std::ios::binary);
        std::stringstream sa, sb; sa << a.rdbuf(); sb << b.rdbuf();
        CHECK(sa.str() == sb.str() && !sa.str().empty());
    }
    UnalignedSequenceVector bad1 = {{"ref1", "ACGT"}, {"ref4", ""}}, bad2 = {{"", "ACGT"}};
    CHECK_THROWS(idx->ConstructIndex(bad1), std::invalid_argument);
    CHECK_THROWS(idx->ConstructIndex(bad2), std::invalid_argument);
    // --- aligner setters
    BWAAligner bwa(idx);
    bwa.SetGapOpen(32); bwa.SetGapExtension(1); bwa.SetMismatchPenalty(18); bwa.SetAScore(2); bwa.SetZDropoff(100);
    bwa.Set3primeClippingPenalty(5); bwa.Set5primeClippingPenalty(5); bwa.SetBandwidth(1000); bwa.SetReseedTrigger(1.5);
    CHECK_THROWS(bwa.SetGapOpen(-1), std::invalid_argument);
    CHECK_THROWS(bwa.SetGapExtension(-1), std::invalid_argument);
    CHECK_THROWS(bwa.SetMismatchPenalty(-18), std::invalid_argument);
    CHECK_THROWS(bwa.SetAScore(-2), std::invalid_argument);
    CHECK_THROWS(bwa.SetZDropoff(-100), std::invalid_argument);
    CHECK_THROWS(bwa.Set3primeClippingPenalty(-5), std::invalid_argument);
    CHECK_THROWS(bwa.Set5primeClippingPenalty(-5), std::invalid_argument);
    CHECK_THROWS(bwa.SetBandwidth(-1000), std::invalid_argument);
    CHECK_THROWS(bwa.SetReseedTrigger(-1.5), std::invalid_argument);
    // empty index => silent no-op
    BWAAligner none(std::make_shared<BWAIndex>());
    BamRecordPtrVector out;
    none.alignSequence("ACGTACGTACGTACGTACGTACGT", "x", out, false, 0.9, 10);
    CHECK(out.empty());
    std::puts("cpu checks OK");
    return 0;
}

// a bam1_t built field by field the way src/BWAAligner.cpp:151-248 lays it out (qname, cigar, 4-bit seq, qual, aux)
static BamRecordPtr make_record(const std::string &name, int flag, int tid, int pos, int mapq, const std::string &cigar, int mtid, int mpos,
                                int isize, const std::string &seq, const std::string &qual)
{
    auto r = std::make_shared<BamRecord>();
    bam1_t *b = r->raw();
    Cigar cg(cigar);
    b->core.tid = tid; b->core.pos = pos; b->core.qual = (uint8_t)mapq; b->core.flag = (uint16_t)flag; b->core.n_cigar = (uint32_t)cg.size();
    b->core.mtid = mtid; b->core.mpos = mpos; b->core.isize = isize;
    b->core.l_qname = (uint16_t)(name.size() + 1); b->core.l_qseq = (int32_t)seq.size();
    b->l_data = (int)(b->core.l_qname + 4 * cg.size() + (seq.size() + 1) / 2 + seq.size());
    b->data = (uint8_t *)std::calloc((size_t)b->l_data, 1);
    b->m_data = (uint32_t)b->l_data;
    std::memcpy(b->data, name.c_str(), name.size() + 1);
    for (size_t k = 0; k < cg.size(); ++k) { uint32_t w = cg[k].raw(); std::memcpy(b->data + b->core.l_qname + 4 * k, &w, 4); }
    uint8_t *sq = bam_get_seq(b);
    for (size_t k = 0; k < seq.size(); ++k) {
        const char *nt = "=ACMGRSVTWYHKDBN";
        const char *f = std::strchr(nt, seq[k]);
        const uint8_t v = f ? (uint8_t)(f - nt) : 15;
        sq[k >> 1] |= (k & 1) ? v : (uint8_t)(v << 4);
    }
    uint8_t *q = bam_get_qual(b);
    if (qual.empty()) { if (!seq.empty()) std::memset(q, 0xff, seq.size()); }
    else for (size_t k = 0; k < seq.size(); ++k) q[k] = (uint8_t)(qual[k] - 33);
    return r;
}

static int fastq_dump(const std::string &file)
{
    FastqReader r;
    if (!r.Open(file)) return 3;
    UnalignedSequence s;
    while (r.GetNextSequence(s)) std::printf("%s\t%s\t%s\t%s\n", s.Name.c_str(), s.Com.c_str(), s.Seq.c_str(), s.Qual.c_str());
    return 0;
}

static int writer_checks(const std::string &tmp, long n)
{
    CHECK_THROWS(BamWriter(99), std::invalid_argument);
    BamHeader hdr("@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chrA\tLN:1000\n@SQ\tSN:chrB\tLN:50000000\n@PG\tID:test\n");
    for (int fmt : {SAM, BAM}) {
        BamWriter w(fmt);
        CHECK(!w.IsOpen() && !w.WriteHeader() && !w.Close());
        CHECK(!w.WriteRecord(*make_record("x", 4, -1, -1, 0, "", -1, -1, 0, "", "")));
        CHECK(w.Open(tmp + (fmt == SAM ? ".sam" : ".bam")));
        CHECK(!w.Open(tmp + ".again"));               // no reopen
        CHECK(!w.WriteHeader());                      // no header yet
        w.SetHeader(hdr);
        CHECK(w.WriteHeader());
        CHECK(!w.BuildIndex());                       // still open
        for (long i = 0; i < n; ++i) {
            std::string seq, qual;
            for (int k = 0; k < 40 + (int)(i % 7); ++k) { seq.push_back("ACGTN"[(i * 7 + k * 3 + k / 5) % 5]); qual.push_back((char)(33 + (i + k) % 41)); }
            const int len = (int)seq.size();
            BamRecordPtr r;
            switch (i % 4) {
            case 0: r = make_record("r" + std::to_string(i), 0, 0, (int)(i % 900), 60, std::to_string(len) + "M", -1, -1, 0, seq, ""); break;
            case 1: r = make_record("r" + std::to_string(i), 16 | 1, 1, (int)(i * 16411 % 49000000), 13,
                                    "5S" + std::to_string(len - 10) + "M2D3I2H", 1, (int)(i * 16411 % 49000000) + 300, -350, seq, qual); break;
            case 2: r = make_record("r" + std::to_string(i), 256, 1, 1 << 20, 0, "10M100N" + std::to_string(len - 10) + "M", 0, 7, 0, seq, qual); break;
            default: r = make_record("r" + std::to_string(i), 4, -1, -1, 0, "", -1, -1, 0, seq, qual); break;
            }
            r->AddIntTag("NA", (int32_t)(i % 3)); r->AddIntTag("NM", (int32_t)(i % 11)); r->AddIntTag("AS", (int32_t)(len - i % 5));
            if (i % 5 == 0) r->AddZTag("XA", "chrA,+" + std::to_string(i) + ",40M,0;");
            CHECK(w.WriteRecord(*r));
        }
        CHECK(!w.WriteRecord(*make_record("bad", 0, 7, 1, 0, "1M", -1, -1, 0, "A", "")) || fmt == BAM);   // tid outside the header: SAM cannot name it
        CHECK(w.Close());
        CHECK(!w.Close() && !w.IsOpen());
    }
    BamWriter c(CRAM);
    CHECK(!c.Open(tmp + ".cram"));
    std::puts("writer checks OK");
    return 0;
}

static int pipeline(const std::string &prefix, const std::string &fastq, long n, const std::string &outp)
{
    auto idx = std::make_shared<BWAIndex>();
    idx->LoadIndex(prefix);
    BWAAligner bwa(idx);
    FastqReader fr(fastq);
    UnalignedSequenceVector reads;
    UnalignedSequence s;
    while ((long)reads.size() < n && fr.GetNextSequence(s)) reads.push_back(s);
    std::vector<BamRecordPtrVector> outs;
    bwa.alignSequences(reads, outs, false, 0.9, 10);
    BamWriter ws(SAM), wb(BAM);
    CHECK(ws.Open(outp + ".sam") && wb.Open(outp + ".bam"));
    ws.SetHeader(idx->HeaderFromIndex()); wb.SetHeader(idx->HeaderFromIndex());
    CHECK(ws.WriteHeader() && wb.WriteHeader());
    for (auto &v : outs) for (auto &r : v) CHECK(ws.WriteRecord(*r) && wb.WriteRecord(*r));
    CHECK(ws.Close() && wb.Close());
    return 0;
}

static void print_rec(long read_no, size_t j, const BamRecord &r)
{
    int32_t as = 0, nm = 0, na = 0;
    r.GetIntTag("AS", as); r.GetIntTag("NM", nm); r.GetIntTag("NA", na);
    std::printf("%ld\t%zu\t%u\t%d\t%d\t%d\t%s\t%d\t%d\t%d\t%s\n", read_no, j, r.AlignmentFlag(), r.ChrID(), r.Position(), r.MapQuality(),
                r.CigarString().c_str(), as, nm, na, r.Sequence().c_str());
}

static std::string rec_key(const BamRecordPtrVector &v)
{
    std::ostringstream o;
    for (auto &r : v) {
        int32_t as = 0, nm = 0, na = 0;
        r->GetIntTag("AS", as); r->GetIntTag("NM", nm); r->GetIntTag("NA", na);
        o << r->AlignmentFlag() << ':' << r->ChrID() << ':' << r->Position() << ':' << r->MapQuality() << ':' << r->CigarString() << ':' << as << ':' << nm << ':' << na
          << ':' << r->Sequence() << ';';
    }
    return o.str();
}

static int thread_checks(const std::string &prefix, const std::string &fastq, long n, int T)
{
    auto idx = std::make_shared<BWAIndex>();
    idx->LoadIndex(prefix);
    const BWAAligner bwa(idx);                      // const: only the reference's const entry points are used below
    std::ifstream fq(fastq);
    std::string h, s, p, q;
    UnalignedSequenceVector reads;
    while (std::getline(fq, h) && std::getline(fq, s) && std::getline(fq, p) && std::getline(fq, q) && (long)reads.size() < n)
        reads.emplace_back(h.substr(1), s, q);
    n = (long)reads.size();
    // concurrent calls are combined into shared GPU round trips (BWAAligner.h, "flat combining"): once with one argument set for every call, once with the
    // glue arguments differing from read to read (a round only takes the calls whose arguments agree; the others are served by the next one)
    for (int mixed = 0; mixed < 2; ++mixed) {
        auto hc = [&](long i) { return mixed && i % 3 == 0; };
        auto ms = [&](long i) { return mixed && i % 5 == 0 ? 1 : 10; };
        std::vector<std::string> pass1((size_t)n), pass2((size_t)n), par((size_t)n);
        for (int rep = 0; rep < 2; ++rep)
            for (long i = 0; i < n; ++i) {
                BamRecordPtrVector out;
                bwa.alignSequence(reads[(size_t)i], out, hc(i), 0.9, ms(i));
                (rep ? pass2 : pass1)[(size_t)i] = rec_key(out);
            }
        std::atomic<long> next(0);
        std::atomic<int> failures(0);
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&]() {
                try {
                    for (long i; (i = next.fetch_add(1)) < n;) {
                        BamRecordPtrVector out;
                        bwa.alignSequence(reads[(size_t)i], out, hc(i), 0.9, ms(i));
                        par[(size_t)i] = rec_key(out);
                        for (auto &r : out) if (r->Qname() != reads[(size_t)i].Name) ++failures;          // a caller gets ITS read's records
                    }
                } catch (const std::exception &e) { std::fprintf(stderr, "thread: %s\n", e.what()); ++failures; }
            });
        for (auto &x : th) x.join();
        CHECK(failures == 0);
        long stable = 0, bad = 0;
        for (long i = 0; i < n; ++i) {
            if (pass1[(size_t)i] != pass2[(size_t)i]) continue;        // depends on the tie-breaking draw
            ++stable;
            if (par[(size_t)i] != pass1[(size_t)i]) { if (!bad) std::fprintf(stderr, "read %ld differs:\n %s\n %s\n", i, par[(size_t)i].c_str(), pass1[(size_t)i].c_str()); ++bad; }
        }
        std::printf("threads=%d reads=%ld mixed=%d stable=%ld mismatches=%ld\n", T, n, mixed, stable, bad);
        CHECK(bad == 0 && stable > n * 9 / 10);
    }
    {   // an error in a shared round reaches every caller of that round as the exception a lone call would throw, and the aligner stays usable
        std::atomic<int> threw(0), fine(0);
        const std::string too_long((size_t)SLX_MAX_READ_LEN + 8, 'A');
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t]() {
                for (int k = 0; k < 4; ++k) {
                    BamRecordPtrVector out;
                    try { bwa.alignSequence(t == 0 ? too_long : reads[(size_t)((t * 4 + k) % n)].Seq, "x", out, false, 0.9, 10); ++fine; }
                    catch (const std::exception &) { ++threw; }
                }
            });
        for (auto &x : th) x.join();
        CHECK(threw >= 4);                       // thread 0's four calls at least (and whoever shared a round with them)
        BamRecordPtrVector out;
        bwa.alignSequence(reads[0].Seq, "after", out, false, 0.9, 10);
        CHECK(!out.empty());
        std::printf("error rounds: %d calls threw, %d were fine\n", threw.load(), fine.load());
    }
    return 0;
}

// every byte a record carries: core fields + the data blob
static std::string rec_bytes(const BamRecordPtrVector &v)
{
    std::string o;
    for (auto &r : v) {
        const bam1_t *b = r->raw();
        o.append(reinterpret_cast<const char *>(&b->core.pos), sizeof b->core.pos);
        o.append(reinterpret_cast<const char *>(&b->core.tid), sizeof b->core.tid);
        const uint32_t misc[6] = {b->core.qual, b->core.flag, b->core.l_qname, b->core.n_cigar, (uint32_t)b->core.l_qseq, (uint32_t)b->l_data};
        o.append(reinterpret_cast<const char *>(misc), sizeof misc);
        o.append(reinterpret_cast<const char *>(b->data), (size_t)b->l_data);
        o.push_back('|');
    }
    return o;
}

// multidev <prefix> <fastq> <n> <copies>: the batch through (a) one device in one chunk, (b) every visible device listed `copies`
// times (a group handle: the C-ABI shards the batch) with the batch cut into small chunks (SEQLIB_AMD_CHUNK) so that the
// pack / align / build pipeline of alignSequences runs several rounds.  Every record must be byte-identical.
static int multidev_checks(const std::string &prefix, const std::string &fastq, long n, int copies)
{
    auto idx = std::make_shared<BWAIndex>();
    idx->LoadIndex(prefix);
    std::ifstream fq(fastq);
    std::string h, s, p, q;
    UnalignedSequenceVector reads;
    while (std::getline(fq, h) && std::getline(fq, s) && std::getline(fq, p) && std::getline(fq, q) && (long)reads.size() < n)
        reads.emplace_back(h.substr(1), s, q);
    reads.emplace_back("empty", "");                  // ragged input: an empty read, a short one, a low-complexity one
    reads.emplace_back("short", "ACGTACGTAC");
    reads.emplace_back("lowcx", std::string(150, 'A'));
    const uint64_t st = slx_lrand48_peek_libc();
    auto rewind = [&]() { unsigned short x[3] = {(unsigned short)(st & 0xffff), (unsigned short)((st >> 16) & 0xffff), (unsigned short)((st >> 32) & 0xffff)}; seed48(x); };
    std::vector<BamRecordPtrVector> one, many;
    {
        BWAAligner bwa(idx);
        bwa.alignSequences(reads, one, false, 0.9, 10);
    }
    rewind();
    const int ndev = slx_device_count();
    CHECK(ndev >= 1);
    std::vector<int> devs;
    for (int c = 0; c < copies; ++c) for (int d = 0; d < ndev; ++d) devs.push_back(d);
    setenv("SEQLIB_AMD_CHUNK", "257", 1);
    setenv("SEQLIB_AMD_THREADS", "5", 1);
    {
        BWAAligner bwa(idx);
        bwa.UseDevices(devs);
        bwa.alignSequences(reads, many, false, 0.9, 10);
    }
    CHECK(one.size() == reads.size() && many.size() == reads.size());
    long bad = 0, recs = 0;
    for (size_t i = 0; i < reads.size(); ++i) {
        recs += (long)one[i].size();
        if (rec_bytes(one[i]) != rec_bytes(many[i])) { if (!bad) std::fprintf(stderr, "read %zu differs\n", i); ++bad; }
    }
    std::printf("devices=%zu (visible %d) reads=%zu records=%ld mismatches=%ld\n", devs.size(), ndev, reads.size(), recs, bad);
    CHECK(bad == 0 && recs > n * 9 / 10);
    return 0;
}

// bwamem <prefix> <fastq> <n>: UseBwaMemRecords -- "read# rec# flag rid pos mapq CIGAR AS NM NA XS XA SA" per record ('*' = tag absent)
static int bwamem_dump(const std::string &prefix, const std::string &fastq, long n)
{
    auto idx = std::make_shared<BWAIndex>();
    idx->LoadIndex(prefix);
    BWAAligner bwa(idx);
    bwa.UseBwaMemRecords();
    CHECK_THROWS(bwa.SetOutputScoreThreshold(-1), std::invalid_argument);
    std::ifstream fq(fastq);
    std::string h, s, p, q;
    UnalignedSequenceVector reads;
    while (std::getline(fq, h) && std::getline(fq, s) && std::getline(fq, p) && std::getline(fq, q) && (long)reads.size() < n)
        reads.emplace_back(h.substr(1), s, q);
    reads.emplace_back("nohit", "ACGTACGTACGTTGCATGCATGCAAACCGGTT");
    std::vector<BamRecordPtrVector> outs;
    bwa.alignSequences(reads, outs, false, 0.9, 10);
    for (size_t i = 0; i < outs.size(); ++i)
        for (size_t j = 0; j < outs[i].size(); ++j) {
            const BamRecord &r = *outs[i][j];
            int32_t as = 0, nm = -1, na = 0, xs = -1;
            std::string xa = "*", sa = "*", md = "*";
            r.GetIntTag("AS", as); r.GetIntTag("NM", nm); r.GetIntTag("NA", na);
            if (!r.GetIntTag("XS", xs)) xs = -1;
            r.GetZTag("XA", xa); r.GetZTag("SA", sa); r.GetZTag("MD", md);
            std::printf("%zu\t%zu\t%u\t%d\t%d\t%d\t%s\t%d\t%d\t%d\t%d\t%s\t%s\t%s\n", i, j, r.AlignmentFlag(), r.ChrID(), r.Position(), r.MapQuality(),
                        r.CigarString().empty() ? "*" : r.CigarString().c_str(), as, nm, na, xs, xa.c_str(), sa.c_str(), md.c_str());
        }
    return 0;
}

// blob <index prefix> <reads file: one read per line, "name<TAB>sequence"> <hardclip> <bwa-mem records>: every record's bam1_t data block as hex
// (qname, CIGAR, 4-bit sequence, qualities, tags: what /root/reference/src/BWAAligner.cpp:151-248 builds), through the per-read call AND the batch call
static int blob_dump(const char *prefix, const char *reads_path, int hardclip, int sam)
{
    BWAIndexPtr idx = std::make_shared<BWAIndex>();
    idx->LoadIndex(prefix);
    BWAAligner al(idx);
    if (sam) al.UseBwaMemRecords(true);
    UnalignedSequenceVector reads;
    std::ifstream in(reads_path);
    std::string line;
    while (std::getline(in, line)) {
        const size_t t = line.find('\t');
        if (t == std::string::npos) continue;
        reads.push_back(UnalignedSequence(line.substr(0, t), line.substr(t + 1)));
    }
    auto dump = [](const char *tag, size_t i, const BamRecordPtrVector &v) {
        for (size_t k = 0; k < v.size(); ++k) {
            const bam1_t *b = v[k]->raw();
            std::printf("%s\t%zu\t%zu\t%d\t%d\t%d\t%d\t%d\t", tag, i, k, b->core.tid, (int)b->core.pos, (int)b->core.flag, (int)b->core.qual, b->l_data);
            for (int x = 0; x < b->l_data; ++x) std::printf("%02x", b->data[x]);
            std::printf("\n");
        }
    };
    std::vector<BamRecordPtrVector> batch;
    al.alignSequences(reads, batch, hardclip != 0, 0.9, 10);
    for (size_t i = 0; i < batch.size(); ++i) dump("B", i, batch[i]);
    for (size_t i = 0; i < reads.size(); ++i) {
        BamRecordPtrVector one;
        al.alignSequence(reads[i], one, hardclip != 0, 0.9, 10);
        dump("S", i, one);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 6 && std::string(argv[1]) == "blob") return blob_dump(argv[2], argv[3], std::atoi(argv[4]), std::atoi(argv[5]));
    if (argc >= 5 && std::string(argv[1]) == "bwamem") return bwamem_dump(argv[2], argv[3], std::atol(argv[4]));
    if (argc >= 6 && std::string(argv[1]) == "multidev") return multidev_checks(argv[2], argv[3], std::atol(argv[4]), std::atoi(argv[5]));
    if (argc >= 4 && std::string(argv[1]) == "cpu") return cpu_checks(argv[2], argv[3]);
    if (argc >= 3 && std::string(argv[1]) == "fastq") return fastq_dump(argv[2]);
    if (argc >= 4 && std::string(argv[1]) == "writer") return writer_checks(argv[2], std::atol(argv[3]));
    if (argc >= 6 && std::string(argv[1]) == "threads") return thread_checks(argv[2], argv[3], std::atol(argv[4]), std::atoi(argv[5]));
    if (argc >= 6 && std::string(argv[1]) == "pipeline") return pipeline(argv[2], argv[3], std::atol(argv[4]), argv[5]);
    if (argc >= 6 && std::string(argv[1]) == "gpu") {
        auto idx = std::make_shared<BWAIndex>();
        idx->LoadIndex(argv[2]);
        BWAAligner bwa(idx);
        const long n1 = std::atol(argv[4]), n2 = std::atol(argv[5]);
        std::ifstream fq(argv[3]);
        std::string h, s, p, q;
        UnalignedSequenceVector reads;
        while (std::getline(fq, h) && std::getline(fq, s) && std::getline(fq, p) && std::getline(fq, q) && (long)reads.size() < n1 + n2)
            reads.emplace_back(h.substr(1), s, q);
        long i = 0;
        for (; i < n1 && i < (long)reads.size(); ++i) {
            BamRecordPtrVector out;
            bwa.alignSequence(reads[(size_t)i], out, false, 0.9, 10);
            for (size_t j = 0; j < out.size(); ++j) print_rec(i, j, *out[j]);
        }
        UnalignedSequenceVector rest(reads.begin() + i, reads.end());
        std::vector<BamRecordPtrVector> outs;
        bwa.alignSequences(rest, outs, false, 0.9, 10);
        for (size_t k = 0; k < outs.size(); ++k)
            for (size_t j = 0; j < outs[k].size(); ++j) print_rec(i + (long)k, j, *outs[k][j]);
        // north-star spelling with a BamRecordVector
        BamRecordVector brv;
        bwa.AlignSequence(reads[0].Seq, "name", brv, false, 0.9, 1);
        std::fprintf(stderr, "AlignSequence/BamRecordVector: %zu record(s), qname %s\n", brv.size(), brv.empty() ? "-" : brv[0].Qname().c_str());
        return 0;
    }
    std::fprintf(stderr, "usage: %s cpu <prefix> <tmp> | gpu <prefix> <fastq> <n_single> <n_batch> | fastq <file> | writer <tmp> <n> | "
                         "pipeline <prefix> <fastq> <n> <out>\n", argv[0]);
    return 2;
}
