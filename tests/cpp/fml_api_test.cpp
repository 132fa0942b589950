// fml_api_test -- drives the C++ mirrors of SeqLib::FermiAssembler and SeqLib::BFC the way the reference's own callers do
// (/root/reference/seq_test/seq_test.cpp:104-160 "correct_and_assemble", :374-392 "fermi_add_reads", :468-503 "fermi_assemble";
// /root/reference/src/seqtools/seqtools.cpp:106-212) and prints what tests/test_cpp_fml.py compares with the CPU checker.
//   fml_api_test cpu
//   fml_api_test gpu <fastq> <n_reads>
//   fml_api_test pipeline <index prefix> <fastq> <n_reads>      reads -> FermiAssembler -> contigs -> BWAAligner::alignSequence (seqtools' fml mode)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include "SeqLib/BFC.h"
#include "SeqLib/BWAAligner.h"
#include "SeqLib/FastqReader.h"
#include "SeqLib/FermiAssembler.h"

using namespace SeqLib;

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed at %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static int cpu()
{
    FermiAssembler f;
    CHECK(f.GetMinOverlap() == 33 && f.NumSequences() == 0);
    f.SetMinOverlap(40); CHECK(f.GetMinOverlap() == 40);
    f.AddRead(UnalignedSequence("", "ACGT", "IIII"));          // no name: ignored (src/FermiAssembler.cpp:54-55)
    f.AddRead(UnalignedSequence("a", "", ""));
    CHECK(f.NumSequences() == 0);
    f.AddRead(UnalignedSequence("a", "ACGT", "IIII"));
    UnalignedSequenceVector v; v.push_back(UnalignedSequence("b", "GGCC", "IIII")); v.push_back(UnalignedSequence("c", "", ""));
    f.AddReads(v);          // AddReads takes every record (:64-81)
    CHECK(f.NumSequences() == 3);
    UnalignedSequenceVector g = f.GetSequences();
    CHECK(g.size() == 3 && g[0].Name == "a" && g[1].Seq == "GGCC");
    f.ClearReads(); CHECK(f.NumSequences() == 0);
    CHECK(f.GetContigs().empty());
    fml_opt_t o; fml_opt_init(&o);
    CHECK(o.min_cnt == 4 && o.max_cnt == 8 && o.mag_opt.flag == (MAG_F_NO_SIMPL | MAG_F_POPOPEN));
    o.min_asm_ovlp = 51;
    FermiAssembler f2(o); CHECK(f2.GetMinOverlap() == 51);
    BFC b;
    CHECK(!b.AddSequence("", "", "x") && !b.AddSequence("ACGT", "II", "x") && b.AddSequence("ACGT", "", "x") && b.AddSequence("ACGT", "IIII", "y"));
    CHECK(b.NumSequences() == 2 && b.GetKMer() == 0 && b.GetKCov() == 0);
    std::string s, q;
    CHECK(b.GetSequence(s, q) && s == "ACGT" && q == "x" && b.GetSequence(s, q) && !b.GetSequence(s, q));
    b.ResetGetSequence(); CHECK(b.GetSequence(s, q));
    b.ClearReads(); CHECK(b.NumSequences() == 0);
    {   // qualities are kept per read: a read without gets a stretch of 'I' in the flat buffer, and no buffer at all when no read has qualities
        std::vector<std::string> sq = {"ACGT", "GG", "TTT"}, ql = {"#I#I", "", "II"};
        detail::FlatReads fr = detail::flat_reads(sq, ql, nullptr);
        CHECK(fr.has_qual && fr.bases == "ACGTGGTTT" && fr.quals == "#I#IIIIII" && fr.offs.size() == 4 && fr.offs[3] == 9);
        std::vector<char> hq = {0, 0, 0};
        CHECK(!detail::flat_reads(sq, ql, &hq).has_qual && detail::flat_reads(sq, ql, &hq).quals.empty());
    }
    bool threw = false;
    try { f.AddRead(UnalignedSequence("a", "ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT", "")); f.CorrectReads(); } catch (const std::runtime_error &) { threw = true; }
    std::printf("no-GPU call %s\n", threw ? "throws" : "runs");
    std::printf("cpu checks OK\n");
    return 0;
}

static int gpu(const char *fq, long n)
{
    FastqReader r(fq);
    UnalignedSequenceVector reads;
    UnalignedSequence u;
    while ((long)reads.size() < n && r.GetNextSequence(u)) reads.push_back(u);
    // 1. seqtools' fml pipeline: AddRead each, CorrectReads, PerformAssembly, GetContigs
    FermiAssembler f;
    for (const UnalignedSequence &x : reads) f.AddRead(x);
    f.CorrectReads();
    UnalignedSequenceVector cor = f.GetSequences();
    for (const UnalignedSequence &x : cor) std::cout << "COR\t" << x.Name << "\t" << x.Seq << "\n";
    f.PerformAssembly();
    for (const std::string &c : f.GetContigs()) std::cout << "CTG\t" << c << "\n";
    std::ostringstream gfa; f.WriteGFA(gfa);
    std::cout << "GFA\t" << gfa.str().size() << "\n";
    // 2. seq_test's correct_and_assemble: BFC Train + ErrorCorrect, then DirectAssemble(kcov)
    BFC b;
    for (const UnalignedSequence &x : reads) b.AddSequence(x.Seq, x.Qual, x.Name);
    b.Train();
    b.ErrorCorrect();
    std::cout << "BFC\t" << b.GetKMer() << "\t" << b.GetKCov() << "\n";
    UnalignedSequenceVector v;
    std::string s, nm;
    while (b.GetSequence(s, nm)) v.push_back(UnalignedSequence(nm, s));
    FermiAssembler d;
    d.AddReads(v);
    d.DirectAssemble(b.GetKCov());
    for (const std::string &c : d.GetContigs()) std::cout << "DIR\t" << c << "\n";
    // 3. the batch entry: two windows in one call
    std::vector<int64_t> win_off = {0, (int64_t)reads.size() / 2, (int64_t)reads.size()};
    std::vector<std::vector<std::string> > ctg;
    FermiAssembler::AssembleWindows(reads, win_off, ctg);
    for (size_t w = 0; w < ctg.size(); ++w) for (const std::string &c : ctg[w]) std::cout << "WIN" << w << "\t" << c << "\n";
    // 4. filter
    FermiAssembler g2;
    g2.AddReads(reads);
    g2.CorrectAndFilterReads();
    size_t kept = 0;
    for (const UnalignedSequence &x : g2.GetSequences()) { kept += !x.Seq.empty(); std::cout << "FLT\t" << x.Seq << "\n"; }
    std::cerr << "kept " << kept << " of " << reads.size() << "\n";
    // 5. two BFC objects own a table each (src/BFC.cpp:208-286: the object's own bfc_ch_t): A trains on the first half, B on the second with another k,
    //    a FermiAssembler runs in between, then A corrects against ITS table
    {
        BFC A, B;
        const size_t half = reads.size() / 2;
        for (size_t i = 0; i < half; ++i) A.AddSequence(reads[i].Seq, reads[i].Qual, reads[i].Name);
        for (size_t i = half; i < reads.size(); ++i) B.AddSequence(reads[i].Seq, reads[i].Qual, reads[i].Name);
        A.SetKmer(17); B.SetKmer(21);
        A.Train(); B.Train();
        FermiAssembler between;
        for (size_t i = 0; i < 200 && i < reads.size(); ++i) between.AddRead(reads[i]);
        between.CorrectReads();
        A.ErrorCorrect();
        B.ErrorCorrect();
        std::cout << "BF2\t" << A.GetKMer() << "\t" << A.GetKCov() << "\t" << B.GetKMer() << "\t" << B.GetKCov() << "\n";
        std::string sa, na;
        while (A.GetSequence(sa, na)) std::cout << "BFA\t" << sa << "\n";
        while (B.GetSequence(sa, na)) std::cout << "BFB\t" << sa << "\n";
    }
    // 6. a mixed read set: every third read has no quality string (AddRead keeps qual = NULL for it, src/FermiAssembler.cpp:52-62)
    {
        FermiAssembler m;
        for (size_t i = 0; i < reads.size(); ++i) m.AddRead(i % 3 == 1 ? UnalignedSequence(reads[i].Name, reads[i].Seq, "") : reads[i]);
        m.CorrectReads();
        for (const UnalignedSequence &x : m.GetSequences()) std::cout << "MIX\t" << x.Seq << "\n";
    }
    return 0;
}

// src/seqtools/seqtools.cpp:106-212 (the "fml" mode with realignment): every read into one FermiAssembler, CorrectReads, PerformAssembly,
// every contig through BWAAligner::alignSequence(contig, "contigN", brv, false, 0.9, 10)
static int pipeline(const char *prefix, const char *fq, long n)
{
    FastqReader r(fq);
    FermiAssembler f;
    UnalignedSequence u;
    long got = 0;
    while (got < n && r.GetNextSequence(u)) { f.AddRead(u); ++got; }
    f.CorrectReads();
    f.PerformAssembly();
    std::vector<std::string> contigs = f.GetContigs();
    BWAIndexPtr idx(new BWAIndex());
    idx->LoadIndex(prefix);
    BWAAligner bwa(idx);
    for (size_t i = 0; i < contigs.size(); ++i) {
        BamRecordPtrVector brv;
        bwa.alignSequence(contigs[i], "contig" + std::to_string(i), brv, false, 0.9, 10);
        std::printf("CTG\t%zu\t%zu\t%zu\n", i, contigs[i].size(), brv.size());
        for (size_t k = 0; k < brv.size(); ++k)
            std::printf("REC\t%zu\t%d\t%d\t%d\t%d\t%s\n", i, brv[k]->ChrID(), brv[k]->Position(), brv[k]->AlignmentFlag(), brv[k]->MapQuality(), brv[k]->CigarString().c_str());
    }
    return 0;
}

int main(int argc, char **argv)
{
    try {
        if (argc >= 5 && std::string(argv[1]) == "pipeline") return pipeline(argv[2], argv[3], std::atol(argv[4]));
        if (argc >= 2 && std::string(argv[1]) == "cpu") return cpu();
        if (argc >= 4 && std::string(argv[1]) == "gpu") return gpu(argv[2], std::atol(argv[3]));
    } catch (const std::exception &e) {
        std::fprintf(stderr, "fml_api_test: %s\n", e.what());
        return 1;
    }
    std::fprintf(stderr, "usage: fml_api_test cpu | gpu <fastq> <n>\n");
    return 2;
}
