// percall_bench -- what a SeqLib user who only swaps the library feels: the reference's own calling convention, one
// BWAAligner::alignSequence call per read (/root/reference/README.md:174-180, /root/reference/src/seqtools/seqtools.cpp:198-210),
// timed on a sample of the bench's reads; and the same loop through the deferred form (alignSequenceAsync + Flush).
//   percall_bench <index prefix> <reads.bin (fixed-length ASCII)> <read_len> <n_per_call> <n_async>
// Prints one JSON line.  Built by seqlib_amd/build.py with g++ against libseqlib_amd.so.
#include <malloc.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "SeqLib/BWAAligner.h"

using namespace SeqLib;

int main(int argc, char **argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: percall_bench <index prefix> <reads.bin> <read_len> <n_per_call> <n_async>\n"); return 2; }
    const std::string prefix = argv[1];
    const int read_len = std::atoi(argv[3]);
    const long n1 = std::atol(argv[4]), n2 = std::atol(argv[5]), n = std::max(n1, n2);
    std::vector<char> raw((size_t)n * (size_t)read_len);
    FILE *fp = std::fopen(argv[2], "rb");
    if (!fp || std::fread(raw.data(), 1, raw.size(), fp) != raw.size()) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
    std::fclose(fp);
    mallopt(M_TOP_PAD, 256 << 20); mallopt(M_TRIM_THRESHOLD, 1 << 30);
    try {
        BWAIndexPtr idx = std::make_shared<BWAIndex>();
        idx->LoadIndex(prefix);
        BWAAligner al(idx);
        std::vector<std::string> seqs, names;
        for (long i = 0; i < n; ++i) { seqs.emplace_back(raw.data() + (size_t)i * read_len, (size_t)read_len); names.push_back("r" + std::to_string(i)); }
        {   // warm-up: device handle, work areas
            BamRecordPtrVector w;
            for (int i = 0; i < 20 && i < n; ++i) al.alignSequence(seqs[(size_t)i], names[(size_t)i], w, false, 0.9, 10);
        }
        size_t rec1 = 0, rec2 = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (long i = 0; i < n1; ++i) {
            BamRecordPtrVector brv;          // the reference's loop: a fresh vector per read (README.md:176)
            al.alignSequence(seqs[(size_t)i], names[(size_t)i], brv, false, 0.9, 10);
            rec1 += brv.size();
        }
        const double s1 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::vector<BamRecordPtrVector> outs((size_t)n2);
        t0 = std::chrono::steady_clock::now();
        for (long i = 0; i < n2; ++i) al.alignSequenceAsync(seqs[(size_t)i], names[(size_t)i], outs[(size_t)i], false, 0.9, 10);
        al.Flush();
        const double s2 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (auto &v : outs) rec2 += v.size();
        std::printf("{\"value\": %.1f, \"unit\": \"reads/s\", \"us_per_call\": %.2f, \"reads\": %ld, \"records\": %zu, \"seconds\": %.4f, "
                    "\"path\": \"C++ SeqLib::BWAAligner::alignSequence, one call per read (the reference's calling convention): one GPU round trip per call\", "
                    "\"deferred\": {\"value\": %.1f, \"unit\": \"reads/s\", \"reads\": %ld, \"records\": %zu, \"seconds\": %.4f, "
                    "\"path\": \"the same loop through alignSequenceAsync + one Flush(): the queued calls run as one batch, records land in the callers' vectors\"}}\n",
                    (double)n1 / s1, s1 / (double)n1 * 1e6, n1, rec1, s1, (double)n2 / s2, n2, rec2, s2);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "percall_bench: %s\n", e.what());
        return 1;
    }
    return 0;
}
