// percall_bench -- what a SeqLib user who only swaps the library feels: the reference's own calling convention, one
// BWAAligner::alignSequence call per read (/root/reference/README.md:174-180, /root/reference/src/seqtools/seqtools.cpp:198-210),
// timed on a sample of the bench's reads; and the same loop through the deferred form (alignSequenceAsync + Flush).
//   percall_bench <index prefix> <reads.bin (fixed-length ASCII)> <read_len> <n_per_call> <n_async> [<caller threads> [<calls per thread>]]
// With caller threads (default 16): the reference's alignSequence is const and re-entrant (SeqLib/BWAAligner.h:51-63), so T threads may share one aligner, each
// looping over its own reads; here concurrent calls are combined into shared GPU round trips (BWAAligner.h, "flat combining").  Records are checked against the
// single-thread loop's (count per read; the draws differ with arrival order, as they do for the reference's threads).
// Prints one JSON line.  Built by seqlib_amd/build.py with g++ against libseqlib_amd.so.
#include <malloc.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include "SeqLib/BWAAligner.h"

using namespace SeqLib;

int main(int argc, char **argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: percall_bench <index prefix> <reads.bin> <read_len> <n_per_call> <n_async>\n"); return 2; }
    const std::string prefix = argv[1];
    const int read_len = std::atoi(argv[3]);
    const long n1 = std::atol(argv[4]), n2 = std::atol(argv[5]), n = std::max(n1, n2);
    const int T = argc > 6 ? std::atoi(argv[6]) : 16;
    const long per_thread = argc > 7 ? std::atol(argv[7]) : std::max(1L, std::min(n, 4 * n1) / std::max(1, T));
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
        // T caller threads on the one aligner, each the reference's loop over its own slice of the reads
        double s3 = 0; size_t rec3 = 0; long n3 = 0;
        uint64_t rounds0 = 0, calls0 = 0, rounds1 = 0, calls1 = 0;
        al.CombinedCallStats(rounds0, calls0);
        double batch_us = 0;
        if (T > 0) {          // what one shared round costs: T reads as one batch through the same entry a round uses
            UnalignedSequenceVector few;
            for (int i = 0; i < T && i < n; ++i) few.emplace_back(names[(size_t)i], seqs[(size_t)i]);
            std::vector<BamRecordPtrVector> o;
            al.alignSequences(few, o, false, 0.9, 10);
            const auto tb = std::chrono::steady_clock::now();
            for (int rep = 0; rep < 50; ++rep) { o.clear(); al.alignSequences(few, o, false, 0.9, 10); }
            batch_us = std::chrono::duration<double>(std::chrono::steady_clock::now() - tb).count() / 50 * 1e6;
        }
        if (T > 0) {
            n3 = std::min<long>(n, per_thread * T);
            const long each = n3 / T;
            n3 = each * T;
            std::atomic<size_t> recs{0};
            std::atomic<int> failed{0};
            std::vector<std::thread> th;
            t0 = std::chrono::steady_clock::now();
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t]() {
                    try {
                        size_t mine = 0;
                        for (long i = t * each; i < (t + 1) * each; ++i) {
                            BamRecordPtrVector brv;
                            al.alignSequence(seqs[(size_t)i], names[(size_t)i], brv, false, 0.9, 10);
                            mine += brv.size();
                            if (!brv.empty() && brv[0]->Qname() != names[(size_t)i]) failed.fetch_add(1);          // a caller must get ITS read's records
                        }
                        recs += mine;
                    } catch (...) { failed.fetch_add(1); }
                });
            for (auto &x : th) x.join();
            s3 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            rec3 = recs.load();
            if (failed.load()) { std::fprintf(stderr, "percall_bench: %d failures in the threaded loop\n", failed.load()); return 1; }
            al.CombinedCallStats(rounds1, calls1);
        }
        std::printf("{\"threads\": {\"value\": %.1f, \"unit\": \"reads/s\", \"caller_threads\": %d, \"reads\": %ld, \"records\": %zu, \"seconds\": %.4f, \"rounds\": %llu, "
                    "\"calls_per_round\": %.2f, \"us_per_batch_of_T_reads\": %.1f, "
                    "\"path\": \"T host threads share ONE const BWAAligner, each looping alignSequence over its own reads (re-entrant in the reference, SeqLib/BWAAligner.h:51-63): concurrent calls are "
                    "combined into shared GPU round trips, draws in arrival order\"}, ",
                    n3 ? (double)n3 / s3 : 0.0, T, n3, rec3, s3, (unsigned long long)(rounds1 - rounds0), rounds1 > rounds0 ? (double)(calls1 - calls0) / (double)(rounds1 - rounds0) : 0.0, batch_us);
        std::printf("\"value\": %.1f, \"unit\": \"reads/s\", \"us_per_call\": %.2f, \"reads\": %ld, \"records\": %zu, \"seconds\": %.4f, "
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
