// bamrec_bench -- times the C++ drop-in class end to end for bench.py's `value_bamrecords`:
//     UnalignedSequenceVector -> SeqLib::BWAAligner::alignSequences -> std::vector<BamRecordPtrVector>
// i.e. the GPU path behind /root/reference/src/BWAAligner.cpp:89-146 PLUS the BamRecord materialisation of :151-248
// (one shared_ptr<BamRecord> + bam1_t blob + three tag appends per hit), on a sample of the bench's reads.
//   bamrec_bench <index prefix> <reads.bin (fixed-length ASCII)> <read_len> <n_reads>
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
    if (argc < 5) { std::fprintf(stderr, "usage: bamrec_bench <index prefix> <reads.bin> <read_len> <n_reads>\n"); return 2; }
    const std::string prefix = argv[1];
    const int read_len = std::atoi(argv[3]);
    const long n = std::atol(argv[4]);
    std::vector<char> raw((size_t)n * (size_t)read_len);
    FILE *fp = std::fopen(argv[2], "rb");
    if (!fp || std::fread(raw.data(), 1, raw.size(), fp) != raw.size()) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
    std::fclose(fp);
    // glibc: let the per-thread arenas grow in large steps and never shrink (10 M records are 30 M allocations from 16 threads; with the
    // default 128 KB steps every step is an mprotect call that serialises the threads on the process's memory map).  SLX_NO_MALLOPT=1 keeps glibc's defaults.
    if (!std::getenv("SLX_NO_MALLOPT")) { mallopt(M_TOP_PAD, 256 << 20); mallopt(M_TRIM_THRESHOLD, 1 << 30); }
    try {
        BWAIndexPtr idx = std::make_shared<BWAIndex>();
        idx->LoadIndex(prefix);
        BWAAligner al(idx);
        UnalignedSequenceVector reads;
        reads.reserve((size_t)n);
        for (long i = 0; i < n; ++i)
            reads.emplace_back("r" + std::to_string(i), std::string(raw.data() + (size_t)i * read_len, (size_t)read_len));
        std::vector<BamRecordPtrVector> out;
        {   // warm-up: device handle, work areas, pinned block
            UnalignedSequenceVector head(reads.begin(), reads.begin() + std::min<long>(n, 300000));
            al.alignSequences(head, out, false, 0.9, 10);
        }
        double best = 1e30;
        size_t records = 0;
        for (int rep = 0; rep < 2; ++rep) {
            out.clear();
            const auto t0 = std::chrono::steady_clock::now();
            al.alignSequences(reads, out, false, 0.9, 10);
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            best = s < best ? s : best;
            records = 0;
            for (auto &v : out) records += v.size();
        }
        std::printf("{\"value\": %.1f, \"unit\": \"reads/s\", \"reads\": %ld, \"records\": %zu, \"seconds\": %.4f, \"host_threads\": %u, "
                    "\"path\": \"C++ SeqLib::BWAAligner::alignSequences: UnalignedSequenceVector -> chunks: pack into pinned staging (host threads) | GPU | BamRecords (host threads), overlapped -> std::vector<BamRecordPtrVector> (sample of the bench reads)\"}\n",
                    (double)n / best, n, records, best, std::getenv("SEQLIB_AMD_THREADS") ? (unsigned)std::atoi(std::getenv("SEQLIB_AMD_THREADS"))
                                                      : (detail::effective_cpus() < 8 ? std::min(16u, detail::effective_cpus() * 8) : detail::effective_cpus()));          // (alignSequences' own rule)
    } catch (const std::exception &e) {
        std::fprintf(stderr, "bamrec_bench: %s\n", e.what());
        return 1;
    }
    return 0;
}
