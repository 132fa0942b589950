// Sanitizer run of the product's HOST-side index code (seqlib_amd/csrc/slx_index.cpp: bwa's file formats, .alt parsing): built with
// -fsanitize=address,undefined by tests/test_sanitizers.py.  The two device builders it can call are not linked on this CPU-only
// build: they are replaced by refusals (nothing here constructs an index).
//   san_index_test <index prefix> <tmp prefix>
#include <cstdio>
#include <cstring>
#include <string>
#include "slx_internal.h"

int slx_gpu_build_fm(slx_index *, const uint8_t *, uint64_t) { slx_set_error("no device in the sanitizer build"); return SLX_ENODEVICE; }
int slx_gpu_build_fm64(slx_index *, const uint8_t *, uint64_t) { slx_set_error("no device in the sanitizer build"); return SLX_ENODEVICE; }

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    slx_index *idx = nullptr;
    if (slx_index_load(argv[1], &idx) != SLX_OK) { std::fprintf(stderr, "%s\n", slx_last_error()); return 1; }
    const std::string out = argv[2];
    if (slx_index_write(idx, out.c_str()) != SLX_OK) return 1;
    // an .alt next to the copy: header line, names with fields, CR, an unknown name, a last line without newline
    FILE *fp = std::fopen((out + ".alt").c_str(), "w");
    std::fprintf(fp, "@HD\tx\n%s\t0\tchr\r\nnot_there\n%s", slx_index_name(idx, slx_index_nseq(idx) - 1), slx_index_name(idx, 0));
    std::fclose(fp);
    slx_index *again = nullptr;
    if (slx_index_load(out.c_str(), &again) != SLX_OK) return 1;
    int n_alt = 0;
    for (const slx_ann &a : again->anns) n_alt += a.is_alt;
    std::printf("nseq=%d l_pac=%lld alt=%d\n", slx_index_nseq(again), (long long)slx_index_l_pac(again), n_alt);
    slx_opt o; slx_opt_init(&o);
    int8_t m[25]; slx_fill_scmat(3, 7, m);
    slx_index *none = nullptr;
    if (slx_index_load((out + ".missing").c_str(), &none) == SLX_OK) return 1;
    const char *nm[1] = {"x"}; const char *sq[1] = {"ACGT"}; int64_t ln[1] = {4};
    if (slx_index_build(nm, sq, ln, 1, &none) != SLX_ENODEVICE) return 1;          // refuses without a device, and leaks nothing
    slx_index_free(again);
    slx_index_free(idx);
    return n_alt == 1 ? 0 : 1;
}
