// slx_fml_asm.hip -- fml_assemble behind the C-ABI (placeholder until the assembly stage lands: fails loudly)
#include "slx_fml_internal.h"

extern "C" int slx_fml_assemble(slx_fml *f, const slx_fml_opt *opt, const char *bases, const char *quals, const uint64_t *offs, int64_t n_reads,
                                const int64_t *win_off, int n_win, slx_fml_utg **utgs, int *n_utg)
{
    slx_set_error("slx_fml_assemble: not built yet");
    return SLX_EUNSUPPORTED;
}
extern "C" int slx_fml_direct_assemble(slx_fml *f, slx_fml_opt *opt, float kcov, const char *bases, const uint64_t *offs, int64_t n_reads, slx_fml_utg **utgs, int *n_utg)
{
    slx_set_error("slx_fml_direct_assemble: not built yet");
    return SLX_EUNSUPPORTED;
}
extern "C" void slx_fml_utgs_free(int n_utg, slx_fml_utg *utgs) {}
