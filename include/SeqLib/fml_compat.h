// fml_compat.h -- the slice of fermi-lite's fml.h / mag.h / bfc.h that /root/reference/SeqLib/FermiAssembler.h:10-15 and BFC.h:3-6
// pull in (`fermi-lite` is an empty submodule of the reference and is not in this image): the option and unitig record types under
// their fermi-lite names, laid over the C-ABI types of include/seqlib_amd_fml.h, so that code written against the reference's
// headers (opt.min_asm_ovlp, opt.mag_opt.flag |= MAG_F_AGGRESSIVE, u->seq, u->ovlp[j].id ...) compiles unchanged.
#pragma once
#include "seqlib_amd_fml.h"

typedef slx_magopt   magopt_t;
typedef slx_fml_opt  fml_opt_t;
typedef slx_fml_ovlp fml_ovlp_t;
typedef slx_fml_utg  fml_utg_t;

#ifndef MAG_F_AGGRESSIVE
#define MAG_F_AGGRESSIVE SLX_MAG_F_AGGRESSIVE
#define MAG_F_POPOPEN    SLX_MAG_F_POPOPEN
#define MAG_F_NO_SIMPL   SLX_MAG_F_NO_SIMPL
#endif
#ifndef BFC_EC_MIN_COV_COEF
#define BFC_EC_MIN_COV_COEF .1
#endif

inline void fml_opt_init(fml_opt_t *opt) { slx_fml_opt_init(opt); }
