// ntt_bn254.hip — BN254 scalar-field instantiation of the NTT (ntt_impl.inc) behind bn254_ntt*, tkmk_bn254_bintt.  The
// reference has no BN254 path; BASELINE.json's configs[0] names a BN254 scalar-field NTT, so the same
// kernels are instantiated over the 254-bit field (two-adicity 28, w_{2^28} = 5^((r-1)/2^28) as the library's own root).
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <mutex>

#include "common.h"
#include "ntt_plan.h"

#define TK_NTT_NS tk_ntt_bn254
#define TK_NTT_FR_PARAMS bn254_fr_params
#define TK_NTT_ABI_FR tkmk_bn254_fr
#define TK_NTT_SYM_ROOT bn254_get_root_of_unity
#define TK_NTT_SYM_INIT bn254_ntt_init_domain
#define TK_NTT_SYM_RELEASE bn254_ntt_release_domain
#define TK_NTT_SYM_DOMAIN_SIZE bn254_ntt_domain_size
#define TK_NTT_SYM_NTT bn254_ntt
#define TK_NTT_SYM_BINTT tkmk_bn254_bintt
#define TK_NTT_ROOT_GENERATOR 5
#include "ntt_impl.inc"
