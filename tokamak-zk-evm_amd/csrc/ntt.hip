// ntt.hip — BLS12-381 scalar-field instantiation of the NTT (ntt_impl.inc) behind bls12_381_ntt*, tkmk_bintt: the field of
// the reference (every committed .r1cs carries this prime; packages/backend/libs/src/bivariate_polynomial/mod.rs:33-55,1422-1478).
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <mutex>

#include "common.h"
#include "ntt_plan.h"

#define TK_NTT_PRIMARY 1
#define TK_NTT_NS tk_ntt_bls12_381
#define TK_NTT_FR_PARAMS bls12_381_fr_params
#define TK_NTT_ABI_FR tkmk_fr
#define TK_NTT_SYM_ROOT bls12_381_get_root_of_unity
#define TK_NTT_SYM_INIT bls12_381_ntt_init_domain
#define TK_NTT_SYM_RELEASE bls12_381_ntt_release_domain
#define TK_NTT_SYM_DOMAIN_SIZE bls12_381_ntt_domain_size
#define TK_NTT_SYM_NTT bls12_381_ntt
#define TK_NTT_SYM_BINTT tkmk_bintt
#define TK_NTT_SYM_BINTT_PADDED tkmk_bintt_padded
#define TK_NTT_ROOT_GENERATOR TKMK_BLS12_381_FR_ROOT_GENERATOR
#define TK_NTT_ROOT_GENERATOR_ENV "TKMK_FR_ROOT_GENERATOR"
#include "ntt_impl.inc"
