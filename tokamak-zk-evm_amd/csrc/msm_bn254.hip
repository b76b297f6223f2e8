// msm_bn254.hip — BN254 (alt_bn128) G1 instantiation of the Pippenger MSM (msm_impl.inc) behind bn254_msm /
// tkmk_bn254_msm_multi (include/tkmk.h).  The reference has no BN254 path; BASELINE.json's
// configs name a 2^24-point BN254 G1 MSM, so the same kernels are instantiated over the 254-bit fields:
// base field 8 x u32 saturated / 10 x 28-bit unsaturated limbs (field_params.h), y^2 = x^3 + 3.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "common.h"
#include "ec_u.h"

#define TK_MSM_NS tk_msm_bn254
#define TK_MSM_FR_PARAMS bn254_fr_params
#define TK_MSM_FQ_PARAMS bn254_fq_params
#define TK_MSM_SCALAR_BITS 254
#define TK_MSM_ABI_FR tkmk_bn254_fr
#define TK_MSM_ABI_AFFINE tkmk_bn254_g1_affine
#define TK_MSM_ABI_PROJ tkmk_bn254_g1_projective
#define TK_MSM_ABI_JOB tkmk_bn254_msm_job
#define TK_MSM_SYM_MSM bn254_msm
#define TK_MSM_SYM_MULTI tkmk_bn254_msm_multi
#define TK_MSM_SYM_PRECOMPUTE bn254_msm_precompute_bases
#include "msm_impl.inc"
