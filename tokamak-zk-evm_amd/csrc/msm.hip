// msm.hip — BLS12-381 G1 instantiation of the Pippenger MSM (msm_impl.inc) behind bls12_381_msm / tkmk_msm_multi
// (include/tkmk.h): the curve of the reference (packages/backend/Cargo.toml:23, libs/src/iotools/mod.rs:2093-2099).
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "common.h"
#include "ec_u.h"

#define TK_MSM_PRIMARY 1
#define TK_MSM_NS tk_msm_bls12_381
#define TK_MSM_FR_PARAMS bls12_381_fr_params
#define TK_MSM_FQ_PARAMS bls12_381_fq_params
#define TK_MSM_SCALAR_BITS 255
#define TK_MSM_ABI_FR tkmk_fr
#define TK_MSM_ABI_AFFINE tkmk_g1_affine
#define TK_MSM_ABI_PROJ tkmk_g1_projective
#define TK_MSM_ABI_JOB tkmk_msm_job
#define TK_MSM_SYM_MSM bls12_381_msm
#define TK_MSM_SYM_MULTI tkmk_msm_multi
#define TK_MSM_SYM_PRECOMPUTE bls12_381_msm_precompute_bases
#define TK_MSM_ABI_JOB_EX tkmk_msm_job_ex
#define TK_MSM_SYM_MULTI_EX tkmk_msm_multi_ex
#define TK_MSM_SYM_CONVERT bls12_381_msm_convert_bases
#include "msm_impl.inc"
