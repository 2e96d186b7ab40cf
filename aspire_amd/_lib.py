"""ctypes binding of libasmc_hip.so (the C ABI declared in include/asmc.h).

The product path has NO CPU fallback: if the shared library is missing `load()` raises, and every
compute entry point needs a HIP device.  (`oracle/` is test infrastructure and is never imported
from this package.)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ASMC_LIB_PATH") or os.path.join(_HERE, "libasmc_hip.so")

ASMC_OK = 0
ASMC_ERR_UNSUPPORTED = -4
ASMC_F64, ASMC_F32 = 0, 1
ASMC_CDF_EXACT, ASMC_CDF_FAST = 0, 1
ASMC_CDF_NORMALIZE = 0x100
ASMC_NOISE_F64, ASMC_NOISE_F32 = 0, 1
ASMC_MAX_BETAS = 32
ASMC_MAX_COMPONENTS = 8
ASMC_MAX_DIMS = 256
COUNT_HOOK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
ASMC_BIS_REC = 40
ASMC_SELECT_THREADS = 262144
ASMC_STUDENT_MAX_ROWS = 16384
ASMC_ABI_VERSION = 23
ASMC_FLOW_COUPLING, ASMC_FLOW_MAF = 0, 1  # asmc_coupling.kind
ASMC_MAX_COUNT_CELLS = 64  # asmc_pcn_set_count_cells
ASMC_CDF_REC = 9
ASMC_CDF_STATE = 36


class AsmcMixture(ctypes.Structure):
    _fields_ = [
        ("n_components", c_int32),
        ("reserved", c_int32),
        ("logw_dev", c_void_p),
        ("mu_dev", c_void_p),
        ("prec_dev", c_void_p),
    ]


class AsmcPcnParams(ctypes.Structure):
    _fields_ = [
        ("d", c_int32),
        ("x_dtype", c_int32),
        ("beta", c_double),
        ("mu_dev", c_void_p),
        ("L_dev", c_void_p),
        ("Linv_dev", c_void_p),
        ("log_likelihood", AsmcMixture),
        ("log_prior", AsmcMixture),
        ("log_q", AsmcMixture),
        ("seed", c_uint64),
        ("gid0", c_uint64),
        ("target_accept", c_double),
        ("adapt", c_int32),
        ("noise", c_int32),
        ("nu", c_double),
    ]


class AsmcCoupling(ctypes.Structure):
    _fields_ = [
        ("dims", c_int32),
        ("n_layers", c_int32),
        ("hidden", c_int32),
        ("kind", c_int32),  # ASMC_FLOW_COUPLING / ASMC_FLOW_MAF
        ("packed_dev", c_void_p),
        ("loc_dev", c_void_p),
        ("scale_dev", c_void_p),
        ("log_scale_sum", c_double),
        ("affine", c_int32),  # ASMC_AFFINE_TANH (0) / ASMC_AFFINE_SOFTCLIP (1)
        ("reserved", c_int32),
    ]


class AsmcTransform(ctypes.Structure):
    _fields_ = [
        ("d", c_int32),
        ("hints", c_int32),
        ("kind_dev", c_void_p),
        ("periodic_dev", c_void_p),
        ("lower_dev", c_void_p),
        ("upper_dev", c_void_p),
        ("mean_dev", c_void_p),
        ("std_dev", c_void_p),
        ("eps", c_double),
        ("unit_logj", c_double),
        ("affine_logj", c_double),
    ]


_vp, _d, _i, _i64, _u64, _u32 = c_void_p, c_double, c_int, c_int64, c_uint64, c_uint32
_pd, _pi64 = POINTER(c_double), POINTER(c_int64)

# name -> (restype, argtypes); must list exactly the functions include/asmc.h declares
SIGNATURES = {
    "asmc_abi_version": (_i, []),
    "asmc_last_error": (c_char_p, []),
    "asmc_device_count": (_i, [POINTER(c_int)]),
    "asmc_ctx_create": (_i, [POINTER(c_void_p), _i, _i64, _i]),
    "asmc_ctx_destroy": (_i, [_vp]),
    "asmc_profile_enable": (_i, [_vp, _i]),
    "asmc_profile_report": (_i, [_vp, c_char_p, _i64]),
    "asmc_weights_max": (_i, [_vp, _i64, _vp, _vp, _vp, _d, _pd, _i, _pd, _pi64, _vp]),
    "asmc_weights_sums": (_i, [_vp, _i64, _vp, _vp, _vp, _d, _pd, _pd, _pd, _i, _pd, _vp]),
    "asmc_weights_stats": (_i, [_vp, _i64, _vp, _vp, _vp, _d, _pd, _i, _pd, _vp]),
    "asmc_weights_m2_lse": (_i, [_vp, _i64, _vp, _vp, _vp, _d, _d, _d, _d, _d, _d, _pd, _vp]),
    "asmc_find_beta": (_i, [_vp, _i64, _vp, _vp, _vp, _d, _d, _d, _pd, _vp]),
    "asmc_importance_step": (_i, [_vp, _i64, _vp, _vp, _vp, _d, _d, _d, _vp, _i64, _vp, _vp, _vp, _vp]),
    "asmc_importance_result": (_i, [_vp, _pd, _vp]),
    "asmc_importance_available": (_i, [_vp]),
    "asmc_importance_result_enqueue": (_i, [_vp, _vp]),
    "asmc_pcn_flow_nonfinite": (_i64, [_vp]),
    "asmc_pcn_lq_nan": (_i64, [_vp]),
    "asmc_pcn_set_count_hook": (_i, [_vp, _vp, _vp, _vp, _i64]),
    "asmc_set_rccl": (_i, [_vp, _vp, _vp]),
    "asmc_set_rccl_allgather": (_i, [_vp, _vp]),
    "asmc_rccl_all_gather": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "asmc_find_beta_shard_rounds": (_i, [_vp, _i64, _vp, _vp, _vp, _d, _d, _d, _i, _i64, _vp, _vp, _i, _i, _vp]),
    "asmc_pcn_set_count_rccl": (_i, [_vp, _vp, _i64]),
    "asmc_pcn_ysplit_begin": (_i, [_vp, _i64, _vp, POINTER(AsmcPcnParams), _d, _vp]),
    "asmc_pcn_ysplit_propose": (_i, [_vp, _i64, POINTER(AsmcPcnParams), _u32, _vp, _vp]),
    "asmc_pcn_ysplit_propose_tr": (_i, [_vp, _i64, POINTER(AsmcPcnParams), _u32, POINTER(AsmcTransform), _vp, POINTER(AsmcMixture), _i,
                                   _vp, _vp, _vp, _vp]),
    "asmc_pcn_ysplit_accept": (_i, [_vp, _i64, POINTER(AsmcPcnParams), _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i,
                                    _vp]),
    "asmc_pcn_ysplit_end": (_i, [_vp, _i64, _vp, POINTER(AsmcPcnParams), _vp]),
    "asmc_pcn_split_begin": (_i, [_vp, _d, _vp]),
    "asmc_pcn_split_adapt": (_i, [_vp, _i64, _d, _i, _i, _vp]),
    "asmc_pcn_split_end": (_i, [_vp, _i, POINTER(c_int64), _pd, _pd, _vp]),
    "asmc_student_estep": (_i, [_vp, _i64, _i, _vp, _pd, _pd, _d, _vp, _pd, _vp]),
    "asmc_student_scale": (_i, [_vp, _i64, _i, _vp, _vp, _pd, _vp, _vp]),
    "asmc_find_beta_shard_reduce": (_i, [_vp, _i64, _vp, _vp, _vp, _d, _i, _vp, _vp]),
    "asmc_find_beta_shard_decide": (_i, [_vp, _vp, _i, _i64, _d, _d, _d, _i, _vp]),
    "asmc_find_beta_shard_result": (_i, [_vp, _pd, _vp]),
    "asmc_weights_m2_lse_dev": (_i, [_vp, _i64, _vp, _vp, _vp, _d, _d, _d, _d, _d, _d, _vp, _vp]),
    "asmc_cdf_total_dev": (_i, [_vp, _vp, _vp]),
    "asmc_pcg64_select_stage_len": (_i64, [_i64]),
    "asmc_pcg64_select": (_i, [_vp, POINTER(c_uint64), _i64, _d, _d, _vp, POINTER(c_int64), _vp]),
    "asmc_pcg64_select_compact": (_i, [_vp, _i64, _vp, _vp, _vp]),
    "asmc_weights_m2": (_i, [_vp, _i64, _vp, _vp, _vp, _d, _d, _d, _d, _pd, _vp]),
    "asmc_log_weights": (_i, [_vp, _i64, _vp, _vp, _vp, _d, _d, _d, _vp, _vp]),
    "asmc_normalized_weights": (_i, [_vp, _i64, _vp, _vp, _vp, _d, _d, _d, _d, _vp, _vp]),
    "asmc_count_nonfinite": (_i, [_vp, _i64, _vp, _pi64, _pi64, _vp]),
    "asmc_cdf": (_i, [_vp, _i64, _vp, _vp, _i, _d, _pd, _vp]),
    "asmc_cdf_normalize": (_i, [_vp, _i64, _vp, _d, _vp]),
    "asmc_cdf_normalize_last": (_i, [_vp, _i64, _vp, _vp]),
    "asmc_pcg64_uniforms": (_i, [_vp, POINTER(c_uint64), _u64, _i64, _vp, _vp]),
    "asmc_cdf_shard_tiles": (_i64, [_i64]),
    "asmc_cdf_shard_records": (_i, [_vp, _i64, _vp, _vp, _d, _i, _vp, _vp]),
    "asmc_cdf_shard_records_dev": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "asmc_cdf_shard_finish_select": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "asmc_weights_m2_lse_shard": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i64, _d, _d, _d, _i, _vp]),
    "asmc_find_beta_shard_round": (_i, [_vp, _i64, _vp, _vp, _vp, _d, _d, _d, _i, _i64, _i, _vp, _vp, _vp]),
    "asmc_normalized_weights_shard": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _d, _vp, _vp, _vp, _vp, _i, _vp]),
    "asmc_rec_token": (_i64, [_vp]),
    "asmc_rec_claim": (_i, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "asmc_shard_step_result": (_i, [_vp, _vp, _i, _pd, _vp]),
    "asmc_shard_step_finish": (_i, [_vp, _vp, _i, _i, _i64, _vp, _vp, _i64, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                   _pd, POINTER(ctypes.c_int), _vp]),
    "asmc_cdf_shard_chain": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i, _i, _vp, _vp]),
    "asmc_cdf_shard_finish": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "asmc_select_range": (_i, [_vp, _i64, _vp, _vp, _vp, _pi64, _vp]),
    "asmc_select_range_dev": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "asmc_systematic_uniforms": (_i, [_vp, _i64, _i64, _i64, _d, _vp, _vp, _vp]),
    "asmc_search": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "asmc_gather": (_i, [_vp, _i64, _i64, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "asmc_gaussian_draw": (_i, [_vp, _i64, _i, _i, _vp, _vp, _u64, _u64, _u32, _vp, _vp, _vp]),
    "asmc_mixture_logpdf": (_i, [_vp, _i64, _i, _i, _vp, POINTER(AsmcMixture), _vp, _vp]),
    "asmc_mixture_logpdf_premap": (_i, [_vp, _i64, _i, _i, _vp, _vp, POINTER(AsmcMixture), _vp, _vp]),
    "asmc_compact_valid": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _pi64, _vp]),
    "asmc_colsum": (_i, [_vp, _i64, _i, _i, _vp, _pd, _vp]),
    "asmc_centered_gram": (_i, [_vp, _i64, _i, _i, _vp, _pd, _pd, _vp]),
    "asmc_mean_gram": (_i, [_vp, _i64, _i, _i, _vp, _i64, _i, _pd, _pd, _vp]),
    "asmc_mean_gram_enqueue": (_i, [_vp, _i64, _i, _i, _vp, _i64, _i, _vp]),
    "asmc_mean_gram_fetch": (_i, [_vp, _i, _pd, _pd, _vp]),
    "asmc_reference_factor": (_i, [_vp, _i, _i64, _i64, _pd, _pd, _vp, _vp]),
    "asmc_colsum_dev": (_i, [_vp, _i64, _i, _i, _vp, _i, _vp, _vp]),
    "asmc_centered_gram_dev": (_i, [_vp, _i64, _i, _i, _vp, _vp, _i64, _vp, _vp]),
    "asmc_reference_factor_dev": (_i, [_vp, _i, _i64, _i64, _vp, _vp, _vp, _vp]),
    "asmc_student_fit": (_i, [_vp, _i64, _i, _vp, _i, _d, _d, _vp, _vp, _vp, _pd, _vp]),
    "asmc_reference_factor_status": (_i, [_vp, POINTER(ctypes.c_int)]),
    "asmc_reference_factor_generation": (_i64, [_vp]),
    "asmc_reference_factor_status_of": (_i, [_vp, _i64, POINTER(ctypes.c_int)]),
    "asmc_pcn_mutate": (
        _i,
        [_vp, _i64, _vp, _vp, _vp, _vp, POINTER(AsmcPcnParams), _i, _u32, _pd, _pi64, _pd, _vp],
    ),
    "asmc_pcn_propose": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _d, _u64, _u64, _u32, _vp]),
    "asmc_pcn_accept": (
        _i,
        [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _u64, _u64, _u32, _pi64, _vp],
    ),
    "asmc_flow_layout": (_i, [_i, _i, _i]),
    "asmc_pcn_set_count_cells": (_i, [_vp, _i]),
    "asmc_coupling_pack_floats": (_i64, [_i, _i, _i]),
    "asmc_coupling_pack": (_i, [_i, _i, _i, POINTER(c_void_p), POINTER(c_void_p), _vp]),
    "asmc_maf_pack_floats": (_i64, [_i, _i, _i]),
    "asmc_maf_pack": (_i, [_i, _i, _i, POINTER(c_void_p), POINTER(c_void_p), _vp]),
    "asmc_coupling_logprob": (_i, [_vp, _i64, _i, _vp, POINTER(AsmcCoupling), _vp, _vp]),
    "asmc_coupling_sample": (_i, [_vp, _i64, _i, POINTER(AsmcCoupling), _u64, _u64, _u32, _vp, _vp, _vp]),
    "asmc_transform_forward": (_i, [_vp, _i64, _i, _vp, _vp, _vp, POINTER(AsmcTransform), _vp]),
    "asmc_transform_inverse": (_i, [_vp, _i64, _i, _vp, _vp, _vp, POINTER(AsmcTransform), _vp]),
    "asmc_pcn_flow_work_bytes": (_i64, [_i64, _i, _i]),
    "asmc_pcn_mutate_flow": (
        _i,
        [_vp, _i64, _vp, _vp, _vp, _vp, POINTER(AsmcPcnParams), POINTER(AsmcCoupling), _vp, _i64, _i, _u32, _pd, _pi64, _pd, _vp],
    ),
    "asmc_pcn_mutate_flow_enqueue": (
        _i,
        [_vp, _i64, _vp, _vp, _vp, _vp, POINTER(AsmcPcnParams), POINTER(AsmcCoupling), _vp, _i64, _i, _u32, _d, _vp],
    ),
    "asmc_pcn_mutate_flow_result": (_i, [_vp, _i, _pd, _pi64, _pd, _vp]),
}

_lib = None


class AsmcError(RuntimeError):
    pass


def load():
    """Load libasmc_hip.so; raises (no fallback) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AsmcError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C aspire_amd/csrc). "
            "aspire_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.asmc_abi_version() != ASMC_ABI_VERSION:
        raise AsmcError("libasmc_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != ASMC_OK:
        msg = load().asmc_last_error()
        raise AsmcError(f"{what or 'asmc call'} failed (rc={rc}): {msg.decode() if msg else ''}")
