"""ctypes binding of libmispec.so (the C ABI declared in include/mispec.h).

There is no CPU fallback anywhere in this package: if the shared library is missing, or no HIP
device is visible when a context is created, an exception is raised.
"""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libmispec.so")
EXTRAS_LIB_PATH = os.path.join(_PKG, "libmispec_extras.so")  # Davidson solver, complex factorisation (mispec_extras.h)
EXTRAS_PREFIXES = ("mispec_davidson_", "mispec_zdense_", "mispec_zfac_")
CSRC = os.path.join(_PKG, "csrc")

MISPEC_OK, MISPEC_EINVAL, MISPEC_ELOGIC, MISPEC_ERUNTIME = 0, -1, -2, -3


class MispecError(RuntimeError):
    """HIP / RCCL failure or a failed small decomposition (std::runtime_error in the C++ API)."""


def build_library(force=False, verbose=False):
    """Compile spectra_amd/libmispec.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else [])
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
    if out.returncode != 0:
        raise RuntimeError("building libmispec.so failed")
    return LIB_PATH


op_fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))
device_op_fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
allgather_fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
allreduce_fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
exchange_fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p,
                          C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p)


class Comm(C.Structure):
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("allgather", allgather_fn), ("allreduce_sum", allreduce_fn),
                ("user", C.c_void_p), ("exchange", exchange_fn)]


class Profile(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_spmv", "n_vtf", "n_gemv", "n_scale", "n_compress", "n_small", "n_host_sync")] + \
               [(n, C.c_double) for n in ("ms_spmv", "ms_vtf", "ms_gemv", "ms_scale", "ms_compress", "ms_small", "spmv_bytes",
                                             "bytes_vtf", "bytes_gemv", "bytes_compress")] + \
               [(n, C.c_int64) for n in ("n_reduce", "n_exchange", "n_exchange_wait", "n_allreduce")] + \
               [(n, C.c_double) for n in ("ms_reduce", "ms_exchange", "ms_exchange_wait", "ms_allreduce")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); must list every symbol include/mispec.h declares (tests check that)
SIGNATURES = {
    "mispec_last_error": (C.c_char_p, []),
    "mispec_version": (C.c_char_p, []),
    "mispec_ctx_create": (C.c_int, [C.c_int, _vp, _vpp]),
    "mispec_ctx_destroy": (C.c_int, [_vp]),
    "mispec_ctx_sync": (C.c_int, [_vp]),
    "mispec_ctx_stream": (_vp, [_vp]),
    "mispec_ctx_set_comm": (C.c_int, [_vp, C.POINTER(Comm)]),
    "mispec_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "mispec_ctx_set_comm_rccl": (C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p]),
    "mispec_loopback_create": (C.c_int, [C.c_int, _vpp]),
    "mispec_loopback_attach": (C.c_int, [_vp, _vp, C.c_int]),
    "mispec_loopback_destroy": (C.c_int, [_vp]),
    "mispec_shard_block": (C.c_int64, [C.c_int64, C.c_int]),
    "mispec_shard_range": (C.c_int, [C.c_int64, C.c_int, C.c_int, _lp, _lp]),
    "mispec_csr_upload": (C.c_int, [_vp, C.c_int64, C.c_int64, _ip, _ip, _dp, _vpp]),
    "mispec_csr_from_csc": (C.c_int, [_vp, C.c_int64, C.c_int64, _ip, _ip, _dp, _vpp]),
    "mispec_csr_from_triangle": (C.c_int, [_vp, C.c_int64, _ip, _ip, _dp, C.c_char, C.c_int, _vpp]),
    "mispec_csr_synth_band": (C.c_int, [_vp, C.c_int64, C.c_uint64, _lp, C.c_int, C.c_int, _vpp]),
    "mispec_csr_destroy": (C.c_int, [_vp]),
    "mispec_csr_rows": (C.c_int64, [_vp]),
    "mispec_csr_cols": (C.c_int64, [_vp]),
    "mispec_csr_local_rows": (C.c_int64, [_vp]),
    "mispec_csr_local_nnz": (C.c_int64, [_vp]),
    "mispec_csr_offset_codes": (C.c_int, [_vp]),
    "mispec_csr_use_offset_codes": (C.c_int, [_vp, C.c_int]),
    "mispec_csr_use_windows": (C.c_int, [_vp, C.c_int]),
    "mispec_csr_windows_info": (C.c_int, [_vp, _lp, _lp, _lp]),
    "mispec_csr_windows_table": (C.c_int, [_vp, _ip, C.c_int64]),
    "mispec_csr_windows_in_use": (C.c_int, [_vp]),
    "mispec_csr_windows_host": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, _ip, _ip, _ip]),
    "mispec_csr_spmv_format": (C.c_int, [_vp]),
    "mispec_csr_set_spmv_format": (C.c_int, [_vp, C.c_int]),
    "mispec_csr_spmv_bytes": (C.c_double, [_vp, C.c_int]),
    "mispec_csr_coeff": (C.c_int, [_vp, C.c_int64, C.c_int64, _dp]),
    "mispec_csr_download": (C.c_int, [_vp, _ip, _ip, _dp]),
    "mispec_spmv": (C.c_int, [_vp, _vp, _vp]),
    "mispec_spmv_host": (C.c_int, [_vp, _dp, _dp]),
    "mispec_spmm_host": (C.c_int, [_vp, _dp, C.c_int64, C.c_int, _dp, C.c_int64]),
    "mispec_spmv_time": (C.c_int, [_vp, _vp, _vp, C.c_int, C.POINTER(C.c_float)]),
    "mispec_symshift_level_plan": (C.c_int, [C.c_int64, C.c_int64, C.c_int, _lp, _lp, _lp, _lp]),
    "mispec_symshift_create": (C.c_int, [_vp, C.c_int64, _ip, _ip, _dp, C.c_char, C.c_int, _vpp]),
    "mispec_symshift_destroy": (C.c_int, [_vp]),
    "mispec_symshift_rows": (C.c_int64, [_vp]),
    "mispec_symshift_bandwidth": (C.c_int, [_vp, _lp, _lp, C.POINTER(C.c_int)]),
    "mispec_csr_tiles_info": (C.c_int, [_vp, _lp, _lp, _lp, _lp]),
    "mispec_csr_staged_info": (C.c_int, [_vp, _lp, _lp, _lp, _lp]),
    "mispec_csr_reorder": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    "mispec_csr_reordering": (C.c_int, [_vp, _dp, _dp]),
    "mispec_csr_permutation": (C.c_int, [_vp, _ip]),
    "mispec_staged_spmv_host": (C.c_int, [C.c_int64, C.c_int64, _ip, _ip, _dp, _dp, _dp, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "mispec_tiles_spmv_host": (C.c_int, [C.c_int64, C.c_int64, _ip, _ip, _dp, _dp, _dp, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "mispec_rcm_order": (C.c_int, [C.c_int64, _ip, _ip, C.c_int, _ip, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "mispec_symshift_set_shift": (C.c_int, [_vp, C.c_double]),
    "mispec_symshift_solve": (C.c_int, [_vp, _vp, _vp]),
    "mispec_symshift_refinement_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int64), _dp, _dp]),
    "mispec_symshift_solve_host": (C.c_int, [_vp, _dp, _dp]),
    "mispec_fac_create": (C.c_int, [_vp, _vp, op_fn, _vp, C.c_int64, C.c_int, C.c_int, _vpp]),
    "mispec_fac_create_shiftsolve": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vpp]),
    "mispec_fac_create_device_op": (C.c_int, [_vp, device_op_fn, _vp, C.c_int64, C.c_int, C.c_int, _vpp]),
    "mispec_fac_create_dense": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vpp]),
    "mispec_dense_upload": (C.c_int, [_vp, C.c_int64, C.c_int64, _dp, C.c_int64, C.c_int, C.c_char, _vpp]),
    # complex scalars (mispec_extras.h): interleaved (re, im) doubles
    "mispec_zdense_upload": (C.c_int, [_vp, C.c_int64, C.c_int64, _dp, C.c_int64, C.c_int, C.c_char, _vpp]),
    "mispec_zdense_destroy": (C.c_int, [_vp]),
    "mispec_zdense_rows": (C.c_int64, [_vp]),
    "mispec_zdense_cols": (C.c_int64, [_vp]),
    "mispec_zdense_gemv_host": (C.c_int, [_vp, _dp, _dp]),
    "mispec_zdense_coeff": (C.c_int, [_vp, C.c_int64, C.c_int64, _dp]),
    "mispec_zfac_create_dense": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vpp]),
    "mispec_zfac_create_op": (C.c_int, [_vp, op_fn, _vp, C.c_int64, C.c_int, C.c_int, _vpp]),
    "mispec_zfac_destroy": (C.c_int, [_vp]),
    "mispec_zfac_init": (C.c_int, [_vp, _dp, _lp]),
    "mispec_zfac_factorize": (C.c_int, [_vp, C.c_int, C.c_int, _lp]),
    "mispec_zfac_subspace_dim": (C.c_int, [_vp]),
    "mispec_zfac_f_norm": (C.c_int, [_vp, _dp]),
    "mispec_zfac_get_H": (C.c_int, [_vp, _dp]),
    "mispec_zfac_get_V": (C.c_int, [_vp, C.c_int, _dp]),
    "mispec_zfac_get_f": (C.c_int, [_vp, _dp]),
    "mispec_dense_destroy": (C.c_int, [_vp]),
    "mispec_dense_rows": (C.c_int64, [_vp]),
    "mispec_dense_cols": (C.c_int64, [_vp]),
    "mispec_dense_gemv": (C.c_int, [_vp, _vp, _vp]),
    "mispec_dense_gemv_host": (C.c_int, [_vp, _dp, _dp]),
    "mispec_dense_gemm_host": (C.c_int, [_vp, _dp, C.c_int64, C.c_int, _dp, C.c_int64]),
    "mispec_dense_coeff": (C.c_int, [_vp, C.c_int64, C.c_int64, C.POINTER(C.c_double)]),
    "mispec_dense_gemv_time": (C.c_int, [_vp, _vp, _vp, C.c_int, C.POINTER(C.c_float)]),
    "mispec_fac_create_product": (C.c_int, [_vp, _vp, _vp, C.c_int, _vpp]),
    "mispec_fac_destroy": (C.c_int, [_vp]),
    "mispec_fac_init": (C.c_int, [_vp, _dp, _lp]),
    "mispec_fac_init_random": (C.c_int, [_vp, C.c_uint64, _lp]),
    "mispec_fac_factorize": (C.c_int, [_vp, C.c_int, C.c_int, _lp]),
    "mispec_fac_subspace_dim": (C.c_int, [_vp]),
    "mispec_fac_f_norm": (C.c_int, [_vp, _dp]),
    "mispec_fac_exchange_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "mispec_fac_get_H": (C.c_int, [_vp, _dp]),
    "mispec_fac_set_H": (C.c_int, [_vp, _dp, C.c_int]),
    "mispec_fac_get_V": (C.c_int, [_vp, C.c_int, _dp]),
    "mispec_fac_get_f": (C.c_int, [_vp, _dp]),
    "mispec_fac_V_dev": (_vp, [_vp, _lp]),
    "mispec_fac_local_rows": (C.c_int64, [_vp]),
    "mispec_fac_tridiag_eigen": (C.c_int, [_vp, _dp, _dp]),
    "mispec_fac_ritz_values": (C.c_int, [_vp, _dp, _dp]),
    "mispec_fac_restart_sym": (C.c_int, [_vp, _dp, C.c_int]),
    "mispec_fac_compress_V": (C.c_int, [_vp, _dp, _dp, C.c_int]),
    "mispec_fac_restart_gen": (C.c_int, [_vp, C.POINTER(C.c_int), _dp, _dp, C.c_int, C.c_int]),
    "mispec_fac_ritz_vectors": (C.c_int, [_vp, _dp, C.c_int, _dp, _vpp]),
    "mispec_fac_residuals": (C.c_int, [_vp, _dp, C.c_int, _dp]),
    "mispec_fac_residuals_complex": (C.c_int, [_vp, _dp, _dp, _dp, C.c_int, _dp]),
    "mispec_fac_profile": (C.c_int, [_vp, C.c_int]),
    "mispec_fac_get_profile": (C.c_int, [_vp, C.POINTER(Profile)]),
    "mispec_tridiag_qr": (C.c_int, [_vp, C.c_int, _dp, C.c_double, _dp, _dp]),
    "mispec_tridiag_eigen": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp]),
    "mispec_restart_sweeps": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp]),
    "mispec_symeigs_create": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, _vpp]),
    "mispec_symeigs_create_op": (C.c_int, [_vp, op_fn, _vp, C.c_int64, C.c_int64, C.c_int64, _vpp]),
    "mispec_symeigs_create_shift": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_double, _vpp]),
    "mispec_davidson_create": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_int64, _vpp]),
    "mispec_davidson_create_dense": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_int64, _vpp]),
    "mispec_davidson_create_device_op": (C.c_int, [_vp, device_op_fn, _vp, C.c_int64, _dp, C.c_int64, C.c_int64, C.c_int64, _vpp]),
    "mispec_davidson_destroy": (C.c_int, [_vp]),
    "mispec_davidson_set_sizes": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64]),
    "mispec_davidson_get_sizes": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mispec_davidson_compute": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_double, _dp, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]),
    "mispec_davidson_info": (C.c_int, [_vp]),
    "mispec_davidson_num_iterations": (C.c_int64, [_vp]),
    "mispec_davidson_num_operations": (C.c_int64, [_vp]),
    "mispec_davidson_eigenvalues": (C.c_int, [_vp, _dp]),
    "mispec_davidson_eigenvectors": (C.c_int, [_vp, _dp, C.c_int64]),
    "mispec_symeigs_create_dense": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, _vpp]),
    "mispec_symeigs_create_device_op": (C.c_int, [_vp, device_op_fn, _vp, C.c_int64, C.c_int64, C.c_int64, _vpp]),
    "mispec_symeigs_create_product": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int64, _vpp]),
    "mispec_symeigs_create_geigs_reginv": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int64, _vpp]),
    "mispec_symeigs_create_geigs_cholesky": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int64, _vpp]),
    "mispec_cholesky_create": (C.c_int, [_vp, C.c_int64, _ip, _ip, _dp, C.c_char, C.c_int, _vpp]),
    "mispec_cholesky_destroy": (C.c_int, [_vp]),
    "mispec_cholesky_rows": (C.c_int64, [_vp]),
    "mispec_cholesky_info": (C.c_int, [_vp]),
    "mispec_cholesky_lower_solve_host": (C.c_int, [_vp, _dp, _dp]),
    "mispec_cholesky_upper_solve_host": (C.c_int, [_vp, _dp, _dp]),
    "mispec_fac_create_geigs_cholesky": (C.c_int, [_vp, _vp, _vp, C.c_int, _vpp]),
    "mispec_symeigs_create_geigs_shift": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int64, C.c_int64, C.c_double, _vpp]),
    "mispec_symshift_create_pencil": (C.c_int, [_vp, C.c_int64, _ip, _ip, _dp, C.c_char, C.c_int, _ip, _ip, _dp, C.c_char, C.c_int, _vpp]),
    "mispec_fac_create_geigs_shift": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_double, C.c_int, _vpp]),
    "mispec_reginv_create": (C.c_int, [_vp, C.c_int64, _ip, _ip, _dp, C.c_char, C.c_int, _vpp]),
    "mispec_reginv_destroy": (C.c_int, [_vp]),
    "mispec_reginv_rows": (C.c_int64, [_vp]),
    "mispec_reginv_perform_op_host": (C.c_int, [_vp, _dp, _dp]),
    "mispec_reginv_solve_host": (C.c_int, [_vp, _dp, _dp]),
    "mispec_reginv_last_iterations": (C.c_int64, [_vp]),
    "mispec_fac_create_geigs_reginv": (C.c_int, [_vp, _vp, _vp, C.c_int, _vpp]),
    "mispec_symeigs_destroy": (C.c_int, [_vp]),
    "mispec_symeigs_init": (C.c_int, [_vp, _dp]),
    "mispec_symeigs_compute": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_double, C.c_int, _lp]),
    "mispec_symeigs_info": (C.c_int, [_vp]),
    "mispec_symeigs_num_iterations": (C.c_int64, [_vp]),
    "mispec_symeigs_num_operations": (C.c_int64, [_vp]),
    "mispec_symeigs_eigenvalues": (C.c_int, [_vp, _dp, _lp]),
    "mispec_symeigs_eigenvectors": (C.c_int, [_vp, C.c_int64, _dp, _lp]),
    "mispec_symeigs_residuals": (C.c_int, [_vp, _dp, _lp]),
    "mispec_symeigs_get_profile": (C.c_int, [_vp, C.POINTER(Profile)]),
    "mispec_symeigs_exchange_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "mispec_symeigs_overlap_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mispec_fac_overlap_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mispec_last_ingest_info": (C.c_int, [_dp, C.c_int]),
    "mispec_ingest_threads": (C.c_int, []),
    "mispec_mirror_triangle_host": (C.c_int, [C.c_int64, _ip, _ip, _dp, C.c_char, C.c_int, _ip, _ip, _dp, C.c_int64, _lp]),
    "mispec_symeigs_profile": (C.c_int, [_vp, C.c_int]),
    "mispec_symeigs_set_orth_mode": (C.c_int, [_vp, C.c_int]),
    "mispec_symeigs_orth_info": (C.c_int, [_vp, C.POINTER(C.c_int), _lp, _lp, _lp, _dp, _dp]),
    "mispec_fac_set_orth_mode": (C.c_int, [_vp, C.c_int]),
    "mispec_fac_orth_info": (C.c_int, [_vp, C.POINTER(C.c_int), _lp, _lp, _lp, _dp, _dp]),
    "mispec_fac_restart_info": (C.c_int, [_vp, _lp, _lp]),
    "mispec_fac_turn_info": (C.c_int, [_vp, _lp, _dp, _lp]),
    "mispec_symeigs_turn_info": (C.c_int, [_vp, _lp, _dp, _lp]),
    "mispec_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "mispec_get_option": (C.c_char_p, [C.c_char_p]),
    "mispec_fac_onered_steps": (C.c_int, [_vp, _lp]),
    "mispec_symeigs_onered_steps": (C.c_int, [_vp, _lp]),
    "mispec_symeigs_restart_info": (C.c_int, [_vp, _lp, _lp]),
    "mispec_geneigs_create": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, _vpp]),
    "mispec_geneigs_create_op": (C.c_int, [_vp, op_fn, _vp, C.c_int64, C.c_int64, C.c_int64, _vpp]),
    "mispec_geneigs_create_shift": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_double, _vpp]),
    "mispec_geneigs_create_complex_shift": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_double, C.c_double, _vpp]),
    "mispec_symshift_set_shift_complex": (C.c_int, [_vp, C.c_double, C.c_double]),
    "mispec_geneigs_create_dense": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, _vpp]),
    "mispec_geneigs_create_device_op": (C.c_int, [_vp, device_op_fn, _vp, C.c_int64, C.c_int64, C.c_int64, _vpp]),
    "mispec_symshift_create_general": (C.c_int, [_vp, C.c_int64, _ip, _ip, _dp, C.c_int, _vpp]),
    "mispec_geneigs_destroy": (C.c_int, [_vp]),
    "mispec_geneigs_init": (C.c_int, [_vp, _dp]),
    "mispec_geneigs_compute": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_double, C.c_int, _lp]),
    "mispec_geneigs_info": (C.c_int, [_vp]),
    "mispec_geneigs_num_iterations": (C.c_int64, [_vp]),
    "mispec_geneigs_num_operations": (C.c_int64, [_vp]),
    "mispec_geneigs_eigenvalues": (C.c_int, [_vp, _dp, _lp]),
    "mispec_geneigs_eigenvectors": (C.c_int, [_vp, C.c_int64, _dp, _lp]),
    "mispec_geneigs_residuals": (C.c_int, [_vp, _dp, _lp]),
    "mispec_geneigs_get_profile": (C.c_int, [_vp, C.POINTER(Profile)]),
    "mispec_geneigs_profile": (C.c_int, [_vp, C.c_int]),
    "mispec_hess_qr_host": (C.c_int, [C.c_int, _dp, C.c_double, _dp, _dp]),
    "mispec_hess_qr_lanes_host": (C.c_int, [C.c_int, _dp, C.c_double, _dp, _dp]),
    "mispec_double_shift_qr_lanes_host": (C.c_int, [C.c_int, _dp, C.c_double, C.c_double, _dp, _dp]),
    "mispec_hess_qr": (C.c_int, [_vp, C.c_int, _dp, C.c_double, _dp, _dp]),
    "mispec_double_shift_qr": (C.c_int, [_vp, C.c_int, _dp, C.c_double, C.c_double, _dp, _dp]),
    "mispec_double_shift_qr_host": (C.c_int, [C.c_int, _dp, C.c_double, C.c_double, _dp, _dp]),
    "mispec_hess_schur_host": (C.c_int, [C.c_int, _dp, _dp, _dp]),
    "mispec_hess_eigen_host": (C.c_int, [C.c_int, _dp, _dp, _dp]),
}

_lib = None


def lib():
    """The loaded libmispec.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "spectra_amd/libmispec.so is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  This package has no CPU fallback.")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        X = None
        for name, (res, args) in SIGNATURES.items():
            if name.startswith(EXTRAS_PREFIXES):
                # components outside the hot path: their own library on top of this one (resolved through the handle of the
                # first, so that Python code keeps saying lib().mispec_davidson_create)
                if X is None:
                    if not os.path.exists(EXTRAS_LIB_PATH):
                        raise ImportError("spectra_amd/libmispec_extras.so is missing — build it with `make -C spectra_amd/csrc`")
                    X = C.CDLL(EXTRAS_LIB_PATH, mode=C.RTLD_GLOBAL)
                fn = getattr(X, name)
                setattr(L, name, fn)
            else:
                fn = getattr(L, name)  # AttributeError here = header and library out of sync
            fn.restype = res
            fn.argtypes = args
        L._extras = X
        _lib = L
    return _lib


def check(rc):
    """Map C error classes to the exception types the reference API throws."""
    if rc == MISPEC_OK:
        return
    msg = lib().mispec_last_error().decode(errors="replace")
    if rc == MISPEC_EINVAL:
        raise ValueError(msg)  # std::invalid_argument
    if rc == MISPEC_ELOGIC:
        raise AssertionError(msg)  # std::logic_error
    raise MispecError(msg)
