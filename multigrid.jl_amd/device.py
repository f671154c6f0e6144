"""ctypes binding of libmgvcycle.so (include/mgvcycle.h) and the device-side hierarchy handle.

This is the Python stand-in for the Julia glue a maintainer would add: it passes
the arrays exactly as Julia's ``SparseMatrixCSC`` holds them - 1-based Int64 ``colptr``/``rowval``,
Float64 ``nzval`` - following the reference's ccall idiom (src/Multigrid/parRelax.jl:61-64).

There is NO CPU fallback: if the HIP library is missing or no GPU is visible, every entry point
raises.

Import order note: PyTorch wheels bundle their own libamdhip64; a process that wants to use torch.cuda as
well must ``import torch`` BEFORE this library is loaded (bench.py and tests/conftest.py do), otherwise the
system HIP runtime this library pulls in shadows torch's and torch.cuda fails to initialise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MGVCYCLE_LIB") or os.path.join(_HERE, "csrc", "libmgvcycle.so")   # override: experiment builds

MG_OP_A, MG_OP_P, MG_OP_R = 0, 1, 2
(MG_K_SPMV, MG_K_RESIDUAL, MG_K_SMOOTH, MG_K_RESTRICT, MG_K_PROLONG, MG_K_DSCALE, MG_K_COARSE,
 MG_K_NORM, MG_K_SMOOTH_PROLONG, MG_K_SMOOTH_RESIDUAL, MG_K_SMOOTH_RESIDUAL_NORM, MG_K_FOUR_STAGE, MG_K_GHOST, MG_K_COUNT) = range(14)
KERNEL_NAMES = ["spmv", "residual", "smooth", "restrict", "prolong", "dscale", "coarse", "norm", "smooth+prolong", "smooth+residual",
                "smooth+residual+norm", "four-stage", "ghost-exchange"]

_ll = C.c_longlong
_dp = C.POINTER(C.c_double)
_lp = C.POINTER(C.c_longlong)
_vp = C.c_void_p

# name -> (restype, argtypes); exactly the symbols include/mgvcycle.h declares
SIGNATURES = {
    "mg_create": (C.c_int, [_ll, _ll, _ll, C.POINTER(_vp)]),
    "mg_set_operator_FP64_INT64": (C.c_int, [_vp, _ll, _ll, _ll, _ll, _lp, _lp, _dp]),
    "mg_set_relax_FP64": (C.c_int, [_vp, _ll, _dp, _ll, _ll, _ll]),
    "mg_set_cycle_type": (C.c_int, [_vp, _ll]),
    "mg_set_relax_type": (C.c_int, [_vp, _ll]),
    "mg_set_grid_hint": (C.c_int, [_vp, _ll, _ll, _ll, _ll]),
    "mg_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "mg_set_coarse_dense_inverse_FP64": (C.c_int, [_vp, _ll, _dp]),
    "mg_graph_launches": (C.c_int, [_vp, _lp, _lp]),
    "mg_set_coarse_lu_FP64_INT64": (C.c_int, [_vp, _ll, _lp, _lp, _dp, _lp, _lp, _dp, _lp, _lp]),
    "mg_set_coarse_gmres_FP64": (C.c_int, [_vp, _ll, _dp]),
    "mg_finalize": (C.c_int, [_vp]),
    "mg_set_nrhs": (C.c_int, [_vp, _ll]),
    "mg_band_form": (C.c_int, [_vp, _ll, C.POINTER(C.c_longlong)]),
    "mg_replace_values_FP64": (C.c_int, [_vp, _ll, _ll, _dp, _ll]),
    "mg_rap_FP64": (C.c_int, [_vp, _dp, _ll, _ll, _dp, _lp]),
    "mg_get_values_FP64": (C.c_int, [_vp, _ll, _ll, _dp, _ll]),
    "mg_get_relax_FP64": (C.c_int, [_vp, _ll, _dp, _ll]),
    "mg_destroy": (C.c_int, [_vp]),
    "mg_cycle_FP64": (C.c_int, [_vp, _dp, _dp, _ll, _ll, _ll]),
    "mg_solve_FP64": (C.c_int, [_vp, _dp, _dp, _ll, _ll, C.c_double, _ll, _lp, _dp]),
    "mg_pcg_FP64": (C.c_int, [_vp, _dp, _dp, _ll, C.c_double, _ll, _lp, _lp, _dp]),
    "mg_pcg_dev_FP64": (C.c_int, [_vp, _vp, _vp, _ll, C.c_double, _ll, _lp, _lp, _dp]),
    "mg_bicgstab_FP64": (C.c_int, [_vp, _dp, _dp, _ll, C.c_double, _ll, _lp, _lp, _dp, _lp]),
    "mg_bicgstab_dev_FP64": (C.c_int, [_vp, _vp, _vp, _ll, C.c_double, _ll, _lp, _lp, _dp, _lp]),
    "mg_fgmres_FP64": (C.c_int, [_vp, _dp, _dp, _ll, _ll, C.c_double, _ll, _lp, _lp, _dp, _lp]),
    "mg_fgmres_dev_FP64": (C.c_int, [_vp, _vp, _vp, _ll, _ll, C.c_double, _ll, _lp, _lp, _dp, _lp]),
    "mg_block_pcg_FP64": (C.c_int, [_vp, _dp, _dp, _ll, _ll, C.c_double, _ll, _lp, _lp, _dp]),
    "mg_block_pcg_dev_FP64": (C.c_int, [_vp, _vp, _vp, _ll, _ll, C.c_double, _ll, _lp, _lp, _dp]),
    "mg_block_bicgstab_FP64": (C.c_int, [_vp, _dp, _dp, _ll, _ll, C.c_double, _ll, _lp, _lp, _dp, _lp]),
    "mg_block_bicgstab_dev_FP64": (C.c_int, [_vp, _vp, _vp, _ll, _ll, C.c_double, _ll, _lp, _lp, _dp, _lp]),
    "mg_block_fgmres_FP64": (C.c_int, [_vp, _dp, _dp, _ll, _ll, _ll, C.c_double, _ll, _lp, _lp, _dp, _lp]),
    "mg_block_fgmres_dev_FP64": (C.c_int, [_vp, _vp, _vp, _ll, _ll, _ll, C.c_double, _ll, _lp, _lp, _dp, _lp]),
    "mg_cycle_mixed_FP32": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), _ll, _ll]),
    "mg_host_register": (C.c_int, [_vp, _ll]),
    "mg_host_unregister": (C.c_int, [_vp]),
    "mg_spmv_FP64": (C.c_int, [_vp, _ll, _ll, C.c_double, _dp, C.c_double, _dp, _ll]),
    "mg_cycle_dev_FP64": (C.c_int, [_vp, _vp, _vp, _ll, _ll, _ll]),
    "mg_solve_dev_FP64": (C.c_int, [_vp, _vp, _vp, _ll, _ll, C.c_double, _ll, _lp, _dp]),
    "mg_spmv_dev_FP64": (C.c_int, [_vp, _ll, _ll, C.c_double, _vp, C.c_double, _vp, _ll]),
    "mg_fused_dev_FP64": (C.c_int, [_vp, _ll, _ll, _vp, _vp, _vp, _ll]),
    "mg_sweep_residual_dev_FP64": (C.c_int, [_vp, _ll, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_double)]),
    "mg_four_stage_dev_FP64": (C.c_int, [_vp, _ll, _vp, _vp, _vp, _vp, C.POINTER(C.c_double)]),
    "mg_four_stage_form": (C.c_int, [_vp, _ll, _lp, _lp]),
    "mg_transpose_hierarchy": (C.c_int, [_vp]),
    "mg_operator_shape": (C.c_int, [_vp, _ll, _ll, _lp]),
    "mg_time_op_dev_FP64": (C.c_int, [_vp, _ll, _ll, _ll, _ll, _dp, _dp]),
    "mg_profile_enable": (C.c_int, [_vp, _ll]),
    "mg_profile_get": (C.c_int, [_vp, _ll, _ll, _dp, _lp, _dp]),
    "mg_profile_reset": (C.c_int, [_vp]),
    "mg_profile_get_moved": (C.c_int, [_vp, _ll, _ll, _dp]),
    "mg_operator_format": (C.c_int, [_vp, _ll, _ll, _lp, _lp, _dp]),
    "mg_operator_rowclasses": (C.c_int, [_vp, _ll, _ll, _lp, _lp, _dp]),
    "mg_operator_rowclass_flags": (C.c_int, [_vp, _ll, _ll, _lp, _lp, _lp, _lp]),
    "mg_sweep_residual_form": (C.c_int, [_vp, _ll, _lp, _lp]),
    "mg_cycle_bytes": (C.c_int, [_vp, _dp]),
    "mg_device_bytes": (C.c_int, [_vp, _dp]),
    "mg_op_create_FP64_INT64": (C.c_int, [_ll, _ll, _ll, _lp, _lp, _dp, C.POINTER(_vp)]),
    "mg_op_create_box_FP64_INT64": (C.c_int, [_ll, _ll, _ll, _lp, _lp, _dp, _ll, _ll, _ll, _ll, C.POINTER(_vp)]),
    "mg_op_create_grid_FP64_INT64": (C.c_int, [_ll, _ll, _ll, _lp, _lp, _dp, _ll, _ll, _ll, _ll, _ll, _ll, _ll,
                                               C.POINTER(_vp)]),
    "mg_op_bind_relax_dev_FP64": (C.c_int, [_vp, _vp, _ll]),
    "mg_op_kernel_variant": (C.c_int, [_vp, _lp, _lp]),
    "mg_op_apply_phase_dev_FP64": (C.c_int, [_vp, _ll, C.c_double, _vp, C.c_double, _vp, _vp, _vp, _ll, _ll, _ll, _vp]),
    "mg_dist_set_level_box": (C.c_int, [_vp, _ll, _ll]),
    "mg_dist_set_relax_type": (C.c_int, [_vp, _ll]),
    "mg_kcycle_step_async_dev_FP64": (C.c_int, [_vp, _vp, _vp, _ll]),
    "mg_op_residual_fused_dev_FP64": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _lp, _vp]),
    "mg_op_can_fuse_next": (C.c_int, [_vp, _vp, _lp]),
    "mg_op_can_sweep_residual": (C.c_int, [_vp, _vp, _vp, _lp, _lp, _lp]),
    "mg_op_sweep_residual_dev_FP64": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _lp, _vp]),
    "mg_op_apply_list_dev_FP64": (C.c_int, [_vp, _ll, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _lp, _vp]),
    "mg_op_destroy": (C.c_int, [_vp]),
    "mg_op_apply_dev_FP64": (C.c_int, [_vp, _ll, C.c_double, _vp, C.c_double, _vp, _vp, _vp, _ll, _vp]),
    "mg_op_apply_rows_dev_FP64": (C.c_int, [_vp, _ll, C.c_double, _vp, C.c_double, _vp, _vp, _vp, _ll, _ll, _vp]),
    "mg_op_info": (C.c_int, [_vp, _lp, _lp, _lp, _dp]),
    "mg_vec_dscale_dev_FP64": (C.c_int, [_vp, _vp, _vp, _ll, _ll, _vp]),
    "mg_vec_xpdr_dev_FP64": (C.c_int, [_vp, _vp, _vp, _vp, _ll, _ll, _vp]),
    "mg_vec_sumsq_dev_FP64": (C.c_int, [_vp, _ll, _vp, _vp, _vp]),
    "mg_set_stream": (C.c_int, [_vp, _vp]),
    "mg_cycle_async_dev_FP64": (C.c_int, [_vp, _vp, _vp, _ll, _ll, _ll]),
    "mg_lu_create_FP64_INT64": (C.c_int, [_ll, _ll, _lp, _lp, _dp, _lp, _lp, _dp, _lp, _lp, C.POINTER(_vp)]),
    "mg_lu_solve_FP64": (C.c_int, [_vp, _dp, _dp, _ll, _ll, _ll]),
    "mg_lu_solve_dev_FP64": (C.c_int, [_vp, _vp, _vp, _ll, _ll, _ll]),
    "mg_lu_destroy": (C.c_int, [_vp]),
    "mg_kaczmarz_create_FP64_INT64": (C.c_int, [_ll, _ll, _lp, _dp, _lp, _ll, _ll, C.POINTER(C.c_uint), _dp, C.POINTER(_vp)]),
    "mg_kaczmarz_apply_FP64": (C.c_int, [_vp, _dp, _dp, _ll, _ll, _ll]),
    "mg_kaczmarz_apply_dev_FP64": (C.c_int, [_vp, _vp, _vp, _ll, _ll, _ll]),
    "mg_kaczmarz_destroy": (C.c_int, [_vp]),
    "mg_dist_unique_id": (C.c_int, [C.c_char_p]),
    "mg_dist_create": (C.c_int, [_ll, _ll, _ll, C.c_char_p, _ll, _ll, _ll, C.POINTER(_vp)]),
    "mg_dist_set_exchange_plugin": (C.c_int, [_vp, _vp, _vp]),
    "mg_dist_set_level": (C.c_int, [_vp, _ll, _ll, _ll, _vp, _vp, _vp, _vp, _vp, _ll, _ll]),
    "mg_dist_set_plan_INT64": (C.c_int, [_vp, _ll, _ll, _ll, _ll, _ll, _lp, _lp, _lp, _ll]),
    "mg_dist_set_tail_INT64": (C.c_int, [_vp, _vp, _ll, _ll, _ll, _lp]),
    "mg_dist_finalize": (C.c_int, [_vp]),
    "mg_dist_cycle_dev_FP64": (C.c_int, [_vp, _vp, _vp, _ll, _ll]),
    "mg_dist_solve_dev_FP64": (C.c_int, [_vp, _vp, _vp, _ll, C.c_double, _ll, _lp, _dp]),
    "mg_dist_set_nrhs": (C.c_int, [_vp, _ll]),
    "mg_dist_release_tail": (C.c_int, [_vp]),
    "mg_dist_comm_count": (C.c_int, [_vp, _lp]),
    "mg_dist_destroy": (C.c_int, [_vp]),
    "mg_ghost_attach": (C.c_int, [_vp, _ll, _ll, _ll, C.c_char_p]),
    "mg_ghost_set_exchange_plugin": (C.c_int, [_vp, _vp, _vp]),
    "mg_ghost_set_side_comm": (C.c_int, [_vp, C.c_char_p]),
    "mg_ghost_set_level_INT64": (C.c_int, [_vp, _ll, _lp, _lp, _lp, _ll, _ll, _lp, _lp, _ll, _lp, _lp]),
    "mg_ghost_finalize": (C.c_int, [_vp]),
    "mg_ghost_set_dry": (C.c_int, [_vp, _ll]),
    "mg_ghost_stats": (C.c_int, [_vp, _lp, _lp]),
    "mg_ghost_comm_count": (C.c_int, [_vp, _lp]),
    "mg_ghost_allreduce_count": (C.c_int, [_vp, _lp]),
    "mg_last_error": (C.c_char_p, []),
    "mg_version": (C.c_char_p, []),
}

_lib = None


class MGDeviceError(RuntimeError):
    pass


def load_library(path: Optional[str] = None):
    """dlopen libmgvcycle.so and bind every symbol of the header.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise MGDeviceError(
            f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the multigrid cycle.")
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export the symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _check(lib, rc: int, what: str):
    if rc != 0:
        msg = lib.mg_last_error()
        raise MGDeviceError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")


def _f64(a):
    return a.ctypes.data_as(_dp)


def _i64(a):
    return a.ctypes.data_as(_lp)


def _julia_arrays(M):
    """scipy CSR -> the (colptr, rowval, nzval) triple Julia holds for the transposed CSC (1-based Int64)."""
    colptr = np.ascontiguousarray(M.indptr, dtype=np.int64) + 1
    rowval = np.ascontiguousarray(M.indices, dtype=np.int64) + 1
    nzval = np.ascontiguousarray(M.data, dtype=np.float64)
    return colptr, rowval, nzval


def _ptr(t) -> int:
    """Device address of a torch tensor / raw int."""
    if isinstance(t, int):
        return t
    if hasattr(t, "data_ptr"):
        if not t.is_cuda:
            raise MGDeviceError("device API called with a CPU tensor")
        if not t.is_contiguous():
            raise MGDeviceError("device API needs contiguous tensors")
        return int(t.data_ptr())
    raise TypeError("expected a torch CUDA tensor or an integer device address")


def _sync_torch(*tensors):
    """The library enqueues on its OWN non-blocking stream, which does not order against torch's: whatever torch still has in
    flight for these tensors (a fill, a copy, the kernel that produced them) must have landed before the library touches them."""
    for t in tensors:
        if t is not None and hasattr(t, "is_cuda") and t.is_cuda:
            import torch
            torch.cuda.current_stream(t.device).synchronize()
            return


DENSE_COARSE_MAX = 16384


class DeviceHierarchy:
    """Owns one ``mg_hierarchy`` handle (HBM copy of As/Ps/Rs/relaxPrecs + coarse inverse)."""

    def __init__(self, param, device_id: int = 0, nrhs: Optional[int] = None, options: Optional[dict] = None):
        """options: per-handle format switches (mg_set_option), e.g. {"no_rowclass": 1} forces the streaming formats."""
        self.lib = load_library()
        self.handle = _vp()
        self.nlevels = len(param.As)
        self.n = int(param.As[0].shape[0])
        self.nrhs = int(nrhs if nrhs is not None else max(1, param.nrhs))
        lib = self.lib
        _check(lib, lib.mg_create(self.nlevels, self.nrhs, int(device_id), C.byref(self.handle)), "mg_create")
        try:
            for key, val in (options or {}).items():
                _check(lib, lib.mg_set_option(self.handle, key.encode(), float(val)), f"mg_set_option({key})")
            self._upload(param)
        except Exception:
            self.close()
            raise

    # -- setup ------------------------------------------------------------------------------------
    def _set_op(self, level, which, M):
        colptr, rowval, nzval = _julia_arrays(M)
        rc = self.lib.mg_set_operator_FP64_INT64(self.handle, level, which, M.shape[0], M.shape[1],
                                                 _i64(colptr), _i64(rowval), _f64(nzval))
        _check(self.lib, rc, f"mg_set_operator(level={level}, which={which})")

    def _upload(self, param):
        lib = self.lib
        nl = self.nlevels
        for l in range(1, nl + 1):
            self._set_op(l, MG_OP_A, param.As[l - 1])
            if l < nl:
                self._set_op(l, MG_OP_P, param.Ps[l - 1])
                self._set_op(l, MG_OP_R, param.Rs[l - 1])
                d = np.ascontiguousarray(param.relaxPrecs[l - 1], dtype=np.float64)
                _check(lib, lib.mg_set_relax_FP64(self.handle, l, _f64(d), d.size,
                                                  int(param.relaxPre(l)), int(param.relaxPost(l))),
                       f"mg_set_relax(level={l})")
        # performance hint only: GMG levels are regular nodal grids (param.Meshes, MGsetup.jl:54)
        for l, mesh in enumerate(getattr(param, "Meshes", []) or []):
            if l < nl and mesh is not None:
                nn = [int(k) + 1 for k in mesh.n] + [1]
                if int(np.prod(nn)) == param.As[l].shape[0]:
                    _check(lib, lib.mg_set_grid_hint(self.handle, l + 1, nn[0], nn[1], nn[2]), "mg_set_grid_hint")
        _check(lib, lib.mg_set_relax_type(self.handle, 1 if param.relaxType == "Jac-GMRES" else 0), "mg_set_relax_type")
        _check(lib, lib.mg_set_cycle_type(self.handle, ord(param.cycleType)), "mg_set_cycle_type")
        if param.LU is None:
            raise MGDeviceError("param.LU is empty: run MGsetup / SA_AMGsetup first")
        self._set_coarse(param)
        _check(lib, lib.mg_finalize(self.handle), "mg_finalize")
        self._schedule = self._schedule_of(param)

    def _schedule_of(self, param):
        nl = self.nlevels
        return (param.cycleType, param.relaxType, tuple(int(param.relaxPre(l)) for l in range(1, nl)),
                tuple(int(param.relaxPost(l)) for l in range(1, nl)))

    def sync_schedule(self, param):
        """Push cycleType / relaxType / sweep counts to the device when they were changed on ``param`` after the upload."""
        sig = self._schedule_of(param)
        if sig == getattr(self, "_schedule", sig):
            self._schedule = sig
            return
        lib = self.lib
        old = self._schedule
        refinalize = False
        if sig[1] != old[1]:
            _check(lib, lib.mg_set_relax_type(self.handle, 1 if param.relaxType == "Jac-GMRES" else 0), "mg_set_relax_type")
            refinalize = True
        if sig[2] != old[2] or sig[3] != old[3]:
            for l in range(1, self.nlevels):
                d = np.ascontiguousarray(param.relaxPrecs[l - 1], dtype=np.float64)
                _check(lib, lib.mg_set_relax_FP64(self.handle, l, _f64(d), d.size, sig[2][l - 1], sig[3][l - 1]),
                       f"mg_set_relax(level={l})")
            refinalize = True
        if sig[0] != old[0]:
            _check(lib, lib.mg_set_cycle_type(self.handle, ord(param.cycleType)), "mg_set_cycle_type")
            refinalize = refinalize or "K" in (sig[0], old[0])
        if refinalize:
            _check(lib, lib.mg_finalize(self.handle), "mg_finalize")
        self._schedule = sig

    def _set_coarse(self, param, force_sparse: bool = False):
        """`z = param.LU\\b` (MGcycle.jl:177) on the device: the explicit inverse for small coarsest levels, the sparse
        L/U factors in the reference's parLU layout (mg_set_coarse_lu_FP64_INT64) above DENSE_COARSE_MAX rows."""
        import scipy.sparse as sp
        lib = self.lib
        nc = int(param.As[-1].shape[0])
        if param.coarseSolveType == "GMRES":                            # param.LU = relaxParam ./ diag(A_c)
            d = np.ascontiguousarray(param.LU, dtype=np.float64)
            _check(lib, lib.mg_set_coarse_gmres_FP64(self.handle, nc, _f64(d)), "mg_set_coarse_gmres")
            return
        if nc <= DENSE_COARSE_MAX and not force_sparse:
            Ainv = np.asfortranarray(param.LU.solve(np.eye(nc)))        # LU \ I, column-major
            _check(lib, lib.mg_set_coarse_dense_inverse_FP64(self.handle, nc, _f64(Ainv)), "mg_set_coarse_dense_inverse")
            return
        lu = param.LU
        L = sp.csr_matrix(lu.L)
        U = sp.csr_matrix(lu.U)
        L.sort_indices()
        U.sort_indices()                                                # lower: diagonal last; upper: diagonal first
        a64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
        Lp, Lc, Lv = a64(L.indptr) + 1, a64(L.indices) + 1, np.ascontiguousarray(L.data, dtype=np.float64)
        Up, Uc, Uv = a64(U.indptr) + 1, a64(U.indices) + 1, np.ascontiguousarray(U.data, dtype=np.float64)
        p = a64(np.argsort(lu.perm_r)) + 1                              # A[p, q] = L U
        q = a64(np.argsort(lu.perm_c)) + 1
        _check(lib, lib.mg_set_coarse_lu_FP64_INT64(self.handle, nc, _i64(Lp), _i64(Lc), _f64(Lv), _i64(Up), _i64(Uc),
                                                    _f64(Uv), _i64(p), _i64(q)), "mg_set_coarse_lu")

    def set_nrhs(self, nrhs: int):
        _check(self.lib, self.lib.mg_set_nrhs(self.handle, int(nrhs)), "mg_set_nrhs")
        self.nrhs = int(nrhs)

    def replace_values(self, level: int, which: int, M):
        nz = np.ascontiguousarray(M.data, dtype=np.float64)
        _check(self.lib, self.lib.mg_replace_values_FP64(self.handle, level, which, _f64(nz), nz.size),
               "mg_replace_values")

    def replace_matrix(self, param, A_new) -> None:
        """replaceMatrixInHierarchy on the device: numeric Galerkin products + relaxPrecs on the fixed patterns
        (mg_rap_FP64), host copies of the hierarchy refreshed from HBM, coarsest level re-factored on the host."""
        import scipy.sparse as sp
        import scipy.sparse.linalg as spla
        lib = self.lib
        nz = np.ascontiguousarray(A_new.data, dtype=np.float64)
        rp = param.relaxParam
        omega = np.ascontiguousarray([float(rp[l]) if isinstance(rp, (list, tuple, np.ndarray)) else float(rp)
                                      for l in range(self.nlevels)], dtype=np.float64)
        kind = 1 if param.relaxType == "SPAI" else 0
        done = C.c_longlong(0)
        _check(lib, lib.mg_rap_FP64(self.handle, _f64(nz), nz.size, kind, _f64(omega), C.byref(done)), "mg_rap")
        param.As[0] = A_new
        for l in range(2, self.nlevels + 1):                       # refresh the host copies (same patterns)
            M = param.As[l - 1]
            vals = np.empty(M.nnz, dtype=np.float64)
            _check(lib, lib.mg_get_values_FP64(self.handle, l, MG_OP_A, _f64(vals), vals.size), "mg_get_values")
            M.data[:] = vals
        for l in range(1, self.nlevels):
            d = np.empty(param.As[l - 1].shape[0], dtype=np.float64)
            _check(lib, lib.mg_get_relax_FP64(self.handle, l, _f64(d), d.size), "mg_get_relax")
            param.relaxPrecs[l - 1] = d
        from .mgsetup import defineCoarsestAinv
        defineCoarsestAinv(param, param.As[-1])                      # (MGsetup.jl:323-355)
        self._set_coarse(param)
        _check(lib, lib.mg_finalize(self.handle), "mg_finalize")

    def close(self):
        if self.handle:
            self.lib.mg_destroy(self.handle)
            self.handle = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host-buffer hot path ------------------------------------------------------------------
    @staticmethod
    def _host_block(a, writable=False):
        if not isinstance(a, np.ndarray) or a.dtype != np.float64:
            raise TypeError("expected a float64 numpy array")
        if a.ndim == 2 and a.shape[1] > 1 and not a.flags.f_contiguous:
            raise ValueError("2-D blocks must be column-major (Julia layout): use np.asfortranarray")
        if a.ndim == 1 and not a.flags.c_contiguous:
            raise ValueError("vectors must be contiguous")
        if writable and not a.flags.writeable:
            raise ValueError("x must be writable (it is updated in place)")
        return a

    def cycle(self, b, x, x_is_zero: int = -1):
        b = self._host_block(b)
        x = self._host_block(x, True)
        nrhs = 1 if b.ndim == 1 else b.shape[1]
        _check(self.lib, self.lib.mg_cycle_FP64(self.handle, _f64(b), _f64(x), b.shape[0], nrhs, int(x_is_zero)),
               "mg_cycle")
        return x

    def solve(self, b, x, tol: float, maxIter: int):
        b = self._host_block(b)
        x = self._host_block(x, True)
        nrhs = 1 if b.ndim == 1 else b.shape[1]
        iters = C.c_longlong(0)
        resvec = np.zeros(int(maxIter) + 1)
        _check(self.lib, self.lib.mg_solve_FP64(self.handle, _f64(b), _f64(x), b.shape[0], nrhs, float(tol),
                                                int(maxIter), C.byref(iters), _f64(resvec)), "mg_solve")
        return x, int(iters.value), resvec[: iters.value + 1]

    def pcg(self, b, x, tol: float, maxIter: int):
        """KrylovMethods.cg / blockCG with the MG cycle as preconditioner; returns (x, flag, iters, resvec)
        (resvec: ||r||/||b|| per iteration; for a block the maximum over the columns)."""
        b = self._host_block(b)
        x = self._host_block(x, True)
        if b.ndim != 1 and b.shape[1] > 1:
            iters, flag = C.c_longlong(0), C.c_longlong(0)
            resmat = np.zeros((max(int(maxIter), 1), b.shape[1]))
            _check(self.lib, self.lib.mg_block_pcg_FP64(self.handle, _f64(b), _f64(x), b.shape[0], b.shape[1], float(tol),
                                                        int(maxIter), C.byref(iters), C.byref(flag), _f64(resmat)), "mg_block_pcg")
            self.last_resmat = resmat[: iters.value]
            return x, int(flag.value), int(iters.value), resmat[: iters.value].max(axis=1) if iters.value else np.zeros(0)
        if b.ndim != 1:
            b, x = b[:, 0], x[:, 0]
        iters, flag = C.c_longlong(0), C.c_longlong(0)
        resvec = np.zeros(max(int(maxIter), 1))
        _check(self.lib, self.lib.mg_pcg_FP64(self.handle, _f64(b), _f64(x), b.shape[0], float(tol), int(maxIter),
                                              C.byref(iters), C.byref(flag), _f64(resvec)), "mg_pcg")
        return x, int(flag.value), int(iters.value), resvec[: iters.value]

    def bicgstab(self, b, x, tol: float, maxIter: int):
        """KrylovMethods.bicgstb with the MG cycle as M1; returns (x, flag, iters, resvec)."""
        b = self._host_block(b)
        x = self._host_block(x, True)
        iters, flag, nres = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        resvec = np.zeros(2 * int(maxIter) + 1)
        if b.ndim != 1 and b.shape[1] > 1:
            _check(self.lib, self.lib.mg_block_bicgstab_FP64(self.handle, _f64(b), _f64(x), b.shape[0], b.shape[1], float(tol),
                                                             int(maxIter), C.byref(iters), C.byref(flag), _f64(resvec),
                                                             C.byref(nres)), "mg_block_bicgstab")
            return x, int(flag.value), int(iters.value), resvec[: nres.value]
        if b.ndim != 1:
            b, x = b[:, 0], x[:, 0]
        _check(self.lib, self.lib.mg_bicgstab_FP64(self.handle, _f64(b), _f64(x), b.shape[0], float(tol), int(maxIter),
                                                   C.byref(iters), C.byref(flag), _f64(resvec), C.byref(nres)), "mg_bicgstab")
        return x, int(flag.value), int(iters.value), resvec[: nres.value]

    def fgmres(self, b, x, inner: int, tol: float, maxIter: int):
        """KrylovMethods.fgmres (flexible) with the MG cycle as preconditioner; returns (x, flag, iters, resvec)."""
        b = self._host_block(b)
        x = self._host_block(x, True)
        iters, flag, nres = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        resvec = np.zeros(max(1, int(inner) * int(maxIter)))
        if b.ndim != 1 and b.shape[1] > 1:
            _check(self.lib, self.lib.mg_block_fgmres_FP64(self.handle, _f64(b), _f64(x), b.shape[0], b.shape[1], int(inner),
                                                           float(tol), int(maxIter), C.byref(iters), C.byref(flag), _f64(resvec),
                                                           C.byref(nres)), "mg_block_fgmres")
            return x, int(flag.value), int(iters.value), resvec[: nres.value]
        if b.ndim != 1:
            b, x = b[:, 0], x[:, 0]
        _check(self.lib, self.lib.mg_fgmres_FP64(self.handle, _f64(b), _f64(x), b.shape[0], int(inner), float(tol),
                                                 int(maxIter), C.byref(iters), C.byref(flag), _f64(resvec), C.byref(nres)),
               "mg_fgmres")
        return x, int(flag.value), int(iters.value), resvec[: nres.value]

    def cycle_mixed_f32(self, b32, z32):
        """getMultigridPreconditioner's mixed-precision branch (SolveFuncs.jl:52-58): Float32 block, Float64 hierarchy."""
        if b32.dtype != np.float32 or z32.dtype != np.float32:
            raise TypeError("expected float32 arrays")
        nrhs = 1 if b32.ndim == 1 else b32.shape[1]
        if b32.ndim == 2 and nrhs > 1 and not (b32.flags.f_contiguous and z32.flags.f_contiguous):
            raise ValueError("2-D blocks must be column-major (Julia layout)")
        fp = C.POINTER(C.c_float)
        _check(self.lib, self.lib.mg_cycle_mixed_FP32(self.handle, b32.ctypes.data_as(fp), z32.ctypes.data_as(fp), b32.shape[0], nrhs),
               "mg_cycle_mixed_FP32")
        return z32

    def pcg_dev(self, b, x, tol: float, maxIter: int):
        _sync_torch(b, x)
        iters, flag = C.c_longlong(0), C.c_longlong(0)
        resvec = np.zeros(max(int(maxIter), 1))
        _check(self.lib, self.lib.mg_pcg_dev_FP64(self.handle, _ptr(b), _ptr(x), self.n, float(tol), int(maxIter),
                                                  C.byref(iters), C.byref(flag), _f64(resvec)), "mg_pcg_dev")
        return int(flag.value), int(iters.value), resvec[: iters.value]

    # -- block Krylov drivers on device tensors (row-major [n][nrhs]); KrylovMethods.blockCG / blockBiCGSTB / blockFGMRES ----------
    def block_pcg_dev(self, b, x, tol: float, maxIter: int):
        """returns (flag, iterations, resmat[iterations][nrhs] of ||r_j|| / ||b_j||)"""
        _sync_torch(b, x)
        iters, flag = C.c_longlong(0), C.c_longlong(0)
        resmat = np.zeros((max(int(maxIter), 1), self.nrhs))
        _check(self.lib, self.lib.mg_block_pcg_dev_FP64(self.handle, _ptr(b), _ptr(x), self.n, self.nrhs, float(tol), int(maxIter),
                                                        C.byref(iters), C.byref(flag), _f64(resmat)), "mg_block_pcg_dev")
        return int(flag.value), int(iters.value), resmat[: iters.value]

    def block_bicgstab_dev(self, b, x, tol: float, maxIter: int):
        _sync_torch(b, x)
        iters, flag, nres = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        resvec = np.zeros(2 * max(int(maxIter), 1) + 1)
        _check(self.lib, self.lib.mg_block_bicgstab_dev_FP64(self.handle, _ptr(b), _ptr(x), self.n, self.nrhs, float(tol), int(maxIter),
                                                             C.byref(iters), C.byref(flag), _f64(resvec), C.byref(nres)), "mg_block_bicgstab_dev")
        return int(flag.value), int(iters.value), resvec[: nres.value]

    def block_fgmres_dev(self, b, x, inner: int, tol: float, maxIter: int):
        _sync_torch(b, x)
        iters, flag, nres = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        resvec = np.zeros(max(int(inner) * int(maxIter), 1))
        _check(self.lib, self.lib.mg_block_fgmres_dev_FP64(self.handle, _ptr(b), _ptr(x), self.n, self.nrhs, int(inner), float(tol), int(maxIter),
                                                           C.byref(iters), C.byref(flag), _f64(resvec), C.byref(nres)), "mg_block_fgmres_dev")
        return int(flag.value), int(iters.value), resvec[: nres.value]

    def bicgstab_dev(self, b, x, tol: float, maxIter: int):
        """solveBiCGSTAB_MG on device tensors (one right-hand side); returns (flag, iterations, resvec: the entry at the start, then
        two per iteration)."""
        _sync_torch(b, x)
        iters, flag, nres = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        resvec = np.zeros(2 * max(int(maxIter), 1) + 1)
        _check(self.lib, self.lib.mg_bicgstab_dev_FP64(self.handle, _ptr(b), _ptr(x), self.n, float(tol), int(maxIter),
                                                       C.byref(iters), C.byref(flag), _f64(resvec), C.byref(nres)), "mg_bicgstab_dev")
        return int(flag.value), int(iters.value), resvec[: nres.value]

    def fgmres_dev(self, b, x, inner: int, tol: float, maxIter: int):
        """solveGMRES_MG on device tensors (one right-hand side); returns (flag, inner steps, resvec)."""
        _sync_torch(b, x)
        iters, flag, nres = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        resvec = np.zeros(max(int(inner) * int(maxIter), 1))
        _check(self.lib, self.lib.mg_fgmres_dev_FP64(self.handle, _ptr(b), _ptr(x), self.n, int(inner), float(tol), int(maxIter),
                                                     C.byref(iters), C.byref(flag), _f64(resvec), C.byref(nres)), "mg_fgmres_dev")
        return int(flag.value), int(iters.value), resvec[: nres.value]

    def spmv(self, level: int, which: int, alpha: float, x, beta: float, y):
        x = self._host_block(x)
        y = self._host_block(y, True)
        nrhs = 1 if x.ndim == 1 else x.shape[1]
        _check(self.lib, self.lib.mg_spmv_FP64(self.handle, level, which, float(alpha), _f64(x), float(beta),
                                               _f64(y), nrhs), "mg_spmv")
        return y

    # -- device-resident hot path (torch CUDA tensors, row-major [n][nrhs]) ---------------------
    def cycle_dev(self, b, x, x_is_zero: int = -1, nrhs: Optional[int] = None):
        _sync_torch(b, x)
        nrhs = self.nrhs if nrhs is None else nrhs
        _check(self.lib, self.lib.mg_cycle_dev_FP64(self.handle, _ptr(b), _ptr(x), self.n, nrhs, int(x_is_zero)),
               "mg_cycle_dev")

    def solve_dev(self, b, x, tol: float, maxIter: int, nrhs: Optional[int] = None):
        _sync_torch(b, x)
        nrhs = self.nrhs if nrhs is None else nrhs
        iters = C.c_longlong(0)
        resvec = np.zeros(int(maxIter) + 1)
        _check(self.lib, self.lib.mg_solve_dev_FP64(self.handle, _ptr(b), _ptr(x), self.n, nrhs, float(tol),
                                                    int(maxIter), C.byref(iters), _f64(resvec)), "mg_solve_dev")
        return int(iters.value), resvec[: iters.value + 1]

    def spmv_dev(self, level, which, alpha, x, beta, y, nrhs: Optional[int] = None):
        _sync_torch(x, y)
        nrhs = self.nrhs if nrhs is None else nrhs
        _check(self.lib, self.lib.mg_spmv_dev_FP64(self.handle, level, which, float(alpha), _ptr(x), float(beta),
                                                   _ptr(y), nrhs), "mg_spmv_dev")

    def fused_dev(self, level, kernel, b, x, out, nrhs: Optional[int] = None):
        _sync_torch(b, x, out)
        nrhs = self.nrhs if nrhs is None else nrhs
        _check(self.lib, self.lib.mg_fused_dev_FP64(self.handle, level, kernel, _ptr(b), _ptr(x), _ptr(out), nrhs),
               "mg_fused_dev")

    def sweep_residual_dev(self, level, b, x, t, r=None, xn=None, want_norm=False):
        """t = x + d.*(b - A x), r = b - A t [, xn = t + d.*r, ||r||] in one pass (two-stage marching kernel)."""
        _sync_torch(b, x, t, r, xn)
        ss = C.c_double(0.0)
        _check(self.lib, self.lib.mg_sweep_residual_dev_FP64(
            self.handle, level, _ptr(b), _ptr(x), _ptr(t), _ptr(r) if r is not None else None,
            _ptr(xn) if xn is not None else None, C.byref(ss) if want_norm else None), "mg_sweep_residual_dev")
        return float(ss.value)

    def four_stage_dev(self, level, b, x, tp, rp, want_norm=True):
        """The solve loop's two fine-level passes across the stopping test as one pass: t = x + d.*(b - A x), r = b - A t (||r||
        returned), xn = t + d.*r, tp = xn + d.*(b - A xn), rp = b - A tp (csr_rowclass_march4_spmv)."""
        _sync_torch(b, x, tp, rp)
        ss = C.c_double(0.0)
        _check(self.lib, self.lib.mg_four_stage_dev_FP64(self.handle, level, _ptr(b), _ptr(x), _ptr(tp), _ptr(rp),
                                                         C.byref(ss) if want_norm else None), "mg_four_stage_dev")
        return float(ss.value)

    def get_values(self, level: int, which: int) -> np.ndarray:
        """nzval of operator `which` of `level` as the device holds it (stored CSR order)."""
        nnz = C.c_longlong(0)
        info = (C.c_longlong * 3)()
        _check(self.lib, self.lib.mg_operator_shape(self.handle, level, which, info), "mg_operator_shape")
        vals = np.zeros(int(info[2]), dtype=np.float64)
        _check(self.lib, self.lib.mg_get_values_FP64(self.handle, level, which, _f64(vals), vals.size), "mg_get_values")
        return vals

    def transpose_hierarchy(self):
        """transposeHierarchy (MGsetup.jl:274-318) on the resident hierarchy; raises MGDeviceError (status MG_ERR_UNSUPPORTED)
        when the library cannot (sparse coarsest factors): the caller then re-uploads."""
        _check(self.lib, self.lib.mg_transpose_hierarchy(self.handle), "mg_transpose_hierarchy")

    def four_stage_form(self, level: int):
        """(available, geometry as sweep_residual_form's)."""
        f = C.c_longlong(0)
        g = (C.c_longlong * 12)()
        _check(self.lib, self.lib.mg_four_stage_form(self.handle, level, C.byref(f), g), "mg_four_stage_form")
        return bool(f.value), [int(v) for v in g]

    def set_stream(self, stream: int):
        """Enqueue on the caller's HIP stream (e.g. ``torch.cuda.current_stream().cuda_stream``)."""
        _check(self.lib, self.lib.mg_set_stream(self.handle, _vp(stream)), "mg_set_stream")

    def cycle_async_dev(self, b, x, x_is_zero: int, nrhs: Optional[int] = None):
        nrhs = self.nrhs if nrhs is None else nrhs
        _check(self.lib, self.lib.mg_cycle_async_dev_FP64(self.handle, _ptr(b), _ptr(x), self.n, nrhs, int(x_is_zero)),
               "mg_cycle_async_dev")

    # -- measurement --------------------------------------------------------------------------------
    def time_op(self, level: int, kernel: int, reps: int = 20):
        ms = C.c_double(0.0)
        bts = C.c_double(0.0)
        _check(self.lib, self.lib.mg_time_op_dev_FP64(self.handle, level, kernel, self.nrhs, int(reps),
                                                      C.byref(ms), C.byref(bts)), "mg_time_op_dev")
        return ms.value, bts.value

    def profile_enable(self, on: bool):
        _check(self.lib, self.lib.mg_profile_enable(self.handle, 1 if on else 0), "mg_profile_enable")

    def profile_reset(self):
        _check(self.lib, self.lib.mg_profile_reset(self.handle), "mg_profile_reset")

    def profile(self):
        """{(level, kernel_name): (total_ms, launches, bytes_per_launch)} for every kernel that ran."""
        out = {}
        for l in range(1, self.nlevels + 1):
            for k in range(MG_K_COUNT):
                ms, n, bts = C.c_double(0), C.c_longlong(0), C.c_double(0)
                _check(self.lib, self.lib.mg_profile_get(self.handle, l, k, C.byref(ms), C.byref(n), C.byref(bts)),
                       "mg_profile_get")
                if n.value:
                    out[(l, KERNEL_NAMES[k])] = (ms.value, int(n.value), bts.value)
        return out

    def profile_moved(self):
        """{(level, kernel_name): bytes one launch of the kernel in use has to move} (device format + each vector once)."""
        out = {}
        for (l, name) in self.profile():
            mv = C.c_double(0)
            _check(self.lib, self.lib.mg_profile_get_moved(self.handle, l, KERNEL_NAMES.index(name), C.byref(mv)),
                   "mg_profile_get_moved")
            out[(l, name)] = mv.value
        return out

    def operator_format(self, level: int, which: int):
        """(number of row patterns [0 = plain CSR], dictionary entries, index-side bytes per nrhs=1 launch)."""
        npat, nd, ib = C.c_longlong(0), C.c_longlong(0), C.c_double(0)
        _check(self.lib, self.lib.mg_operator_format(self.handle, level, which, C.byref(npat), C.byref(nd), C.byref(ib)),
               "mg_operator_format")
        return int(npat.value), int(nd.value), ib.value

    def operator_rowclasses(self, level: int, which: int):
        """(number of row classes [0 = not stored that way], dictionary entries, matrix-side bytes one nrhs=1 launch of
        the kernel in use streams)."""
        nc, nd, mb = C.c_longlong(0), C.c_longlong(0), C.c_double(0)
        _check(self.lib, self.lib.mg_operator_rowclasses(self.handle, level, which, C.byref(nc), C.byref(nd), C.byref(mb)),
               "mg_operator_rowclasses")
        return int(nc.value), int(nd.value), mb.value

    def operator_rowclass_flags(self, level: int, which: int):
        """(implicit first column, relaxPrec read from the class dictionary) of a row-class operator."""
        a, b, c, e = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        _check(self.lib, self.lib.mg_operator_rowclass_flags(self.handle, level, which, C.byref(a), C.byref(b), C.byref(c),
                                                             C.byref(e)), "mg_operator_rowclass_flags")
        return bool(a.value), bool(b.value)

    def sweep_residual_form(self, level: int):
        """(form, geometry): 0 two launches, 2 csr_rowclass_march2_spmv, 5 csr_rowclass_march27_spmv (27-point levels; geometry of
        the pair, [7] = workgroups of its single-product geometry), 3 (4: band form) csr_rowclass_march3_spmv with its tile geometry
        [tiles per line, tiles per column, TX, TY, rows per lane, workgroups, LDS bytes, est. fill bytes per row x 100,
        threads per workgroup, lockstep segments (0: balanced ranges), planes per segment, class-table entries]."""
        f = C.c_longlong(0)
        g = (C.c_longlong * 12)()
        _check(self.lib, self.lib.mg_sweep_residual_form(self.handle, level, C.byref(f), g), "mg_sweep_residual_form")
        return int(f.value), [int(v) for v in g]

    def band_form(self, level: int):
        """[held, canonical slots, symmetric reads, value planes streamed per pass] of the level's band form (mg_band_form)."""
        g = (C.c_longlong * 4)()
        _check(self.lib, self.lib.mg_band_form(self.handle, level, g), "mg_band_form")
        return [int(v) for v in g]

    def operator_kernel_variant(self, level: int, which: int) -> int:
        """-1 streaming formats, 0 csr_rowclass_spmv, 1 csr_rowclass_window_spmv, 2 csr_rowclass_tile_spmv, 3 csr_rowclass_march_spmv,
        4 csr_rowclass_lane_spmv, 7 csr_rowclass_marchr_spmv (marching restriction), 8 the small-grid-level kernels (mg_small.hpp)."""
        return self.operator_kernel_info(level, which)[0]

    def operator_kernel_info(self, level: int, which: int):
        """(kernel variant as above, number of exception rows)."""
        a, b, c, e = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        _check(self.lib, self.lib.mg_operator_rowclass_flags(self.handle, level, which, C.byref(a), C.byref(b), C.byref(c),
                                                             C.byref(e)), "mg_operator_rowclass_flags")
        return int(c.value), int(e.value)

    def graph_launches(self):
        """(replays so far, graphs cached) of the HIP graphs the launch-bound coarse sub-cycles run as."""
        a, b = C.c_longlong(0), C.c_longlong(0)
        _check(self.lib, self.lib.mg_graph_launches(self.handle, C.byref(a), C.byref(b)), "mg_graph_launches")
        return int(a.value), int(b.value)

    def cycle_bytes(self) -> float:
        v = C.c_double(0)
        _check(self.lib, self.lib.mg_cycle_bytes(self.handle, C.byref(v)), "mg_cycle_bytes")
        return v.value

    def device_bytes(self) -> float:
        v = C.c_double(0)
        _check(self.lib, self.lib.mg_device_bytes(self.handle, C.byref(v)), "mg_device_bytes")
        return v.value


class DeviceOperator:
    """One CSR operator resident in HBM on its own (``mg_operator``): the building block of the multi-GPU
    cycle, where a rank holds its rows of A/P/R with halo columns appended.  Asynchronous on `stream`."""

    def __init__(self, M, device_id: int = 0, box=None, regular_cols=None, coarse_box=None):
        """box=(n1,n2,n3), regular_cols: the BOX form of a sharded level's local A (mg_op_create_box_FP64_INT64): M is
        square [owned box in natural order | halo] with empty halo rows.  With coarse_box=(c1,c2,c3) as well: the grid form
        of a local P (mg_op_create_grid_FP64_INT64): rows = the owned fine box `box`, columns = [owned coarse box | halo]."""
        self.lib = load_library()
        self.handle = _vp()
        colptr, rowval, nzval = _julia_arrays(M)
        if M.nnz == 0:                                   # keep the arrays non-empty for ctypes
            rowval = np.zeros(1, dtype=np.int64)
            nzval = np.zeros(1)
        self.shape = M.shape
        self.nnz = int(M.nnz)
        self.box = box is not None
        if regular_cols is not None and box is None and coarse_box is None:
            # only the owned | halo column split (a local restriction): rows reading a halo column go to phase 2
            _check(self.lib, self.lib.mg_op_create_grid_FP64_INT64(int(device_id), M.shape[0], M.shape[1], _i64(colptr),
                                                                   _i64(rowval), _f64(nzval), int(regular_cols),
                                                                   0, 0, 0, 0, 0, 0, C.byref(self.handle)), "mg_op_create_grid")
        elif box is not None and coarse_box is not None:
            f = (list(box) + [1, 1])[:3]
            c = (list(coarse_box) + [1, 1])[:3]
            _check(self.lib, self.lib.mg_op_create_grid_FP64_INT64(int(device_id), M.shape[0], M.shape[1], _i64(colptr),
                                                                   _i64(rowval), _f64(nzval), int(regular_cols),
                                                                   int(f[0]), int(f[1]), int(f[2]), int(c[0]), int(c[1]),
                                                                   int(c[2]), C.byref(self.handle)), "mg_op_create_grid")
        elif box is not None:
            n1, n2, n3 = (list(box) + [1, 1])[:3]
            _check(self.lib, self.lib.mg_op_create_box_FP64_INT64(int(device_id), M.shape[0], M.shape[1], _i64(colptr),
                                                                  _i64(rowval), _f64(nzval), int(n1), int(n2), int(n3),
                                                                  int(regular_cols), C.byref(self.handle)), "mg_op_create_box")
        else:
            _check(self.lib, self.lib.mg_op_create_FP64_INT64(int(device_id), M.shape[0], M.shape[1], _i64(colptr),
                                                              _i64(rowval), _f64(nzval), C.byref(self.handle)),
                   "mg_op_create")

    def bind_relax(self, d, n: int):
        """The relaxPrec vector (CUDA tensor) this operator is swept with: read from the class dictionary where possible."""
        _check(self.lib, self.lib.mg_op_bind_relax_dev_FP64(self.handle, _ptr(d), int(n)), "mg_op_bind_relax_dev")

    def kernel_variant(self):
        """(kernel variant as DeviceHierarchy.operator_kernel_variant, number of exception rows)."""
        a, b = C.c_longlong(0), C.c_longlong(0)
        _check(self.lib, self.lib.mg_op_kernel_variant(self.handle, C.byref(a), C.byref(b)), "mg_op_kernel_variant")
        return int(a.value), int(b.value)

    def apply(self, kernel, x, y, b=None, d=None, alpha=1.0, beta=0.0, nrhs=1, stream=0, row_offset=0, phase=0):
        _check(self.lib, self.lib.mg_op_apply_phase_dev_FP64(self.handle, int(kernel), float(alpha), _ptr(x), float(beta),
                                                             _ptr(y), _ptr(b) if b is not None else None,
                                                             _ptr(d) if d is not None else None, int(nrhs),
                                                             int(row_offset), int(phase), _vp(stream)),
               "mg_op_apply_phase_dev")

    def can_sweep_residual(self, x, d):
        """(yes, rows of list 1, rows of list 2): can the two-stage pass serve this operator with these vectors, and how
        many rows does it leave to ``apply_list``?"""
        y, a, b = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        _check(self.lib, self.lib.mg_op_can_sweep_residual(self.handle, _ptr(x), _ptr(d), C.byref(y), C.byref(a), C.byref(b)),
               "mg_op_can_sweep_residual")
        return bool(y.value), int(a.value), int(b.value)

    def sweep_residual(self, x, b, d, t=None, r=None, xn=None, partials=None, stream=0):
        """t = x + d.*(b - M x) and r = b - M t [xn = t + d.*r, ||r||^2 partials] in one pass; returns the number of partials."""
        n = C.c_longlong(0)
        opt = lambda v: _ptr(v) if v is not None else None
        _check(self.lib, self.lib.mg_op_sweep_residual_dev_FP64(self.handle, _ptr(x), _ptr(b), _ptr(d), opt(t), opt(r), opt(xn),
                                                                opt(partials), C.byref(n), _vp(stream)), "mg_op_sweep_residual_dev")
        return int(n.value)

    def apply_list(self, which, kernel, x, y, b, d=None, y2=None, partials=None, stream=0):
        """The rows the two-stage pass leaves out (list 1: rows without a class; list 2: rows next to them), from the CSR arrays."""
        n = C.c_longlong(0)
        opt = lambda v: _ptr(v) if v is not None else None
        _check(self.lib, self.lib.mg_op_apply_list_dev_FP64(self.handle, int(which), int(kernel), _ptr(x), opt(y), _ptr(b), opt(d),
                                                            opt(y2), opt(partials), C.byref(n), _vp(stream)), "mg_op_apply_list_dev")
        return int(n.value)

    def close(self):
        if self.handle:
            self.lib.mg_op_destroy(self.handle)
            self.handle = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_register(a: np.ndarray):
    """Page-lock a long-lived numpy array (mg_host_register); the caller unregisters it before it is freed."""
    lib = load_library()
    _check(lib, lib.mg_host_register(_vp(a.ctypes.data), a.nbytes), "mg_host_register")


def host_unregister(a: np.ndarray):
    lib = load_library()
    _check(lib, lib.mg_host_unregister(_vp(a.ctypes.data)), "mg_host_unregister")


def vec_dscale(d, b, x, n, nrhs=1, stream=0):
    lib = load_library()
    _check(lib, lib.mg_vec_dscale_dev_FP64(_ptr(d), _ptr(b), _ptr(x), int(n), int(nrhs), _vp(stream)), "mg_vec_dscale_dev")


def vec_xpdr(x, d, r, xout, n, nrhs=1, stream=0):
    lib = load_library()
    _check(lib, lib.mg_vec_xpdr_dev_FP64(_ptr(x), _ptr(d), _ptr(r), _ptr(xout), int(n), int(nrhs), _vp(stream)),
           "mg_vec_xpdr_dev")


def vec_sumsq(x, length, workspace, out, stream=0):
    lib = load_library()
    _check(lib, lib.mg_vec_sumsq_dev_FP64(_ptr(x), int(length), _ptr(workspace), _ptr(out), _vp(stream)),
           "mg_vec_sumsq_dev")
