"""ctypes view of include/opmgpu.h (the C ABI of libopmgpu.so).

The library is the product; this module only loads it and declares signatures.  There is no
CPU fallback: if the HIP extension is missing, `load()` raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libopmgpu.so")

OK, EINVAL, ENODEVICE, ENUMERICAL, ELINSOLVE, EBREAKDOWN, ESINGULAR, ENOMEM, ECOMM = range(9)
HC_GAS_ONLY, HC_GAS_AND_OIL, HC_OIL_ONLY = 0, 1, 2
RELAX_DAMPEN, RELAX_SOR = 0, 1
ORDER_NATURAL, ORDER_MULTICOLOR = 0, 1
K_SPMV, K_ILU_APPLY, K_ILU_FACTOR, K_ASSEMBLE, K_DOT, K_AXPY, K_PROPS, K_STREAM_COPY, K_CPR_APPLY, K_VCYCLE, K_CPR_SETUP, K_SPMV_COLD = range(12)
KT_NAMES = ["cell_props", "flux", "wells", "convergence", "ilu0_factor", "cpr_setup", "spmv_fused_dot1", "spmv_fused_dot2", "ilu0_apply", "amg_vcycle",
            "cpr_other", "vector_updates", "update_state"]
PERF_K = 36
UNIQUE_ID_BYTES = 128

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_int8)


class Grid(C.Structure):
    _fields_ = [("nc", C.c_int32), ("nconn", C.c_int32), ("conn_cells", _ip), ("trans", _dp),
                ("pv", _dp), ("z", _dp), ("gravity", C.c_double), ("thpres", _dp),
                ("pvtnum", _ip), ("satnum", _ip), ("eps", _dp * 8),
                ("scalecrs", C.c_int32), ("eps_v", _dp * 5), ("imbnum", _ip), ("ieps", _dp * 8)]


class Tables(C.Structure):
    _fields_ = [("n_pvt_regions", C.c_int32), ("n_sat_regions", C.c_int32),
                ("has_disgas", C.c_int32), ("has_vapoil", C.c_int32),
                ("surface_density", _dp), ("pvtw", _dp),
                ("oil_node_ptr", _ip), ("oil_rs", _dp), ("oil_psat", _dp), ("oil_invb_sat", _dp),
                ("oil_invbmu_sat", _dp), ("oil_col_ptr", _ip), ("oil_col_p", _dp),
                ("oil_col_invb", _dp), ("oil_col_invbmu", _dp),
                ("gas_node_ptr", _ip), ("gas_pg", _dp), ("gas_rvsat", _dp), ("gas_invb_sat", _dp),
                ("gas_invbmu_sat", _dp), ("gas_col_ptr", _ip), ("gas_col_rv", _dp),
                ("gas_col_invb", _dp), ("gas_col_invbmu", _dp),
                ("swof_ptr", _ip), ("swof_sw", _dp), ("swof_krw", _dp), ("swof_krow", _dp),
                ("swof_pcow", _dp),
                ("sgof_ptr", _ip), ("sgof_sg", _dp), ("sgof_krg", _dp), ("sgof_krog", _dp),
                ("sgof_pcgo", _dp),
                ("rock_pref", C.c_double), ("rock_comp", C.c_double),
                ("vap1", C.c_double), ("vap2", C.c_double),
                ("rocktab_n", C.c_int32), ("rocktab_p", _dp), ("rocktab_pvmult", _dp), ("rocktab_transmult", _dp)]


class WellsSpec(C.Structure):
    _fields_ = [("nw", C.c_int32), ("well_connpos", _ip), ("well_cells", _ip), ("WI", _dp), ("type", _ip), ("allow_cf", _ip),
                ("depth_ref", _dp), ("comp_frac", _dp), ("ctrl_type", _ip), ("ctrl_target", _dp), ("ctrl_distr", _dp),
                ("ctrl_ptr", _ip), ("ctrl_vfp", _ip), ("ctrl_alq", _dp)]


class VfpTable(C.Structure):
    _fields_ = [("id", C.c_int32), ("is_injector", C.c_int32), ("flo_type", C.c_int32), ("wfr_type", C.c_int32), ("gfr_type", C.c_int32),
                ("datum_depth", C.c_double), ("nflo", C.c_int32), ("nthp", C.c_int32), ("nwfr", C.c_int32), ("ngfr", C.c_int32),
                ("nalq", C.c_int32), ("flo", _dp), ("thp", _dp), ("wfr", _dp), ("gfr", _dp), ("alq", _dp), ("data", _dp)]


class Transport(C.Structure):
    """opmgpu_transport: caller-supplied all-reduce / neighbour exchange (include/opmgpu.h)"""
    _fields_ = [("self", C.c_void_p), ("allreduce", C.c_void_p), ("exchange", C.c_void_p), ("destroy", C.c_void_p), ("allreduce_exchange", C.c_void_p)]


class NewtonCtl(C.Structure):
    _fields_ = [("min_iter", C.c_int32), ("use_update_stabilization", C.c_int32), ("relax_type", C.c_int32),
                ("relax_max", C.c_double), ("relax_increment", C.c_double), ("relax_rel_tol", C.c_double)]


class Params(C.Structure):
    _fields_ = [("dp_max_rel", C.c_double), ("ds_max", C.c_double), ("dr_max_rel", C.c_double),
                ("max_residual_allowed", C.c_double), ("tolerance_mb", C.c_double),
                ("tolerance_cnv", C.c_double), ("matbalscale", C.c_double * 3),
                ("linear_solver_reduction", C.c_double), ("linear_solver_maxiter", C.c_int32),
                ("ilu_relaxation", C.c_double), ("ilu_ordering", C.c_int32),
                ("ignore_convergence_failure", C.c_int32), ("use_cpr", C.c_int32),
                ("newton_use_gmres", C.c_int32), ("linear_solver_restart", C.c_int32),
                ("solve_welleq_initially", C.c_int32), ("tolerance_wells", C.c_double), ("tolerance_well_control", C.c_double),
                ("dbhp_max_rel", C.c_double), ("update_equations_scaling", C.c_int32),
                ("gmres_verify_residual", C.c_int32), ("cpr_reference_transform", C.c_int32),
                ("cpr_relax", C.c_double), ("cpr_ilu_n", C.c_int32), ("cpr_use_amg", C.c_int32), ("cpr_use_bicgstab", C.c_int32),
                ("cpr_solver_tol", C.c_double), ("cpr_stage2_relax", C.c_double), ("preconditioner_single", C.c_int32), ("cpr_max_ell_iter", C.c_int32), ("ilu_fillin_level", C.c_int32)]


# use_cpr = 1 with the pressure stage this library was built around: ONE AMG V-cycle per application (cpr_use_amg = 1 without the inner
# Krylov solve the reference's external CPRPreconditioner wraps around it -- cpr_max_ell_iter = 0, a library extension; include/opmgpu.h)
CPR_AMG_VCYCLE = dict(use_cpr=1, cpr_use_amg=1, cpr_max_ell_iter=0)


def default_params(**over):
    """BlackoilModelParameters.cpp:76-102 / FlowLinearSolverParameters defaults."""
    p = Params()
    p.dp_max_rel, p.ds_max, p.dr_max_rel = 0.3, 0.2, 1e9
    p.max_residual_allowed, p.tolerance_mb, p.tolerance_cnv = 1e7, 1e-5, 1e-2
    p.matbalscale[:] = [1.1169, 1.0031, 0.0031]
    p.linear_solver_reduction, p.linear_solver_maxiter = 1e-2, 150
    p.ilu_relaxation, p.ilu_ordering, p.ignore_convergence_failure, p.use_cpr = 0.9, ORDER_MULTICOLOR, 0, 0
    p.newton_use_gmres, p.linear_solver_restart = 0, 40
    p.solve_welleq_initially, p.tolerance_wells, p.tolerance_well_control, p.dbhp_max_rel = 1, 1e-4, 1e-7, 1.0
    p.update_equations_scaling = 0
    p.gmres_verify_residual, p.cpr_reference_transform = 0, 0
    p.cpr_relax, p.cpr_ilu_n, p.cpr_use_amg, p.cpr_use_bicgstab = 1.0, 0, 0, 1            # NewtonIterationBlackoilCPR.hpp:59-63
    p.ilu_fillin_level = 0                                                                                                                   # ISTLSolver.hpp:205
    p.cpr_solver_tol, p.cpr_stage2_relax, p.preconditioner_single, p.cpr_max_ell_iter = 1e-2, 1.0, 0, 25                                       # external CPRPreconditioner (recollection)
    for k, v in over.items():
        if k == "matbalscale":
            p.matbalscale[:] = list(v)
        else:
            setattr(p, k, v)
    if p.use_cpr and "linear_solver_maxiter" not in over:
        p.linear_solver_maxiter = 50          # the reference's CPR plug-in has its own default (NewtonIterationBlackoilCPR.cpp:61-66: maxit 50, restart 40)
    return p


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def iptr(a):
    return None if a is None else a.ctypes.data_as(_ip)


def bptr(a):
    return None if a is None else a.ctypes.data_as(_bp)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


SIGNATURES = {
    # name: (restype, argtypes)
    "opmgpu_default_params": (None, [C.POINTER(Params)]),
    "opmgpu_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Grid), C.POINTER(Tables), C.POINTER(Params)]),
    "opmgpu_destroy": (None, [C.c_void_p]),
    "opmgpu_last_error": (C.c_char_p, [C.c_void_p]),
    "opmgpu_set_wells": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip]),
    "opmgpu_set_state": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _bp]),
    "opmgpu_get_state": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _bp]),
    "opmgpu_assemble": (C.c_int, [C.c_void_p, C.c_double, C.c_int, _dp, _dp, _dp, _dp, _bp]),
    "opmgpu_perf_props": (C.c_int, [C.c_void_p, _dp]),
    "opmgpu_add_well_terms": (C.c_int, [C.c_void_p, _dp, C.c_int, _ip, _dp]),
    "opmgpu_add_well_rhs": (C.c_int, [C.c_void_p, _dp]),
    "opmgpu_perf_dx": (C.c_int, [C.c_void_p, _dp]),
    "opmgpu_convergence": (C.c_int, [C.c_void_p, C.c_double, _dp, _dp, _dp, _dp, C.POINTER(C.c_int)]),
    "opmgpu_solve": (C.c_int, [C.c_void_p, C.c_int, _dp, C.POINTER(C.c_int), _dp]),
    "opmgpu_update_state": (C.c_int, [C.c_void_p, _dp, C.c_double]),
    "opmgpu_stabilize_update": (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    "opmgpu_get_matbalscale": (C.c_int, [C.c_void_p, _dp]),
    "opmgpu_cpr_elliptic_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "opmgpu_cpr_correction_factors": (C.c_int, [C.c_void_p, _dp, _dp]),
    "opmgpu_point_ilu_apply": (C.c_int, [C.c_void_p, _dp, _dp, C.c_double]),
    "opmgpu_comm_set_coarse_blocks": (C.c_int, [C.c_void_p, C.c_int, _ip]),
    "opmgpu_nonlinear_iteration": (C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_int, C.POINTER(NewtonCtl), C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, _dp]),
    "opmgpu_update_hysteresis": (C.c_int, [C.c_void_p]),
    "opmgpu_set_hysteresis": (C.c_int, [C.c_void_p, _dp, _dp]),
    "opmgpu_get_hysteresis": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp]),
    "opmgpu_get_cpr_weights": (C.c_int, [C.c_void_p, _dp]),
    "opmgpu_set_solve_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "opmgpu_set_device_wells": (C.c_int, [C.c_void_p, C.POINTER(WellsSpec)]),
    "opmgpu_well_state_set": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp]),
    "opmgpu_well_state_get": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp]),
    "opmgpu_well_convergence": (C.c_int, [C.c_void_p, _dp, _dp]),
    "opmgpu_set_vfp_tables": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(VfpTable)]),
    "opmgpu_well_controls_set": (C.c_int, [C.c_void_p, _ip, _dp]),
    "opmgpu_well_controls_get": (C.c_int, [C.c_void_p, _ip, _dp, _ip, _ip]),
    "opmgpu_perf_pvt": (C.c_int, [C.c_void_p, _dp, _dp]),
    "opmgpu_average_b": (C.c_int, [C.c_void_p, _dp]),
    "opmgpu_save_state": (C.c_int, [C.c_void_p]),
    "opmgpu_restore_state": (C.c_int, [C.c_void_p]),
    "opmgpu_relative_change": (C.c_int, [C.c_void_p, _dp]),
    "opmgpu_well_controls_set_targets": (C.c_int, [C.c_void_p, _dp, _dp]),
    "opmgpu_region_state_sums": (C.c_int, [C.c_void_p, _ip, C.c_int, _dp]),
    "opmgpu_voidage_coefficients": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _ip, _dp]),
    "opmgpu_compute_fluid_in_place": (C.c_int, [C.c_void_p, _ip, C.c_int, _dp, _dp]),
    "opmgpu_set_sat_oil_max": (C.c_int, [C.c_void_p, _dp]),
    "opmgpu_update_sat_oil_max": (C.c_int, [C.c_void_p]),
    "opmgpu_get_sat_oil_max": (C.c_int, [C.c_void_p, _dp]),
    "opmgpu_create_solver": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Params)]),
    "opmgpu_solve_bsr": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, _dp, _dp, C.c_int, _dp, C.POINTER(C.c_int), _dp]),
    "opmgpu_load_bsr": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, _dp, C.c_int]),
    "opmgpu_spmv": (C.c_int, [C.c_void_p, _dp, _dp]),
    "opmgpu_ilu0_factor": (C.c_int, [C.c_void_p]),
    "opmgpu_ilu0_apply": (C.c_int, [C.c_void_p, _dp, _dp]),
    "opmgpu_ilu0_get": (C.c_int, [C.c_void_p, _dp]),
    "opmgpu_get_ordering": (C.c_int, [C.c_void_p, _ip, _ip, _ip]),
    "opmgpu_get_residual": (C.c_int, [C.c_void_p, _dp]),
    "opmgpu_get_jacobian_nnzb": (C.c_int, [C.c_void_p, _ip]),
    "opmgpu_get_jacobian_bsr": (C.c_int, [C.c_void_p, _ip, _ip, _dp]),
    "opmgpu_time_kernel": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp]),
    "opmgpu_last_timings": (C.c_int, [C.c_void_p, _dp, _dp, _dp]),
    "opmgpu_iteration_marks": (C.c_int, [C.c_void_p, C.c_int]),
    "opmgpu_iteration_marks_get": (C.c_int, [C.c_void_p, C.c_int, _dp, _ip, _ip, _dp, C.POINTER(C.c_int)]),
    "opmgpu_kernel_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "opmgpu_kernel_timing_get": (C.c_int, [C.c_void_p, _dp, C.POINTER(C.c_int64)]),
    "opmgpu_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "opmgpu_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_int32, C.c_int, _ip, _ip, _ip, _ip, _ip]),
    "opmgpu_comm_init_transport": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Transport), C.c_int32, C.c_int, _ip, _ip, _ip, _ip, _ip]),
    "opmgpu_plan_ordering": (C.c_int, [C.c_int, _ip, _ip, C.c_int, _ip, _ip, _ip]),
    "opmgpu_version": (C.c_char_p, []),
    "opmgpu_device_count": (C.c_int, []),
}

_lib = None


def load(path=None):
    """Load libopmgpu.so and attach signatures.  Raises if the HIP extension is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    try:
        # PyTorch wheels bundle their own HIP/HSA/RCCL.  Importing torch FIRST makes libopmgpu.so bind to the same
        # runtime instance, so device pointers, streams and RCCL communicators are shared instead of duplicated.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise RuntimeError(
            "libopmgpu.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % path)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
