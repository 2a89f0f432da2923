"""ctypes binding of the product library ``unconfined_amd/libucf.so`` (the C ABI of
include/ucf.h: HIP kernels for gfx950 + host-side plan builder).

There is deliberately no fallback: if the shared library is missing or no HIP
device is usable, every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .abi import UcfDerived, UcfParams, UcfStats

HERE = os.path.dirname(os.path.abspath(__file__))
# UCF_LIB_PATH: an experimental build of the SAME library (tools/ubench/build_variant.sh) for A/B timing; the product
# path is unconfined_amd/libucf.so and nothing else is ever loaded without that variable
LIB_PATH = os.environ.get("UCF_LIB_PATH") or os.path.join(HERE, "libucf.so")

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")

EXPORTS = [
    "ucf_version", "ucf_last_error", "ucf_status_string",
    "ucf_plan_create", "ucf_plan_create_on", "ucf_device_count", "ucf_plan_destroy", "ucf_plan_update", "ucf_plan_derived", "ucf_nondimensionalise", "ucf_plan_j0z", "ucf_plan_tanh_sinh",
    "ucf_plan_gauss_lobatto", "ucf_plan_set_mode", "ucf_plan_set_timing", "ucf_plan_kernel_ms", "ucf_plan_kernel_times",
    "ucf_plan_reserve", "ucf_plan_alloc_count", "ucf_build_id",
    "ucf_shard_rows", "ucf_drawdown_grid_shard_device", "ucf_drawdown_grid_multi", "ucf_drawdown_batch_multi",
    "ucf_drawdown_grid_allgather", "ucf_comm_unique_id", "ucf_comm_create", "ucf_comm_destroy",
    "ucf_logspace", "ucf_linspace", "ucf_zlay", "ucf_split_vector",
    "ucf_drawdown_batch", "ucf_drawdown_batch_device", "ucf_drawdown_grid", "ucf_drawdown_grid_device",
    "ucf_drawdown_multi", "ucf_screen_average",
    "ucf_eval_samples", "ucf_pvalues", "ucf_dehoog", "ucf_wynn_epsilon", "ucf_extraptozero", "ucf_bessel_k01",
    "ucf_debug_stages", "ucf_debug_wynn", "ucf_debug_dehoog_tiles",
    "ucf_fp64_fma_peak", "ucf_sincos_table", "ucf_exp2_table",
]


class UcfError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"ucf status {status}: {message}")
        self.status = status
        self.message = message


def build(verbose: bool = False) -> str:
    """compile the HIP library in-tree (hipcc, gfx950); returns the .so path"""
    res = subprocess.run(["make", "-C", os.path.join(HERE, "csrc"), "-j6"], capture_output=True, text=True)
    if verbose or res.returncode:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode:
        raise RuntimeError("building libucf.so failed")
    return LIB_PATH


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the drawdown path)")
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.ucf_version.restype = C.c_int
    lib.ucf_last_error.restype = C.c_char_p
    lib.ucf_status_string.restype = C.c_char_p
    lib.ucf_status_string.argtypes = [C.c_int]
    lib.ucf_plan_create.argtypes = [C.POINTER(UcfParams), C.POINTER(vp)]
    lib.ucf_plan_destroy.argtypes = [vp]
    lib.ucf_plan_create_on.argtypes = [C.POINTER(UcfParams), C.c_int, C.POINTER(vp)]
    lib.ucf_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.ucf_plan_update.argtypes = [vp, C.POINTER(UcfParams)]
    lib.ucf_plan_destroy.restype = None
    lib.ucf_plan_derived.argtypes = [vp, C.POINTER(UcfDerived)]
    lib.ucf_nondimensionalise.argtypes = [C.POINTER(UcfParams), C.POINTER(UcfDerived)]
    lib.ucf_plan_j0z.argtypes = [vp, C.c_int, _dp]
    lib.ucf_plan_tanh_sinh.argtypes = [vp, C.c_int, C.c_int, _dp, vp]
    lib.ucf_plan_gauss_lobatto.argtypes = [vp, C.c_int, _dp, _dp]
    lib.ucf_plan_set_mode.argtypes = [vp, C.c_int]
    lib.ucf_plan_set_timing.argtypes = [vp, C.c_int]
    lib.ucf_plan_kernel_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_char_p)]
    lib.ucf_plan_kernel_times.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
    lib.ucf_plan_reserve.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.ucf_plan_alloc_count.argtypes = [vp]
    lib.ucf_plan_alloc_count.restype = C.c_longlong
    lib.ucf_build_id.restype = C.c_char_p
    lib.ucf_shard_rows.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.ucf_drawdown_grid_shard_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_int, _dp, _ip, vp, vp, vp, vp]
    lib.ucf_drawdown_grid_allgather.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_int, _dp, _ip, vp, vp, vp, vp, vp]
    lib.ucf_comm_unique_id.argtypes = [C.c_char_p]
    lib.ucf_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
    lib.ucf_comm_destroy.argtypes = [vp]
    lib.ucf_drawdown_grid_multi.argtypes = [C.POINTER(vp), C.c_int, C.c_int, _dp, _ip, C.c_int, _dp, C.c_int, _dp, _ip, _dp, _dp,
                                            C.POINTER(UcfStats)]
    lib.ucf_drawdown_batch_multi.argtypes = [C.POINTER(vp), C.c_int, C.c_int, _dp, _dp, _ip, C.c_int, _dp, _ip, _dp, _dp, C.POINTER(UcfStats)]
    lib.ucf_logspace.argtypes = [C.c_int, C.c_int, C.c_int, _dp]
    lib.ucf_linspace.argtypes = [C.c_double, C.c_double, C.c_int, _dp]
    lib.ucf_zlay.argtypes = [vp, C.c_int, _dp, _ip]
    lib.ucf_split_vector.argtypes = [vp, C.c_int, _dp, _ip]
    lib.ucf_drawdown_batch.argtypes = [vp, C.c_int, _dp, _dp, _ip, C.c_int, _dp, _ip, _dp, _dp, C.POINTER(UcfStats)]
    lib.ucf_drawdown_batch_device.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, _dp, _ip, vp, vp, vp, vp]
    lib.ucf_drawdown_grid.argtypes = [vp, C.c_int, _dp, _ip, C.c_int, _dp, C.c_int, _dp, _ip, _dp, _dp, C.POINTER(UcfStats)]
    lib.ucf_drawdown_grid_device.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, C.c_int, _dp, _ip, vp, vp, vp, vp]
    lib.ucf_drawdown_multi.argtypes = [C.POINTER(vp), C.c_int, C.c_int, _dp, _dp, C.c_int, _dp, C.c_int, _dp, _dp]
    lib.ucf_screen_average.argtypes = [C.c_int, C.c_int, _dp, _dp]
    lib.ucf_eval_samples.argtypes = [vp, C.c_int, _dp, C.c_double, C.c_int, _dp, C.c_int, _dp, _ip, _dp]
    lib.ucf_pvalues.argtypes = [vp, C.c_double, _dp]
    lib.ucf_dehoog.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, _dp, _dp, _dp, _dp]
    lib.ucf_wynn_epsilon.argtypes = [C.c_int, C.c_int, _dp, _dp, _ip]
    lib.ucf_extraptozero.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp]
    lib.ucf_bessel_k01.argtypes = [C.c_int, _dp, _dp, _ip]
    lib.ucf_debug_stages.argtypes = [vp, C.c_int, C.c_int, _dp, _ip, C.c_int, _dp, C.c_int, _dp, _ip, _dp, _ip, _dp, _dp, _dp, _ip]
    lib.ucf_debug_wynn.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _ip]
    lib.ucf_debug_dehoog_tiles.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _dp, _dp, _dp, _dp]
    lib.ucf_fp64_fma_peak.argtypes = [C.POINTER(C.c_double)]
    lib.ucf_sincos_table.argtypes = [C.POINTER(C.c_double)]
    lib.ucf_exp2_table.argtypes = [C.POINTER(C.c_double)]
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        lib = load()
        raise UcfError(rc, (lib.ucf_last_error() or b"").decode() or lib.ucf_status_string(rc).decode())
