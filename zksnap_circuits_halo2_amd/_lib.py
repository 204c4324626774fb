"""ctypes binding of libzkhip.so (include/zkhip.h).  Fails loudly when the library is missing: there is no
CPU fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzkhip.so")

# every symbol include/zkhip.h declares (tests check the export list against the header)
_SIGS = {
    "zkhip_init": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "zkhip_shutdown": (None, []),
    "zkhip_last_error": (C.c_char_p, []),
    "zkhip_device_name": (C.c_int, [C.c_char_p, C.c_size_t]),
    "zkhip_device_count": (C.c_int, []),
    "zkhip_set_msm_shards": (C.c_int, [C.c_int]),
    "zkhip_msm_shards": (C.c_int, []),
    "zkhip_msm_g1": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "zkhip_msm_g2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "zkhip_msm_g2_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zkhip_test_g2_op": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "zkhip_msm_g1_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "zkhip_ntt_fr_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "zkhip_ifft_scaled_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]),
    "zkhip_coeff_to_extended_batch": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "zkhip_vm_jit_source": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "zkhip_vm_jit_compile": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "zkhip_set_ntt_fanout": (C.c_int, [C.c_int]),
    "zkhip_ntt_fanout": (C.c_int, []),
    "zkhip_register_bases": (C.c_int, [C.c_void_p, C.c_size_t]),
    "zkhip_unregister_bases": (C.c_int, [C.c_void_p]),
    "zkhip_ntt_fr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "zkhip_ifft_scaled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "zkhip_coeff_to_extended": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "zkhip_extended_to_coeff": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "zkhip_mul_periodic": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32]),
    "zkhip_fr_eval_polynomial": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zkhip_fr_kate_division": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zkhip_fr_batch_invert": (C.c_int, [C.c_void_p, C.c_size_t]),
    "zkhip_fr_prefix_product": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "zkhip_fr_eval_polynomial_device": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "zkhip_fr_eval_polynomial_batch_device": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "zkhip_fr_kate_division_device": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "zkhip_fr_batch_invert_device": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "zkhip_fr_prefix_product_device": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zkhip_fr_eval_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]),
    "zkhip_fr_eval_rows_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    "zkhip_fr_grand_product": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "zkhip_fr_grand_product_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zkhip_lookup_permute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zkhip_lookup_permute_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "zkhip_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "zkhip_free": (C.c_int, [C.c_void_p]),
    "zkhip_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "zkhip_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "zkhip_sync": (C.c_int, []),
    "zkhip_stream_sync": (C.c_int, [C.c_void_p]),
    "zkhip_msm_g1_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zkhip_prepare_bases_device": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]),
    "zkhip_g1_fft_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "zkhip_g_to_lagrange_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "zkhip_g_to_lagrange": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "zkhip_prepare_bases_device_c": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint64)]),
    "zkhip_release_bases": (C.c_int, [C.c_uint64]),
    "zkhip_prepared_window_bits": (C.c_int, [C.c_uint64]),
    "zkhip_msm_g1_prepared_device": (C.c_int, [C.c_uint64, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zkhip_msm_g1_prepared_batch_device": (C.c_int, [C.c_uint64, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zkhip_msm_g1_device_c": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]),
    "zkhip_ntt_fr_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "zkhip_ntt_fr_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t, C.c_void_p]),
    "zkhip_ifft_scaled_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_size_t, C.c_void_p]),
    "zkhip_ifft_scaled_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "zkhip_mul_periodic_device": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_void_p]),
    "zkhip_coeff_to_extended_device": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "zkhip_extended_to_coeff_device": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_void_p]),
    "zkhip_g1_sum_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "zkhip_g1_sum": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "zkhip_msm_window_bits": (C.c_int, [C.c_size_t]),
    "zkhip_g1_fixed_base_mul_device": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zkhip_g1_gen_walk_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zkhip_g1_batch_normalize": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "zkhip_g1_batch_normalize_device": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zkhip_g1_check_points": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]),
    "zkhip_g1_check_points_device": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.c_void_p]),
    "zkhip_msm_g1_registered_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zkhip_msm_g1_registered_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zkhip_fr_gather_mul_device": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zkhip_g1_compress": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]),
    "zkhip_g1_compress_device": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]),
    "zkhip_g1_decompress": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]),
    "zkhip_g1_decompress_device": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.c_void_p]),
    "zkhip_permutation_products": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "zkhip_permutation_products_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p]),
    "zkhip_fr_eval_rows_sum_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "zkhip_fr_linear_combination_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zkhip_multiopen_gwc_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "zkhip_multiopen_shplonk_begin_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "zkhip_multiopen_shplonk_finish_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "zkhip_multiopen_shplonk_abort": (C.c_int, [C.c_void_p]),
    "zkhip_profile_enable": (C.c_int, [C.c_int]),
    "zkhip_profile_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "zkhip_profile_read_calls": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "zkhip_test_field_op": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "zkhip_test_g1_op": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
}



class ProverQueryC(C.Structure):   # zkhip_prover_query (80 bytes)
    _fields_ = [("point", C.c_uint64 * 4), ("d_poly", C.c_void_p), ("eval", C.c_uint64 * 4), ("has_eval", C.c_uint32), ("reserved", C.c_uint32)]


class VmOperand(C.Structure):      # zkhip_vm_operand
    _fields_ = [("kind", C.c_uint8), ("rot", C.c_uint8), ("index", C.c_uint16)]


class VmInsn(C.Structure):         # zkhip_vm_insn (16 bytes)
    _fields_ = [("op", C.c_uint8), ("dst", C.c_uint8), ("reserved", C.c_uint16), ("a", VmOperand), ("b", VmOperand), ("c", VmOperand)]


class VmProgram(C.Structure):      # zkhip_vm_program
    _fields_ = [("insns", C.POINTER(VmInsn)), ("n_insns", C.c_uint32),
                ("constants", C.c_void_p), ("n_constants", C.c_uint32),
                ("rotations", C.POINTER(C.c_int32)), ("n_rotations", C.c_uint32),
                ("rot_scale", C.c_int32), ("result_reg", C.c_uint32), ("omega", C.c_void_p)]


VM_REGS = 16
_lib = None


class ZkhipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libzkhip error {code}: {msg}")
        self.code = code


def load() -> C.CDLL:
    """Load libzkhip.so (built in-tree by `__graft_entry__.build()` / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("ZKHIP_LIB_PATH", LIB_PATH)      # A/B measurements of two builds on one box (tools/): never set in production
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc, gfx950).  There is no CPU fallback.")
    # Load order matters: libtorch_hip.so asks for "libamdhip64.so" (no version), so if libzkhip.so came first and bound
    # /opt/rocm's libamdhip64.so.7, torch would later load its own bundled copy -> two HIP runtimes in one process, and the
    # second one finds no GPU.  With torch first, our libamdhip64.so.7 request resolves to the copy torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGS)


def check(rc: int) -> None:
    if rc != 0:
        raise ZkhipError(rc, load().zkhip_last_error().decode(errors="replace"))
