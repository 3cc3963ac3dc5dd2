"""ctypes binding of libvolta_hip.so (include/volta_hip.h).  There is no CPU fallback: if the HIP
library is missing the import fails loudly (build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C volta_amd/csrc`)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvolta_hip.so")


class VoltaHipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "volta_amd: %s not found. The HIP kernels are the product; there is no fallback path. "
        "Build with `make -C volta_amd/csrc` (hipcc --offload-arch=gfx950)." % LIB_PATH)

lib = C.CDLL(LIB_PATH)

c_p = C.c_void_p
i32 = C.c_int32


class Dropout(C.Structure):
    _fields_ = [("seed", c_p), ("site", C.c_uint32), ("threshold", C.c_uint32), ("scale", C.c_float)]


def dropout_cfg(seed_ptr, site, p):
    if p <= 0.0 or not seed_ptr:
        return Dropout(None, 0, 0, 1.0)
    return Dropout(seed_ptr, site, min(int(p * 4294967296.0), 0xFFFFFFFF), 1.0 / (1.0 - p))


class GemmProblem(C.Structure):
    _fields_ = [("A", c_p), ("B", c_p), ("C", c_p), ("C2", c_p), ("bias", c_p), ("R", c_p), ("bias_grad", c_p),
                ("dyn", c_p), ("M", i32), ("N", i32), ("K", i32), ("lda", i32), ("ldb", i32), ("ldc", i32),
                ("ldr", i32), ("n_store", i32)]


class LnArgs(C.Structure):
    _fields_ = [("d", c_p), ("x", c_p), ("gamma", c_p), ("beta", c_p), ("y", c_p), ("z", c_p), ("mean", c_p),
                ("rstd", c_p), ("M", i32), ("H", i32), ("split_row", i32), ("post", i32), ("out_scale", C.c_float),
                ("drop", Dropout)]


class LnBwdArgs(C.Structure):
    _fields_ = [("dy", c_p), ("z", c_p), ("mean", c_p), ("rstd", c_p), ("gamma", c_p), ("dz", c_p), ("dd", c_p),
                ("partial", c_p), ("dgamma", c_p), ("dbeta", c_p), ("M", i32), ("H", i32), ("split_row", i32),
                ("post", i32), ("out_scale", C.c_float), ("drop", Dropout)]


class AttnArgs(C.Structure):
    _fields_ = [("q", c_p * 2), ("k", c_p * 2), ("v", c_p * 2), ("ld", i32 * 2), ("L", i32 * 2), ("mask", c_p * 2),
                ("ctx", c_p * 2), ("ldo", i32 * 2), ("lse", c_p * 2), ("B", i32), ("nh", i32), ("gate", (i32 * 2) * 2),
                ("drop", (Dropout * 2) * 2), ("scale", C.c_float)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("dctx", c_p * 2), ("dq", c_p * 2), ("dk", c_p * 2), ("dv", c_p * 2), ("ldg", i32 * 2)]


NT, NN, TN = 0, 1, 2
EPI_BF16, EPI_GELU, EPI_MULR, EPI_ADDR, EPI_F32, EPI_RELU = range(6)


def _sig(name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


_sig("vk_version", C.c_int)
_sig("vk_device_arch", C.c_char_p)
_sig("vk_last_error", C.c_char_p)
_sig("vk_set_seed", C.c_int, c_p, C.c_uint64, c_p)
_sig("vk_cast_f32_bf16", C.c_int, c_p, c_p, C.c_int64, c_p)
_sig("vk_gemm_grouped", C.c_int, C.c_int, C.c_int, C.POINTER(GemmProblem), C.c_int, c_p)
_sig("vk_gated_attn_fwd", C.c_int, C.POINTER(AttnArgs), c_p)
_sig("vk_gated_attn_bwd", C.c_int, C.POINTER(AttnArgs), C.POINTER(AttnBwdArgs), c_p)
_sig("vk_ln_fwd", C.c_int, C.POINTER(LnArgs), c_p)
_sig("vk_ln_bwd_partial_rows", C.c_int, C.c_int)
_sig("vk_ln_bwd", C.c_int, C.POINTER(LnBwdArgs), c_p)


def check(rc):
    if rc != 0:
        raise VoltaHipError(lib.vk_last_error().decode())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None
