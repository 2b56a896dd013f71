"""ctypes binding of libvspw_hip.so (the C ABI declared in include/vspw_hip.h).

The header is the single source of truth: its declarations are parsed here to build the ctypes signatures, and the
CPU test-suite uses the same parser to check that the built library exports every declared symbol.
There is NO fallback: if the library is missing, or a tensor is not on the GPU, the product path raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "vspw_hip.h")
# VSPW_HIP_LIB: a diagnostic build of the same library (tools/diag/*: -DVSPW_NT_TIMING / -DVSPW_NT_DBG variants)
LIB_PATH = os.environ.get("VSPW_HIP_LIB") or os.path.join(_HERE, "lib", "libvspw_hip.so")


class ConvDesc(ctypes.Structure):
    """struct vspw_conv_desc (include/vspw_hip.h)."""

    _fields_ = [(n, ctypes.c_int) for n in ("n", "h", "w", "c", "oh", "ow", "k", "kh", "kw", "stride", "pad", "dil", "pad_w")]


_CTYPES = {
    "int": ctypes.c_int,
    "long long": ctypes.c_longlong,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "size_t": ctypes.c_size_t,
}


def _arg_ctype(decl: str):
    decl = decl.strip()
    if "*" in decl:
        if "vspw_conv_desc" in decl:
            return ctypes.POINTER(ConvDesc)
        if "float* const*" in decl:
            return ctypes.POINTER(ctypes.c_void_p)
        if decl.startswith("const int*"):
            return ctypes.POINTER(ctypes.c_int)
        return ctypes.c_void_p
    # strip the parameter name
    toks = decl.replace("const ", "").split()
    ty = " ".join(toks[:-1]) if len(toks) > 1 else toks[0]
    return _CTYPES[ty]


def parse_header(path: str = HEADER):
    """Return {name: (restype, [argtypes])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|size_t|long long)\s+(vspw_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = " ".join(args.split())
        if args in ("", "void"):
            argtypes = []
        else:
            argtypes = [_arg_ctype(a) for a in args.split(",")]
        out[name] = (_CTYPES[ret], argtypes)
    return out


_lib = None


def load(check_symbols: bool = False):
    """Load the shared library (once). Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None and not check_symbols:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libvspw_hip.so is missing (%s). Build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback for the hot path." % LIB_PATH
        )
    # torch bundles its own HIP/HSA runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's).  It must be in the
    # process BEFORE this library is opened, so that the library's libamdhip64 dependency resolves to that copy: with
    # the opposite order two HIP runtimes coexist and every launch here fails with hipErrorNoDevice.
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH)
    decls = parse_header()
    missing = []
    for name, (ret, argtypes) in decls.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = ret
        fn.argtypes = argtypes
    if missing:
        raise RuntimeError("libvspw_hip.so does not export: " + ", ".join(missing))
    if lib.vspw_abi_version() != 7:
        raise RuntimeError("libvspw_hip.so ABI version mismatch")
    _lib = lib
    return lib


_ERR = {-1: "VSPW_EINVAL (bad argument / geometry / workspace)", -2: "VSPW_ELAUNCH (HIP launch failure)"}


def check(rc: int, what: str):
    if rc != 0:
        detail = ""
        if rc == -2 and _lib is not None:
            try:
                fn = _lib.vspw_last_hip_error_string
                fn.restype = ctypes.c_char_p
                detail = " [hipError %d: %s]" % (_lib.vspw_last_hip_error(), fn().decode())
            except Exception:  # pragma: no cover - diagnostics only
                pass
        raise RuntimeError("%s failed: %s%s" % (what, _ERR.get(rc, str(rc)), detail))


trace = None  # optional callable(name, args) -> context manager (ops.kernel_timer: HIP events around HBM-bound kernels)


def call(name: str, *args):
    """Call an int-returning entry point and raise on a non-zero status."""
    if trace is not None:
        with trace(name, args):
            rc = getattr(load(), name)(*args)
    else:
        rc = getattr(load(), name)(*args)
    if rc != 0:
        check(rc, name)


def query(name: str, *args):
    """Call a size_t-returning *_workspace()/partials() entry point."""
    return getattr(load(), name)(*args)
