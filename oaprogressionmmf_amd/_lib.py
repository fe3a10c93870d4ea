"""ctypes binding of libkoaf.so (the C ABI declared in include/koaf.h).

The prototypes are parsed from the header itself, so the Python side cannot drift from the C side.
There is no CPU fallback: if the library is missing, `lib()` raises -- the product path must fail
loudly rather than silently run something else.
"""
import ctypes
import os
import re
from pathlib import Path

_ROOT = Path(__file__).resolve().parent
HEADER = _ROOT.parent / "include" / "koaf.h"
LIB_PATH = _ROOT / "csrc" / "libkoaf.so"


class KoafOperand(ctypes.Structure):
    _fields_ = [
        ("ptr", ctypes.c_void_p),
        ("ld", ctypes.c_int64),
        ("bs0", ctypes.c_int64),
        ("bs1", ctypes.c_int64),
        ("tap_stride", ctypes.c_int64),
        ("tap_stride_h", ctypes.c_int64),
        ("kind", ctypes.c_int32),
        ("gather", ctypes.c_int32),
        ("H", ctypes.c_int32),
        ("W", ctypes.c_int32),
        ("C", ctypes.c_int32),
        ("CS", ctypes.c_int32),
        ("PH", ctypes.c_int32),
        ("PW", ctypes.c_int32),
        ("KH", ctypes.c_int32),
        ("KW", ctypes.c_int32),
        ("stride", ctypes.c_int32),
        ("pad", ctypes.c_int32),
        ("pad_w", ctypes.c_int32),
        ("_pad1", ctypes.c_int32),
        ("tf", ctypes.c_int32),
        ("tf_bs", ctypes.c_int32),
        ("sc", ctypes.c_void_p),
        ("sh", ctypes.c_void_p),
        ("planes", ctypes.c_void_p),
        ("plane_stride", ctypes.c_int64),
        ("amax", ctypes.c_void_p),
        ("fscale", ctypes.c_float),
        ("_pad4", ctypes.c_int32),
        ("ptr2", ctypes.c_void_p),
        ("sc2", ctypes.c_void_p),
        ("zeros", ctypes.c_void_p),
        ("side", ctypes.c_void_p),
        ("sh2", ctypes.c_void_p),
    ]


class KoafGemm(ctypes.Structure):
    _fields_ = [
        ("A", KoafOperand),
        ("B", KoafOperand),
        ("M", ctypes.c_int32),
        ("N", ctypes.c_int32),
        ("K", ctypes.c_int32),
        ("nb0", ctypes.c_int32),
        ("nb1", ctypes.c_int32),
        ("splitk", ctypes.c_int32),
        ("bm", ctypes.c_int32),
        ("bn", ctypes.c_int32),
        ("C", ctypes.c_void_p),
        ("ldc", ctypes.c_int64),
        ("cbs0", ctypes.c_int64),
        ("cbs1", ctypes.c_int64),
        ("alpha", ctypes.c_float),
        ("prec", ctypes.c_int32),
        ("bias", ctypes.c_void_p),
        ("residual", ctypes.c_void_p),
        ("ldr", ctypes.c_int64),
        ("rbs0", ctypes.c_int64),
        ("rbs1", ctypes.c_int64),
        ("stats", ctypes.c_void_p),
        ("stats_ld", ctypes.c_int64),
        ("stats_bs", ctypes.c_int64),
        ("cmap", ctypes.c_int32),
        ("cm_PH", ctypes.c_int32),
        ("cm_PW", ctypes.c_int32),
        ("cm_H", ctypes.c_int32),
        ("cm_W", ctypes.c_int32),
        ("cm_py", ctypes.c_int32),
        ("cm_px", ctypes.c_int32),
        ("_pad2", ctypes.c_int32),
        ("bnb_mode", ctypes.c_int32),
        ("fmt", ctypes.c_int32),
        ("bnb_c", ctypes.c_void_p),
        ("bnb_y", ctypes.c_void_p),
        ("bnb_sc", ctypes.c_void_p),
        ("bnb_sh", ctypes.c_void_p),
        ("bnb_mean", ctypes.c_void_p),
        ("bnb_invstd", ctypes.c_void_p),
        ("bnb2_c", ctypes.c_void_p),
        ("bnb2_mean", ctypes.c_void_p),
        ("bnb2_invstd", ctypes.c_void_p),
        ("bnb_part", ctypes.c_void_p),
        ("bnb_amax", ctypes.c_void_p),
        ("m_base", ctypes.c_int32),
        ("part_row0", ctypes.c_int32),
        ("stats_shift", ctypes.c_void_p),
        ("status", ctypes.c_void_p),
        ("act16", ctypes.c_int32),
        ("_pad5", ctypes.c_int32),
        ("out_planes", ctypes.c_void_p),
        ("out_sc", ctypes.c_void_p),
        ("out_sh", ctypes.c_void_p),
        ("out_ps", ctypes.c_int64),
    ]


class KoafBnb(ctypes.Structure):
    _fields_ = [
        ("mode", ctypes.c_int32),
        ("_pad", ctypes.c_int32),
        ("dz_amax", ctypes.c_void_p),
        ("c", ctypes.c_void_p),
        ("y", ctypes.c_void_p),
        ("sc", ctypes.c_void_p),
        ("sh", ctypes.c_void_p),
        ("mean", ctypes.c_void_p),
        ("invstd", ctypes.c_void_p),
        ("c2", ctypes.c_void_p),
        ("mean2", ctypes.c_void_p),
        ("invstd2", ctypes.c_void_p),
    ]


class KoafWPlane(ctypes.Structure):
    _fields_ = [
        ("src_off", ctypes.c_int64),
        ("f_off", ctypes.c_int64),
        ("d_off", ctypes.c_int64),
        ("tile0", ctypes.c_int64),
        ("R", ctypes.c_int32),
        ("taps", ctypes.c_int32),
        ("C", ctypes.c_int32),
        ("Kp", ctypes.c_int32),
        ("Rp", ctypes.c_int32),
        ("_pad", ctypes.c_int32),
    ]


class KoafWImg(ctypes.Structure):
    _fields_ = [("f", ctypes.c_void_p), ("d", ctypes.c_void_p), ("amax", ctypes.c_void_p)]


class KoafTail(ctypes.Structure):
    _fields_ = [("idt", ctypes.c_void_p), ("y_out", ctypes.c_void_p), ("idt_sc", ctypes.c_void_p), ("idt_sh", ctypes.c_void_p)]


class KoafEmit(ctypes.Structure):
    _fields_ = [("planes", ctypes.c_void_p), ("sc", ctypes.c_void_p), ("sh", ctypes.c_void_p)]


class KoafBnApply(ctypes.Structure):
    _fields_ = [("dz", ctypes.c_void_p), ("c", ctypes.c_void_p), ("coef", ctypes.c_void_p), ("amax", ctypes.c_void_p)]


_SCALARS = {
    "int": ctypes.c_int,
    "int32_t": ctypes.c_int32,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "uint32_t": ctypes.c_uint32,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
}


def _ctype(decl: str):
    decl = decl.strip()
    if "*" in decl:
        base = decl.replace("const", "").replace("*", "").split()[0]
        if base == "KoafGemm":
            return ctypes.POINTER(KoafGemm)
        if base == "KoafBnb":
            return ctypes.POINTER(KoafBnb)
        if base == "KoafWImg":
            return ctypes.POINTER(KoafWImg)
        if base == "KoafBnApply":
            return ctypes.POINTER(KoafBnApply)
        if base == "KoafTail":
            return ctypes.POINTER(KoafTail)
        if base == "KoafEmit":
            return ctypes.POINTER(KoafEmit)
        if base == "char":
            return ctypes.c_char_p
        return ctypes.c_void_p
    toks = decl.replace("const", "").split()
    return _SCALARS[toks[0]]


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function prototype in koaf.h"""
    text = Path(path).read_text()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"typedef struct.*?\}\s*\w+;", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)          # preprocessor lines
    protos = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(koaf_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if args in ("void", ""):
            argtypes = []
        else:
            argtypes = []
            for a in args.split(","):
                a = a.strip()
                # drop the parameter name (last identifier) unless it is part of the type
                mm = re.match(r"(.*?)(\w+)$", a)
                argtypes.append(_ctype(mm.group(1) if mm.group(1).strip() else a))
        protos[name] = (_ctype(ret) if ret != "void" else None, argtypes)
    return protos


_LIB = None


class KoafError(RuntimeError):
    pass


def lib():
    """Load libkoaf.so (once).  Raises if the HIP extension has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = Path(os.environ.get("KOAF_LIB", LIB_PATH))
    if not path.exists():
        raise KoafError(
            f"libkoaf.so not found at {path}: build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C oaprogressionmmf_amd/csrc` -- there is no CPU fallback")
    handle = ctypes.CDLL(str(path))
    for name, (restype, argtypes) in parse_header().items():
        fn = getattr(handle, name)  # AttributeError if a declared symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _LIB = handle
    return _LIB


def check(rc, what=""):
    if rc != 0:
        msg = lib().koaf_last_error()
        raise KoafError(f"{what}: rc={rc}: {msg.decode() if msg else '?'}")
