"""ctypes binding of libtacotron2_amd.so, generated at import time from include/tacotron2_amd.h (the header is
the single source of truth for struct layouts and signatures).

The product path FAILS LOUDLY when the HIP library is missing: there is no PyTorch/CPU fallback anywhere in
this package (a silent fallback would void every parity claim).
"""
from __future__ import annotations

import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "tacotron2_amd.h")
# (T2_LIB_PATH: another build of the same C ABI, e.g. the diagnostic one with in-kernel phase stamps - tacotron2_amd/build.py)
LIB_PATH = os.environ.get("T2_LIB_PATH") or os.path.join(HERE, "libtacotron2_amd.so")

_SCALARS = {"int": C.c_int, "float": C.c_float, "int64_t": C.c_int64, "int32_t": C.c_int32, "uint64_t": C.c_uint64,
            "double": C.c_double}


class T2Error(RuntimeError):
    pass


def _strip_comments(src: str) -> str:
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def _parse_header(path: str):
    src = _strip_comments(open(path).read())
    structs = {}
    order = []
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        body, name = m.group(1), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            mm = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*(.*)$", decl, flags=re.S)
            base, ptr, rest = mm.group(2), mm.group(3), mm.group(4)
            for item in rest.split(","):
                item = item.strip()
                is_ptr = bool(ptr)
                if item.startswith("*"):
                    is_ptr, item = True, item[1:].strip()
                am = re.match(r"(\w+)\s*\[(\d+)\]$", item)
                fname, n = (am.group(1), int(am.group(2))) if am else (item, 0)
                fields.append((fname, base, is_ptr, n))
        structs[name] = fields
        order.append(name)
    funcs = {}
    for m in re.finditer(r"\b(int64_t|int|const\s+char\s*\*)\s+(t2_\w+)\s*\((.*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        alist = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*(\w+)?$", a)
                alist.append((mm.group(2), bool(mm.group(3))))
        funcs[name] = (ret, alist)
    return structs, order, funcs


def _ctype(base: str, is_ptr: bool, structs_c: dict):
    if is_ptr:
        return C.c_void_p if base not in structs_c else C.POINTER(structs_c[base])
    if base in _SCALARS:
        return _SCALARS[base]
    if base in structs_c:
        return structs_c[base]
    raise T2Error(f"unknown C type {base}")


_structs, _order, _funcs = _parse_header(HEADER)
S: dict = {}
for _name in _order:
    _fields = []
    for fname, base, is_ptr, n in _structs[_name]:
        ct = _ctype(base, is_ptr, S)
        _fields.append((fname, ct * n if n else ct))
    S[_name] = type(_name, (C.Structure,), {"_fields_": _fields})

DECLARED_SYMBOLS = sorted(_funcs)

_lib = None


def lib():
    """Load the shared library (once).  Raises T2Error if it is missing - never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise T2Error(f"{LIB_PATH} not found: build it with `python -m tacotron2_amd.build` "
                      "(hipcc --offload-arch=gfx950). There is no fallback path.")
    # torch first: its wheel bundles its own libamdhip64 (torch/lib), and the dynamic loader only shares ONE HIP runtime
    # between torch and this library when torch's copy is already loaded (same SONAME); loaded the other way round the
    # process ends up with two runtimes and launches on torch's streams fail with "no ROCm-capable device"
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    for name, (ret, alist) in _funcs.items():
        fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = C.c_char_p if "char" in ret else (C.c_int64 if ret == "int64_t" else C.c_int)
        fn.argtypes = [C.c_void_p if p else _SCALARS[b] for b, p in alist]
    # ABI check: the struct layouts this binding mirrors from the header must be the ones the library was compiled with - a
    # stale build (or another library selected through T2_LIB_PATH) would otherwise read garbage operand blocks
    L.t2_sizeof.argtypes = [C.c_char_p]
    for name, st in S.items():
        got = L.t2_sizeof(name.encode())
        if got != C.sizeof(st):
            raise T2Error(f"{LIB_PATH}: struct {name} is {got} bytes in the library, {C.sizeof(st)} in include/tacotron2_amd.h - "
                          "rebuild it (python -m tacotron2_amd.build [--stamps])")
    _lib = L
    return L


def _addr(x):
    """torch.Tensor | ctypes struct | int | None -> void* value."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    if isinstance(x, C.Structure) or isinstance(x, C.Array):
        return C.addressof(x)
    return x


PRE_CALL = []      # hooks run in front of every library call (tacotron2_amd.engine: the deferred multi-region zero)


def call(name: str, *args):
    L = lib()
    for hook in PRE_CALL:
        hook()
    fn = getattr(L, name)
    conv = []
    for a, at in zip(args, fn.argtypes):
        conv.append(_addr(a) if at is C.c_void_p else a)
    rc = fn(*conv)
    if rc != 0:
        raise T2Error(f"{name} failed (rc={rc}): {L.t2_last_error().decode()}")


def call_value(name: str, *args):
    """Call a function whose return value is data (the *_plan functions return a byte count, -1 on error)."""
    L = lib()
    fn = getattr(L, name)
    conv = [(_addr(a) if at is C.c_void_p else a) for a, at in zip(args, fn.argtypes)]
    return fn(*conv)


def make(struct_name: str, **kw):
    """Build a header struct; tensors become device pointers, None -> NULL."""
    st = S[struct_name]()
    st._keep = [v for v in kw.values() if hasattr(v, "data_ptr")]   # keep operand tensors alive with the struct
    for k, v in kw.items():
        fld = getattr(st, k)
        if isinstance(fld, C.Array):
            for i, item in enumerate(v):
                fld[i] = item.data_ptr() if hasattr(item, "data_ptr") else item
            st._keep += [item for item in v if hasattr(item, "data_ptr")]
        else:
            setattr(st, k, _addr(v) if (hasattr(v, "data_ptr") or v is None) else v)
    return st


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
