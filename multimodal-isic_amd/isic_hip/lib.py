"""ctypes binding of ``libisic_hip.so`` (the C ABI declared in ``include/isic_hip.h``).

The prototypes are parsed from the header itself, so the binding can never drift
from the declared ABI.  There is NO fallback: if the shared library is missing
or a symbol is absent, importing / calling raises immediately.
"""
from __future__ import annotations

import ctypes
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "libisic_hip.so")
_HEADER_CANDIDATES = (
    os.path.join(os.path.dirname(PKG), "include", "isic_hip.h"),
    os.path.join(PKG, "include", "isic_hip.h"),
)

ERRORS = {0: "ISIC_OK", -1: "ISIC_ERR_BAD_ARG", -2: "ISIC_ERR_UNSUPPORTED", -3: "ISIC_ERR_WORKSPACE",
          -4: "ISIC_ERR_LAUNCH"}

_SCALARS = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint32_t": ctypes.c_uint32,
    "uint64_t": ctypes.c_uint64, "size_t": ctypes.c_size_t, "float": ctypes.c_float, "double": ctypes.c_double,
}


class IsicHipError(RuntimeError):
    """``code``: the negative ISIC_ERR_* value of a failed C-ABI call (None for host-side errors)."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


ERR_UNSUPPORTED = -2


def header_path():
    for p in _HEADER_CANDIDATES:
        if os.path.exists(p):
            return p
    raise IsicHipError("include/isic_hip.h not found next to the package")


def test_header_path():
    """include/isic_hip_test.h: test / benchmark-only entry points (kernel variants pinned per call)."""
    return os.path.join(os.path.dirname(header_path()), "isic_hip_test.h")


def parse_header(path=None):
    """-> {name: (restype, [(ctype, param_name, is_pointer)])} for every prototype."""
    text = open(path or header_path()).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    protos = {}
    for m in re.finditer(r"\b(int|size_t|const\s+char\s*\*)\s+(isic_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        restype = ctypes.c_char_p if "char" in ret else _SCALARS[ret]
        args = []
        for prm in (p.strip() for p in params.split(",")):
            if not prm or prm == "void":
                continue
            is_ptr = "*" in prm
            toks = prm.replace("*", " * ").split()
            pname = toks[-1]
            base = [t for t in toks[:-1] if t not in ("const", "*")]
            if is_ptr:
                args.append((ctypes.c_void_p, pname, True))
            else:
                args.append((_SCALARS[base[-1]], pname, False))
        protos[name] = (restype, args)
    return protos


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise IsicHipError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU/PyTorch fallback for this path.")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self.public = set(self.protos)                      # the drop-in ABI (include/isic_hip.h)
        if os.path.exists(test_header_path()):
            self.protos.update(parse_header(test_header_path()))
        self.fn = {}
        for name, (restype, args) in self.protos.items():
            try:
                f = getattr(self.cdll, name)
            except AttributeError as e:
                raise IsicHipError(f"libisic_hip.so does not export {name} declared in isic_hip.h") from e
            f.restype = restype
            f.argtypes = [a[0] for a in args]
            self.fn[name] = f
        if self.fn["isic_target_arch"]() != b"gfx950":
            raise IsicHipError("libisic_hip.so was not built for gfx950")


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        if x.numel() == 0:
            return x.data_ptr() or None
        if not x.is_cuda:
            raise IsicHipError("libisic_hip expects device tensors (got a CPU tensor)")
        if not (x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))):
            raise IsicHipError("libisic_hip expects dense (contiguous or channels_last) tensors")
        return x.data_ptr()
    return int(x)


def current_stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args, stream=None):
    """Invoke ``name`` on the current torch stream; tensors -> device pointers,
    None -> NULL.  Raises ``IsicHipError`` on a non-zero return code."""
    L = lib()
    f = L.fn[name]
    spec = L.protos[name][1]
    has_stream = bool(spec) and spec[-1][1] == "stream"
    conv = []
    n_user = len(spec) - (1 if has_stream else 0)
    if len(args) != n_user:
        raise TypeError(f"{name} takes {n_user} arguments ({[s[1] for s in spec[:n_user]]}), got {len(args)}")
    for a, (ct, pname, is_ptr) in zip(args, spec):
        conv.append(_ptr(a) if is_ptr else a)
    if has_stream:
        conv.append(current_stream() if stream is None else stream)
    rc = f(*conv)
    if L.protos[name][0] is ctypes.c_int and rc != 0:
        raise IsicHipError(f"{name} failed: {ERRORS.get(rc, rc)}", code=rc)
    return rc
