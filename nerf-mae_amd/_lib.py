"""ctypes binding of libnerfmae_hip.so (C ABI: include/nerfmae_hip.h).

Signatures are parsed from the header, so the binding cannot drift from the ABI.  There is NO fallback:
if the shared library is missing or fails to load, importing the ops raises (the product path must
fail loudly without the HIP extension)."""
from __future__ import annotations

import ctypes

import torch  # noqa: F401  (must be imported BEFORE the .so is dlopen-ed: torch bundles its own libamdhip64; loading ours first would put two HIP runtimes in one process)
import os
import re
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(ROOT, "include", "nerfmae_hip.h")
LIB_PATH = os.environ.get("NMH_LIB_PATH") or os.path.join(_HERE, "csrc", "libnerfmae_hip.so")   # (NMH_LIB_PATH: another build of the same ABI, for same-box A/B runs)

_CT = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "double": ctypes.c_double,
    "void": None, "const char*": ctypes.c_char_p,
}


def _ctype(t: str):
    t = " ".join(t.split())
    if t.endswith("*") and t != "const char*":
        return ctypes.c_void_p
    return _CT[t]


def parse_header(path: str = HEADER) -> Dict[str, Tuple[str, List[Tuple[str, str]]]]:
    """-> {name: (return type, [(arg type, arg name), ...])} for every NMH_API declaration."""
    src = open(path).read()
    out = {}
    for m in re.finditer(r"^NMH_API\s+(.+?)\s*\b(nmh_\w+)\s*\(([^;]*)\)\s*;", src, re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        lst = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.+?)\s*(\w+)$", a)
                lst.append((mm.group(1).strip(), mm.group(2)))
        out[name] = (ret, lst)
    return out


class NmhError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
                              "There is no CPU fallback for the NeRF-MAE HIP ops.")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.sigs = parse_header()
        for name, (ret, args) in self.sigs.items():
            fn = getattr(self.cdll, name)
            fn.restype = _ctype(ret)
            fn.argtypes = [_ctype(t) for t, _ in args]
        self.cdll.nmh_error_string.restype = ctypes.c_char_p
        self.profile = None  # dict -> every int-returning call is bracketed by HIP events on the launch stream (tools/op_breakdown.py)

    def call(self, name: str, *args):
        """Call an int-returning entry; tensors -> device pointers, None -> NULL; raises NmhError on failure."""
        ret, sig = self.sigs[name]
        if len(args) != len(sig):
            raise TypeError(f"{name}: expected {len(sig)} args ({[n for _, n in sig]}), got {len(args)}")
        conv = []
        for a, (t, n) in zip(args, sig):
            if t.endswith("*") and t != "const char*":
                if a is None:
                    conv.append(None)
                elif hasattr(a, "data_ptr"):
                    conv.append(a.data_ptr())
                elif isinstance(a, (ctypes.Array, ctypes.c_void_p)):
                    conv.append(ctypes.cast(a, ctypes.c_void_p))
                else:
                    conv.append(int(a))
            elif t in ("float", "double"):
                conv.append(float(a))
            else:
                conv.append(int(a))
        prof = self.profile
        if prof is not None and ret == "int":
            import torch as _t
            key = (name,) + tuple(a for a, (t, n) in zip(args, sig) if not t.endswith("*") and isinstance(a, int))
            e0, e1 = _t.cuda.Event(enable_timing=True), _t.cuda.Event(enable_timing=True)
            e0.record(_t.cuda.current_stream())
            rc = getattr(self.cdll, name)(*conv)
            e1.record(_t.cuda.current_stream())
            prof.setdefault(key, []).append((e0, e1))
        else:
            rc = getattr(self.cdll, name)(*conv)
        if ret == "int" and name != "nmh_version" and not name.endswith("_supported") and rc != 0:  # int64_t/char* returns and capability queries are values, not status codes
            raise NmhError(f"{name} failed with code {rc}: {self.cdll.nmh_error_string(rc).decode()}")
        return rc


_LIB = None


def lib() -> _Lib:
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB
