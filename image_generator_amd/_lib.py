"""ctypes binding of libmi355_denoiser.so (the C ABI declared in include/mi355_denoiser.h).

The prototypes are read from the header itself so the header stays the single source of truth.
There is no fallback: if the shared library is missing or a symbol cannot be bound, import fails.
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
# MI355_DENOISER_LIB selects another build of the same ABI (A/B comparisons of kernel variants)
LIB_PATH = os.environ.get("MI355_DENOISER_LIB") or os.path.join(_HERE, "libmi355_denoiser.so")
HEADER_PATH = os.path.join(_ROOT, "include", "mi355_denoiser.h")

_CTYPES = {
    "int": ctypes.c_int, "float": ctypes.c_float, "size_t": ctypes.c_size_t, "long long": ctypes.c_longlong,
    "double": ctypes.c_double, "unsigned": ctypes.c_uint,
}


def parse_header(path: str = HEADER_PATH):
    """Returns {name: (restype, [argtypes])} for every function the header declares."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = "\n".join(l for l in text.splitlines() if not l.strip().startswith("#"))
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(mi355_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        protos[name] = (_map_type(ret), [] if args in ("", "void") else [_map_type(_strip_name(a)) for a in args.split(",")])
    return protos


def _strip_name(arg: str) -> str:
    arg = arg.strip()
    if "*" in arg:
        return arg[: arg.rindex("*") + 1]
    return " ".join(arg.split()[:-1])


def _map_type(t: str):
    t = t.replace("const", "").strip()
    if t.endswith("*"):
        base = t[:-1].strip()
        if base == "char":
            return ctypes.c_char_p
        return ctypes.c_void_p
    if t == "void":
        return None
    return _CTYPES[" ".join(t.split())]


class LibraryError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                          "(there is no fallback path)")
    lib = ctypes.CDLL(LIB_PATH)
    other_build = bool(os.environ.get("MI355_DENOISER_LIB"))
    for name, (res, args) in parse_header().items():
        if other_build and not hasattr(lib, name):
            continue                     # an older build under A/B comparison: calls to what it lacks fail when made
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    # MI355_TUNE="key:value,..." -> mi355_tune_set (kernel-variant switches of the launchers, for A/B runs of whole programs)
    for item in filter(None, os.environ.get("MI355_TUNE", "").split(",")):
        key, value = item.split(":")
        lib.mi355_tune_set(int(key), int(value))
    return lib


lib = _load()


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib.mi355_last_error()
        raise LibraryError(f"{what or 'mi355 call'} failed ({rc}): {msg.decode() if msg else ''}")
