"""ctypes binding of libctn_hip.so, generated from include/ctn_hip.h at import time.

There is no CPU or eager fallback: if the HIP library is missing the first use raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "ctn_hip.h")
# CTN_LIB_PATH: an alternative build of the same ABI (benchmarks/gemm_ablate_build.sh: phase-ablation builds)
LIB_PATH = os.environ.get("CTN_LIB_PATH") or os.path.join(_HERE, "libctn_hip.so")

_SCALARS = {"int": ctypes.c_int, "long long": ctypes.c_longlong, "float": ctypes.c_float,
            "size_t": ctypes.c_size_t, "double": ctypes.c_double}


def parse_header(path=HEADER):
    """-> {name: (restype, [argtype, ...], [argname, ...])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(ctn_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret == "const char*" or ret == "const char *":
            restype = ctypes.c_char_p
        else:
            restype = _SCALARS[ret]
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                nm = re.search(r"(\w+)$", a).group(1)
                ty = a[: -len(nm)].strip()
                argnames.append(nm)
                if "*" in ty:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_SCALARS[ty.replace("const ", "").strip()])
        protos[name] = (restype, argtypes, argnames)
    return protos


class CtnError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        self._dll = None
        self.probe = None       # list -> every lib.call is bracketed by two timing events (bench.py's roofline leg)
        self.protos = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise CtnError("HIP library %s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback for the product path)" % LIB_PATH)
            dll = ctypes.CDLL(LIB_PATH)
            for name, (res, args, _) in self.protos.items():
                fn = getattr(dll, name)   # AttributeError if the library lacks a declared symbol
                fn.restype = res
                fn.argtypes = args
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        if name.startswith("ctn_"):
            return getattr(self.load(), name)
        raise AttributeError(name)

    def call(self, name, *args):
        """Call an int-status entry point; raise CtnError with the library's message on failure."""
        if self.probe is not None:              # measurement hook (bench.py): HIP events around this call on its stream
            return self._probed_call(name, args)
        rc = getattr(self.load(), name)(*args)
        if rc != 0:
            raise CtnError("%s failed (%d): %s" % (name, rc, self._dll.ctn_last_error().decode()))

    def _probed_call(self, name, args):
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()                             # torch's current stream == the stream every wrapper hands to the C ABI
        rc = getattr(self.load(), name)(*args)
        e1.record()
        if rc != 0:
            raise CtnError("%s failed (%d): %s" % (name, rc, self._dll.ctn_last_error().decode()))
        self.probe.append((name, args, e0, e1))


lib = _Lib()
