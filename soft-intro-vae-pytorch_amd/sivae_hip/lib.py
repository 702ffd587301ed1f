"""ctypes binding of libsivae_hip.so (the C ABI declared in include/sivae_hip.h).

The prototypes are parsed from the header itself, so the Python side cannot drift from the ABI: a
symbol that is declared but not exported (or the other way round) fails at import time.

There is NO fallback: if the shared library is missing, `load()` raises. The product path never
routes through torch ops or the CPU oracle for the kernels declared in the header.
"""
import ctypes
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(os.path.dirname(_HERE))
# SIVAE_LIB: another build of the same ABI (A/B measurements inside one process launch — tools/profile.sh `ab`)
LIB_PATH = os.environ.get("SIVAE_LIB") or os.path.join(_HERE, "libsivae_hip.so")
HEADER_PATH = os.path.join(_REPO, "include", "sivae_hip.h")
BUILD_SCRIPT = os.path.join(os.path.dirname(_HERE), "csrc", "build.sh")

ERR_NAMES = {
    0: "SIVAE_OK",
    -1: "SIVAE_ERR_NULL",
    -2: "SIVAE_ERR_SHAPE",
    -3: "SIVAE_ERR_KSIZE",
    -4: "SIVAE_ERR_WORKSPACE",
    -5: "SIVAE_ERR_RANGE",
    -6: "SIVAE_ERR_MODE",
}


class SivaeError(RuntimeError):
    def __init__(self, fn, code):
        self.fn, self.code = fn, code
        name = ERR_NAMES.get(code, "hipError_t %d" % code if code > 0 else "unknown")
        super().__init__("%s failed: %d (%s)" % (fn, code, name))


_CTYPE = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "size_t": ctypes.c_size_t,
    "unsigned long long": ctypes.c_ulonglong,
    "long long": ctypes.c_longlong,
    "sivae_stream_t": ctypes.c_void_p,
    "const char*": ctypes.c_char_p,
}


def _ctype_of(decl):
    decl = re.sub(r"/\*.*?\*/", "", decl).strip()
    if "*" in decl and not decl.startswith("const char"):
        return ctypes.c_void_p
    # drop the parameter name
    for key in sorted(_CTYPE, key=len, reverse=True):
        if decl.startswith(key):
            return _CTYPE[key]
    raise ValueError("cannot map C type: %r" % decl)


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"\b(int|size_t|const char\*)\s+(sivae_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                argtypes.append(_ctype_of(a.strip()))
        protos[name] = (_CTYPE.get(ret, ctypes.c_int) if ret != "const char*" else ctypes.c_char_p, argtypes)
    return protos


_lib = None
_protos = None


def build(force=False):
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["bash", BUILD_SCRIPT])
    return LIB_PATH


def load():
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libsivae_hip.so not found at %s — build it with `bash %s` (hipcc, gfx950). "
            "There is no CPU/torch fallback for the Soft-IntroVAE kernels." % (LIB_PATH, BUILD_SCRIPT))
    lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (restype, argtypes) in _protos.items():
        fn = getattr(lib, name, None)
        if fn is None:
            # a declared symbol the .so does not export is an error — except for an OLDER build of the ABI selected with
            # SIVAE_LIB for an A/B measurement (tools/profile.sh `ab`), which may lack the newest entry points
            if os.environ.get("SIVAE_LIB") and os.environ.get("SIVAE_LIB_ALLOW_MISSING") == "1":
                continue
            raise AttributeError("%s does not export %s (declared in %s)" % (LIB_PATH, name, HEADER_PATH))
        fn.restype = restype
        fn.argtypes = argtypes
    ver = lib.sivae_abi_version()
    if ver != 1:
        raise RuntimeError("libsivae_hip ABI version %d, expected 1" % ver)
    _lib = lib
    return lib


def prototypes():
    load()
    return dict(_protos)


_sha = None


def sha256():
    """sha256 over the kernel SOURCES the library is built from (csrc/*.hip, *.h, *.inc, build.sh, in name order) — what a PMC
    summary is stamped with (tools/profile.sh -> tools/pmc_*.py --lib-sha) and what bench.py compares against, so that
    counter figures taken with other kernels are marked stale.  (The sources, not the .so: a rebuild of the same sources
    on another machine need not be byte-identical.)"""
    global _sha
    if _sha is None:
        import hashlib
        h = hashlib.sha256()
        src = os.path.dirname(BUILD_SCRIPT)
        for name in sorted(os.listdir(src)):
            if name.endswith((".hip", ".h", ".inc", ".sh")):
                h.update(name.encode())
                with open(os.path.join(src, name), "rb") as f:
                    h.update(f.read())
        _sha = h.hexdigest()
    return _sha


def call(name, *args):
    """Call an int-returning entry point and raise SivaeError on a non-zero status."""
    fn = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        raise SivaeError(name, rc)
    return rc
