"""Test-support code that must not live in the product library (see squatter.hip)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def load():
    """ctypes handle of tests/support/libsivae_testsupport.so (built on first use when hipcc is present)"""
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libsivae_testsupport.so")
        if not os.path.exists(path):
            subprocess.check_call(["bash", os.path.join(_HERE, "build.sh")])
        _lib = ctypes.CDLL(path)
        _lib.testsupport_squatter.restype = ctypes.c_int
        _lib.testsupport_squatter.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong,
                                              ctypes.c_void_p, ctypes.c_void_p]
    return _lib
