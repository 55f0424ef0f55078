"""Stand-in for numba.extending.get_cython_function_address (TEST TOOL).

Returns the raw C function pointer stored in a Cython module's __pyx_capi__
capsule -- the same pointer numba would hand out.
"""
import ctypes
import importlib


def get_cython_function_address(module_name, function_name):
    capsule = importlib.import_module(module_name).__pyx_capi__[function_name]
    api = ctypes.pythonapi
    api.PyCapsule_GetName.restype = ctypes.c_char_p
    api.PyCapsule_GetName.argtypes = [ctypes.py_object]
    api.PyCapsule_GetPointer.restype = ctypes.c_void_p
    api.PyCapsule_GetPointer.argtypes = [ctypes.py_object, ctypes.c_char_p]
    return api.PyCapsule_GetPointer(capsule, api.PyCapsule_GetName(capsule))
