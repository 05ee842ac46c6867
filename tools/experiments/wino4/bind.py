"""ctypes signatures of the F(4x4,3x3) experiment on the loaded library (ZSG_LIB_PATH must point at a build made with EXPERIMENTS=1)"""
import ctypes as C

from zsgnet_pytorch_amd import _lib as L


def bind():
    lib = L.lib
    if not hasattr(lib, "zsg_conv_wino4"):
        raise RuntimeError("this libzsg.so was built without the experiments: make -C zsgnet-pytorch_amd/csrc EXPERIMENTS=1 OUT=... and set ZSG_LIB_PATH")
    lib.zsg_wino4_u_elems.restype, lib.zsg_wino4_u_elems.argtypes = C.c_int64, [C.c_int32, C.c_int32]
    lib.zsg_wino4_weights.restype, lib.zsg_wino4_weights.argtypes = C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.zsg_conv_wino4.restype, lib.zsg_conv_wino4.argtypes = C.c_int32, [L.DP] + [C.c_void_p] * 7
    return lib
