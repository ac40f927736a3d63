"""The few HIP runtime calls GPU tests need (streams, device buffers), taken from the runtime the library itself
is linked with -- a second runtime in the process (torch bundles its own) does not find the GPU once this one
holds it."""
import ctypes as C

import numpy as np


class Hip:
    def __init__(self):
        self.rt = C.CDLL("libamdhip64.so")

    def ok(self, rc):
        assert rc == 0, "HIP error %d" % rc

    def stream(self):
        s = C.c_void_p()
        self.ok(self.rt.hipStreamCreate(C.byref(s)))
        return s

    def destroy_stream(self, s):
        self.ok(self.rt.hipStreamDestroy(s))

    def zeros(self, nbytes):
        p = C.c_void_p()
        self.ok(self.rt.hipMalloc(C.byref(p), C.c_size_t(nbytes)))
        self.ok(self.rt.hipMemset(p, 0, C.c_size_t(nbytes)))
        return p

    def free(self, p):
        self.ok(self.rt.hipFree(p))

    def to_host(self, p, shape):
        out = np.empty(shape, np.float32)
        self.ok(self.rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, C.c_size_t(out.nbytes), 2))
        return out

    def sync(self):
        self.ok(self.rt.hipDeviceSynchronize())

    def free_bytes(self):
        free, total = C.c_size_t(), C.c_size_t()
        self.ok(self.rt.hipMemGetInfo(C.byref(free), C.byref(total)))
        return int(free.value)
