"""Raw hipEvent_t pairs (via ctypes on the HIP runtime torch already loaded) for kernel-exact timing
with cp2_ema_flat_timed: the events are attached to the launch itself (hipExtLaunchKernelGGL), so the
elapsed time is the kernel's own duration, as rocprofv3 --kernel-trace reports it."""
import ctypes

_hip = None


def _rt():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        _hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        _hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
    return _hip


class EventPair:
    def __init__(self):
        self.start, self.stop = ctypes.c_void_p(), ctypes.c_void_p()
        for ev in (self.start, self.stop):
            rc = _rt().hipEventCreate(ctypes.byref(ev))
            if rc:
                raise RuntimeError(f"hipEventCreate failed: {rc}")

    def elapsed_ms(self) -> float:
        """Valid after the launch stream has been synchronised."""
        ms = ctypes.c_float()
        rc = _rt().hipEventElapsedTime(ctypes.byref(ms), self.start, self.stop)
        if rc:
            raise RuntimeError(f"hipEventElapsedTime failed: {rc}")
        return float(ms.value)

    def __del__(self):
        try:
            _rt().hipEventDestroy(self.start), _rt().hipEventDestroy(self.stop)
        except Exception:
            pass
