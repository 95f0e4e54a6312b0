"""Device buffers with a REAL guard behind them, for the over-read tests: HIP's virtual-memory API reserves an address range one
granule longer than the buffer and maps physical memory only under the buffer, so the first byte past it is unmapped by
construction — whatever the caching allocator's state (PyTorch's `torch.empty` may hand out the middle of a cached segment, which
made part of round 2's guard runs vacuous).  Test infrastructure only."""
import ctypes

import torch

_HIP = None


class _Loc(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("id", ctypes.c_int)]


class _Flags(ctypes.Structure):
    _fields_ = [("compressionType", ctypes.c_ubyte), ("gpuDirectRDMACapable", ctypes.c_ubyte), ("usage", ctypes.c_ushort)]


class _Prop(ctypes.Structure):            # hipMemAllocationProp
    _fields_ = [("type", ctypes.c_int), ("requestedHandleType", ctypes.c_int), ("location", _Loc),
                ("win32HandleMetaData", ctypes.c_void_p), ("allocFlags", _Flags)]


class _Access(ctypes.Structure):          # hipMemAccessDesc
    _fields_ = [("location", _Loc), ("flags", ctypes.c_int)]


def _hip():
    global _HIP
    if _HIP is None:
        _HIP = ctypes.CDLL("libamdhip64.so")
    return _HIP


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with hipError {rc}")


class GuardedBuffer:
    """`nbytes` of device memory whose LAST byte is the last mapped byte of its address range."""

    def __init__(self, nbytes, device_index=0):
        hip = _hip()
        torch.cuda.init()
        prop = _Prop(1, 0, _Loc(1, device_index), None, _Flags(0, 0, 0))     # pinned device memory on `device_index`
        gran = ctypes.c_size_t()
        _check(hip.hipMemGetAllocationGranularity(ctypes.byref(gran), ctypes.byref(prop), 0), "hipMemGetAllocationGranularity")
        g = max(gran.value, 1 << 21)
        self.size = -(-max(1, nbytes) // g) * g
        self.reserved = self.size + g                                         # + one granule that stays unmapped
        self.base = ctypes.c_void_p()
        _check(hip.hipMemAddressReserve(ctypes.byref(self.base), ctypes.c_size_t(self.reserved), ctypes.c_size_t(0), None, ctypes.c_ulonglong(0)),
               "hipMemAddressReserve")
        self.handle = ctypes.c_void_p()
        _check(hip.hipMemCreate(ctypes.byref(self.handle), ctypes.c_size_t(self.size), ctypes.byref(prop), ctypes.c_ulonglong(0)), "hipMemCreate")
        _check(hip.hipMemMap(self.base, ctypes.c_size_t(self.size), ctypes.c_size_t(0), self.handle, ctypes.c_ulonglong(0)), "hipMemMap")
        acc = _Access(_Loc(1, device_index), 3)
        _check(hip.hipMemSetAccess(self.base, ctypes.c_size_t(self.size), ctypes.byref(acc), ctypes.c_size_t(1)), "hipMemSetAccess")
        self.nbytes = nbytes
        self.ptr = self.base.value + self.size - nbytes                       # flush against the end of the mapping
        self.device_index = device_index

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}

    def tensor(self, shape, dtype):
        t = torch.as_tensor(self, device=f"cuda:{self.device_index}")
        return t.view(dtype).view(shape)

    def close(self):
        if self.base is not None and self.base.value:
            torch.cuda.synchronize()
            hip = _hip()
            hip.hipMemUnmap(self.base, ctypes.c_size_t(self.size))
            hip.hipMemRelease(self.handle)
            hip.hipMemAddressFree(self.base, ctypes.c_size_t(self.reserved))
            self.base = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass
