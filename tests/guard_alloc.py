"""Device buffers with a REAL guard behind them, for the over-read tests: HIP's virtual-memory API reserves an address range one
granule longer than the buffer and maps physical memory only under the buffer, so the first byte past it is unmapped by
construction — whatever the caching allocator's state (PyTorch's `torch.empty` may hand out the middle of a cached segment, which
made part of round 2's guard runs vacuous).  Mappings are pooled and never unmapped while the process lives: a virtual address that
is unmapped and mapped again onto other physical pages is read STALE by shaders for a while - PyTorch's own kernels included, while
hipMemcpy of the same range returns the bytes written (root-caused in round 4: tools/remap_probe.py, profiles/r04/remap_probe.json,
profiles/r03/stale_input_note.md) - which is a property of hipMemUnmap / hipMemMap address re-use on this driver stack, not of the
code under test.  Test infrastructure only."""
import ctypes

import torch

_HIP = None
_POOL = []     # free mappings (base, size, reserved, handle): reused for any buffer that fits, never unmapped


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


class _Whole:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class GuardedBuffer:
    """`nbytes` of device memory whose LAST byte is the last mapped byte of its address range."""

    def __init__(self, nbytes, device_index=0):
        hip = _hip()
        torch.cuda.init()
        self.nbytes = nbytes
        self.device_index = device_index
        fit = [m for m in _POOL if m[4] == device_index and m[1] >= nbytes]
        if fit:
            m = min(fit, key=lambda m: m[1])
            _POOL.remove(m)
            self.base, self.size, self.reserved, self.handle, _ = m
        else:
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
        self.ptr = self.base.value + self.size - nbytes                       # flush against the end of the mapping

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}

    def tensor(self, shape, dtype):
        """View of the buffer; the mapped bytes in FRONT of it are filled with 0xFF first (NaN as f16 / bf16 / f32): the end is guarded
        by the unmapped granule, the start by poison that would show in the scores."""
        whole = _Whole(self.base.value, self.size)
        torch.as_tensor(whole, device=f"cuda:{self.device_index}")[: self.size - self.nbytes].fill_(0xFF)
        t = torch.as_tensor(self, device=f"cuda:{self.device_index}")
        return t.view(dtype).view(shape)

    def close(self):
        """Back to the pool (the mapping stays: see the module docstring)."""
        if self.base is not None:
            torch.cuda.synchronize()
            _POOL.append((self.base, self.size, self.reserved, self.handle, self.device_index))
            self.base = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass
