"""Root-cause probe for round 3's stale-input observation (tests/guard_alloc.py docstring): with guarded mappings
UNMAPPED and re-made per buffer (pool off) one search once saw stale inputs.  Which buffer, and through which read path?

    python tools/remap_probe.py [cycles] [seed]

Per cycle, exactly what the guard tests did before the pool: reserve + create + map a fresh range for the corpus and one for the
queries (hipMemAddressReserve / hipMemCreate / hipMemMap, data copied in with a torch kernel), search, unmap + release + free.
Recycled virtual addresses and recycled physical pages then meet in new combinations every cycle.  After each fill, BEFORE the
search, the same bytes are read back three ways and compared with what was written:
    torch     : view.clone() (a plain vector-load kernel, default cache policy; a small grid: a few CUs)
    torch_wide: an int64 sum over 256 stride-0 copies of the buffer (a multi-block torch reduction: plain loads from MANY CUs on every
                XCD, which is what distinguishes the library's persistent 256-workgroup kernels from the clone above)
    memcpy    : hipMemcpy device -> host of the range (blit / SDMA path)
    rr        : the search itself against the oracle (corpus through non-temporal LDS-DMA, queries through global loads and the
                prep kernel, range / threshold tables through the scalar cache)
and again after the search.  Every mismatch is printed with the buffer, the path, the first differing offset, and whether the
wrong bytes equal what the PREVIOUS tenant of that virtual address (or of any earlier buffer) held.  One JSON summary at the end."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import oracle as O
from ragroute_amd.flat_index import FlatIndex
from tests import guard_alloc as G
from tests.util import int_data


class Fresh(G.GuardedBuffer):
    """GuardedBuffer without the pool: every buffer is a new reservation + mapping, close() unmaps and frees it."""

    def __init__(self, nbytes, device_index=0):
        saved, G._POOL[:] = list(G._POOL), []
        try:
            super().__init__(nbytes, device_index)
        finally:
            G._POOL[:] = saved

    def close(self):
        if self.base is not None and self.base.value:
            torch.cuda.synchronize()
            hip = G._hip()
            hip.hipMemUnmap(self.base, ctypes.c_size_t(self.size))
            hip.hipMemRelease(self.handle)
            hip.hipMemAddressFree(self.base, ctypes.c_size_t(self.reserved))
            self.base = None


def read_memcpy(buf, nbytes):
    out = np.empty(nbytes, np.uint8)
    G._check(G._hip().hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(buf.ptr), ctypes.c_size_t(nbytes), 2), "hipMemcpy")
    return out


def main():
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 17
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    shapes = [(768, 33_000, (1, 4, 100, 256)), (384, 20_011, (3, 256)), (1024, 40_000, (1, 128, 256)), (2048, 30_000, (4, 256)), (4096, 12_345, (5, 256))]
    history = {}          # virtual address -> bytes the previous tenant held (first 4096 bytes are enough to recognise it)
    events, searches, va_reuse = [], 0, 0

    def check(tag, buf, want_bytes, when):
        t8 = buf.tensor((len(want_bytes),), torch.uint8)
        wide = int(t8.unsqueeze(0).expand(256, -1).sum(dtype=torch.int64))       # read FIRST: the first many-CU access after the fill
        got_t = t8.clone().cpu().numpy()
        got_m = read_memcpy(buf, len(want_bytes))
        want_wide = 256 * int(want_bytes.astype(np.int64).sum())
        if wide != want_wide:
            events.append({"buffer": tag, "path": "torch_wide", "when": when, "sum": wide, "expected": want_wide,
                           "narrow_clone_right": bool(np.array_equal(got_t, want_bytes)), "memcpy_right": bool(np.array_equal(got_m, want_bytes))})
            print(json.dumps(events[-1]), flush=True)
        for path, got in (("torch", got_t), ("memcpy", got_m)):
            if not np.array_equal(got, want_bytes):
                off = int(np.flatnonzero(got != want_bytes)[0])
                prev = history.get(buf.ptr)
                events.append({"buffer": tag, "path": path, "when": when, "first_bad_offset": off, "bad_bytes": int((got != want_bytes).sum()),
                               "equals_previous_tenant": bool(prev is not None and off < len(prev) and np.array_equal(got[off: off + 64], prev[off: off + 64]))})
                print(json.dumps(events[-1]), flush=True)

    for c in range(cycles):
        d, n, nqs = shapes[c % len(shapes)]
        xb = int_data(rng, n, d)
        idx = FlatIndex(d, device=dev)
        host = torch.zeros((n, idx.dim), dtype=torch.float16)
        host[:, :d] = torch.from_numpy(xb).half()
        want_b = host.view(torch.uint8).reshape(-1).numpy()
        kb = Fresh(len(want_b))
        va_reuse += kb.ptr in history
        view = kb.tensor(tuple(host.shape), torch.float16)
        view.copy_(host)
        check("corpus", kb, want_b, "after fill")
        idx.adopt(view)
        for nq in nqs:
            xq = int_data(rng, nq, d)
            qh = torch.zeros((nq, idx.dim), dtype=torch.float16)
            qh[:, :d] = torch.from_numpy(xq).half()
            want_q = qh.view(torch.uint8).reshape(-1).numpy()
            kq = Fresh(len(want_q))
            va_reuse += kq.ptr in history
            qv = kq.tensor(tuple(qh.shape), torch.float16)
            qv.copy_(qh)
            check("queries", kq, want_q, "after fill")
            D, I = idx.search_prepared(qv, 10)
            torch.cuda.synchronize()
            searches += 1
            Dr, Ir = O.flat_search_ip(xb, xq, 10)
            if not (np.array_equal(I.cpu().numpy(), Ir) and np.array_equal(D.cpu().numpy(), Dr)):
                bad_q = [q for q in range(nq) if not np.array_equal(I[q].cpu().numpy(), Ir[q])]
                events.append({"buffer": "search result", "path": "rr", "cycle": c, "d": d, "n": n, "nq": nq, "bad_queries": bad_q[:8], "n_bad": len(bad_q)})
                print(json.dumps(events[-1]), flush=True)
                check("corpus", kb, want_b, "after a wrong search")       # were the inputs wrong for torch / memcpy too?
                check("queries", kq, want_q, "after a wrong search")
                D2, I2 = idx.search_prepared(qv, 10)                       # ... and does the SAME search repeat the error?
                torch.cuda.synchronize()
                events.append({"repeat_of_the_same_search_is_correct": bool(np.array_equal(I2.cpu().numpy(), Ir))})
                print(json.dumps(events[-1]), flush=True)
            history[kq.ptr] = want_q[:4096].copy()
            kq.close()
        history[kb.ptr] = want_b[:4096].copy()
        del idx, view
        kb.close()
    print(json.dumps({"cycles": cycles, "searches": searches, "buffers_at_a_recycled_virtual_address": int(va_reuse), "mismatch_events": len(events), "events": events[:40]}))


if __name__ == "__main__":
    main()
