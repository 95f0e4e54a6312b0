"""Diagnostic (RR_SCAN_TIMELINE=1, DEVELOPMENT library: RR_LIB_OVERRIDE=ragroute_amd/libragroute_hip_dev.so): per-workgroup start /
first-tile-ready / end times of the LAST chunk scan of a search (s_memrealtime, 100 MHz).    python tools/timeline.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RR_SCAN_TIMELINE"] = "1"
import torch
from ragroute_amd.flat_index import FlatIndex

n, d, nq, k = (int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000), 768, 256, 32
dev = torch.device("cuda:0")
idx = FlatIndex(d, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1234)
xb = torch.empty((n, d), dtype=torch.float16, device=dev)
for s in range(0, n, 1 << 20):
    e = min(n, s + (1 << 20))
    xb[s:e] = (torch.randn((e - s, d), generator=g, device=dev) / d ** 0.5).to(torch.float16)
idx.adopt(xb)
xq = torch.randn((nq, d), generator=g, device=dev).to(torch.float16)
for _ in range(4):
    idx.search_prepared(xq, k)
torch.cuda.synchronize()
ws = idx._ws[k]
grid = 256


def up(v):
    return (v + 255) // 256 * 256


# offset of the dense-sample buffer (idle during chunk scans: the timeline lands there) in capi.hip::carve's layout, dim <= 768, k = 32
off = sum(up(b) for b in (32 * 256 * 4, 32 * 4, 32 * 4, 32 * 4, 32 * 8, 256 * 4, 12 * 32 * 16, 256 * 4, 256 * 1024 * 8, 256 * grid * 4 * 4, grid * 8 * 64 * 8))
raw = ws[off: off + (3 * grid + 65 * grid) * 8].view(torch.int64).cpu().double()
t = raw[: 3 * grid].reshape(3, grid)
first_ready = raw[3 * grid + 64 * grid:] / 100.0
start, end, xcc = t[0] / 100.0, t[1] / 100.0, t[2]   # us
t0 = start.min()
print(f"launch span {(end.max() - t0).item():.1f} us; starts spread {(start.max() - t0).item():.1f} us")
dur = end - start
print(f"first tile ready after start: median {(first_ready - start).median().item():.2f} us, max {(first_ready - start).max().item():.2f} us")
print(f"workgroup durations: min {dur.min().item():.1f} median {dur.median().item():.1f} max {dur.max().item():.1f} us")
print(f"ends: first {(end.min() - t0).item():.1f} median {(end.median() - t0).item():.1f} last {(end.max() - t0).item():.1f} us  -> tail after median {(end.max() - end.median()).item():.1f} us")
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"  XCC {x}: {int(m.sum())} WGs, median end {(end[m].median() - t0).item():.1f} us, last {(end[m].max() - t0).item():.1f} us")
