"""Diagnostic (library built with RR_DEV_VARIANTS=1, RR_SCAN_VARIANT=2): share of the scan loop's cycles per segment, from in-kernel s_memtime stamps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RR_SCAN_VARIANT"] = "2"
import torch
from ragroute_amd.flat_index import FlatIndex

n, d, nq, k = 10_000_000, 768, 256, 32
dev = torch.device("cuda:0")
idx = FlatIndex(d, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1234)
xb = torch.empty((n, d), dtype=torch.float16, device=dev)
for s in range(0, n, 1 << 20):
    e = min(n, s + (1 << 20))
    xb[s:e] = (torch.randn((e - s, d), generator=g, device=dev) / d ** 0.5).to(torch.float16)
idx.adopt(xb)
xq = torch.randn((nq, d), generator=g, device=dev).to(torch.float16)
for _ in range(3):
    idx.search_prepared(xq, k)
torch.cuda.synchronize()
ws = idx._ws[k]
grid = 256
off = 1024 + 1024 + 256 * 1024 * 8 + 256 * grid * 4 * 4 + grid * 8 * 64 * 8
dbg = ws[off: off + grid * 4 * 6 * 8].view(torch.int64).reshape(grid, 4, 6).cpu().double()
clk = dbg[:, :, 4] / dbg[:, :, 5] * 100e6
print(f'in-kernel clock: median {clk.median().item()/1e9:.3f} GHz (min {clk.min().item()/1e9:.3f}, max {clk.max().item()/1e9:.3f}); loop {dbg[:,:,5].mean().item()/100:.1f} us')
dbg = dbg[:, :, :4]
steps = (n // 32 - 4194304 // 32) / grid
names = ["wait DMA (vmcnt)", "barrier", "MFMA stream (+LDS reads, DMA issue)", "epilogue + loop"]
tot = dbg.sum(-1).mean().item()
print(f"last chunk: {steps:.0f} tiles per workgroup, {tot / steps:.0f} cycles per tile (stamps included)")
for i, nm in enumerate(names):
    v = dbg[:, :, i]
    print(f"  {nm:40s} {v.mean().item() / steps:8.0f} cyc/tile  {100 * v.mean().item() / tot:5.1f} %   (min wave {v.min().item() / steps:.0f}, max wave {v.max().item() / steps:.0f})")
