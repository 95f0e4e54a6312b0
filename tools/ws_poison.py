"""Does any search depend on what the workspace held before?  Fill the index workspace with a byte pattern before each search."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from oracle import oracle as O
from ragroute_amd.flat_index import FlatIndex
from tests.util import int_data
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
bad = 0
for d, n in [(768, 20_000), (1024, 20_000), (2048, 12_345), (4096, 12_345), (4096, 3_000)]:
    xb = int_data(rng, n, d)
    idx = FlatIndex(d, device=dev); idx.add(xb)
    for nq in (1, 5, 16, 17, 64, 100, 256):
        xq = int_data(rng, nq, d)
        q = idx.prepare_queries(xq)
        Dr, Ir = O.flat_search_ip(xb, xq, 10)
        for pat in (0xFF, 0x00, 0x3C, 0x7B):
            idx._workspace(10).fill_(pat)
            D, I = idx.search_prepared(q, 10); torch.cuda.synchronize()
            ok = np.array_equal(I.cpu().numpy(), Ir) and np.array_equal(D.cpu().numpy(), Dr)
            if not ok:
                bad += 1
                print("MISMATCH d", d, "n", n, "nq", nq, "pattern", hex(pat), flush=True)
print("mismatches", bad)
