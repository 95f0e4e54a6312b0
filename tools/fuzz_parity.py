"""GPU: tests/fuzz_cases.py for COUNT cases from SEED (development; the bounded sample is tests/test_fuzz_gpu.py).

    python tools/fuzz_parity.py [SEED] [COUNT] [big|segments]

big: 70 K ... 2 M-row corpora; segments: random segment tables + route masks for rr_flat_search_segments (oracle chain).

Prints every case before it runs (so a fault names its case) and a summary line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tests.fuzz_cases import cases, run_case, run_segment_case, segment_cases


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda:0")
    bad, t0 = [], time.time()
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    segs = len(sys.argv) > 3 and sys.argv[3] == "segments"
    todo = segment_cases(seed, count) if segs else cases(seed, count, max_work=3e10 if big else 6e9, big=big)
    for i, c in enumerate(todo):
        print(i, json.dumps(c), flush=True)
        if not (run_segment_case if segs else run_case)(c, dev):
            bad.append(c)
            print("MISMATCH", json.dumps(c), flush=True)
    print(json.dumps({"seed": seed, "cases": count, "mismatches": len(bad), "seconds": round(time.time() - t0, 1), "bad": bad}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
