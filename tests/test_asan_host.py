"""CPU: the C ABI's host code (capi.hip: argument validation, workspace carving, chunk and segment schedules) under
AddressSanitizer (the kernels keep their ordinary objects - there is no GPU ASan on this pool).  The calls are the ones
that return before any device work, so they run without a GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CALLS = r"""
import ctypes, os, sys
sys.path.insert(0, %r)
from ragroute_amd import _lib
L = _lib.lib()
S = _lib.SegmentStruct
assert L.rr_flat_search(None, 0, 10, 768, None, 1, 0, None, None, 0, None, 0, None, 0, None) == -1
assert L.rr_flat_search(None, 0, 10, 700, None, 1, 5, None, None, 0, None, 0, None, 0, None) == -2
assert L.rr_flat_search(None, 0, 10, 768, None, 0, 5, None, None, 0, None, 0, None, 0, None) == 0
assert L.rr_flat_search(None, 0, 1 << 33, 768, None, 1, 5, None, None, 0, None, 0, None, 0, None) == -2
for n_segs in (1, 2, 32):
    segs = (S * n_segs)(*[S(256 * i, 100, i << 40, i, 0) for i in range(n_segs)])
    total = 256 * (n_segs - 1) + 100
    assert L.rr_flat_search_segments(None, 0, total, 768, segs, n_segs, None, 0, 5, None, None, None, 0, None, 64, None) == 0      # nq = 0
    assert L.rr_flat_search_segments(None, 0, total, 768, segs, n_segs, None, 1, 5, None, None, None, 0, None, 64, None) == -1     # null pointers
    assert L.rr_flat_search_segments(None, 0, total - 1, 768, segs, n_segs, None, 1, 5, None, None, None, 0, None, 64, None) == -1 # past the matrix
segs = (S * 2)(S(0, 300, 0, 0, 0), S(256, 10, 0, 1, 0))                                                                            # overlap
assert L.rr_flat_search_segments(None, 0, 1000, 768, segs, 2, None, 1, 5, None, None, None, 0, None, 64, None) == -1
assert L.rr_flat_search_segments(None, 0, 1000, 768, segs, 33, None, 1, 5, None, None, None, 0, None, 64, None) == -1
segs = (S * 1)(S(256, (1 << 63) - 1, 0, 0, 0))                                  # row_begin + n_rows would overflow int64
assert L.rr_flat_search_segments(None, 0, 1000, 768, segs, 1, None, 1, 5, None, None, None, 0, None, 64, None) == -1
segs = (S * 1)(S(1 << 62, 10, 0, 0, 0))                                          # row_begin beyond the matrix
assert L.rr_flat_search_segments(None, 0, 1000, 768, segs, 1, None, 1, 5, None, None, None, 0, None, 64, None) == -1
assert L.rr_flat_search_segments(None, 0, 1000, 768, None, 1, None, 1, 5, None, None, None, 0, None, 64, None) == -1
assert L.rr_merge_topk(None, None, 1, 9000, 5, 1, None, None, None) == -2
assert L.rr_merge_topk_gathered(None, 8, 256 * 32 * 12, 256 * 32 * 4, 1, 256, 32, 32, 1, None, None, None) == -1
assert L.rr_merge_topk_gathered(None, 100, 256 * 100 * 12, 256 * 100 * 4, 1, 256, 100, 10, 1, None, None, None) == -2
assert L.rr_rows_to_half(None, 1, 8, 4, None, 0, 8, 0, None) == -1
assert L.rr_router_mlp(None, None, 1, None, None, None) == -1
assert L.rr_flat_search_workspace_bytes(0) == 0 and L.rr_screen_dim(1024) == 1024 and L.rr_screen_dim(4096) == -2
assert L.rr_profile_end(None, None, None) == -1 and L.rr_profile_begin(0) == -1
assert L.rr_flat_scan_kernel_name(768, 256) == b"flat_scan16_kernel" and L.rr_flat_scan_kernel_name(1000, 5) == b""
assert b"-O3" in L.rr_build_flags() or b"-O1" in L.rr_build_flags()
for d in range(1, 9000, 37):
    L.rr_padded_dim(d)
print("asan calls ok")
"""


def test_host_code_of_the_c_abi_under_address_sanitizer(tmp_path):
    from ragroute_amd import _build
    try:
        lib, rt = _build.build_asan_host(str(tmp_path))
    except RuntimeError as e:
        pytest.skip(f"no AddressSanitizer toolchain for the host pass: {str(e)[:200]}")
    if not os.path.exists(rt):
        pytest.skip("ASan runtime not found")
    env = dict(os.environ, RR_LIB_OVERRIDE=lib, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99", HIP_VISIBLE_DEVICES="-1")
    res = subprocess.run([sys.executable, "-c", CALLS % ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "asan calls ok" in res.stdout, (res.returncode, res.stdout[-500:], res.stderr[-3000:])
    assert "AddressSanitizer" not in res.stderr, res.stderr[-3000:]
