"""CPU: the C-ABI library loads and exports every symbol include/ragroute_hip.h declares; argument
validation that needs no GPU returns the documented status codes (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "ragroute_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rr_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    from ragroute_amd import _lib
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/ragroute_hip.h but not exported"
    assert set(names) == set(_lib.EXPORTS)


def test_version_and_padded_dim():
    from ragroute_amd import _lib
    L = _lib.lib()
    assert L.rr_version() >= 100
    assert [L.rr_padded_dim(d) for d in (1, 100, 128, 129, 768)] == [128, 128, 128, 256, 768]
    assert [L.rr_padded_dim(d) for d in (769, 1024, 1100, 1537, 4096)] == [896, 1024, 1280, 1664, 4096]  # above 1536: multiples of 128 (wide-row kernel)
    assert L.rr_padded_dim(8193) == -2 and L.rr_padded_dim(0) == -1


def test_argument_validation_without_gpu():
    from ragroute_amd import _lib
    L = _lib.lib()
    assert L.rr_flat_search(None, 0, 10, 768, None, 1, 0, None, None, 0, None, 0, None, 0, None) == -1      # k = 0
    assert b"k must be" in L.rr_last_error()
    assert L.rr_flat_search(None, 0, 10, 768, None, 1, 2000, None, None, 0, None, 0, None, 0, None) == -1   # k > 1024
    assert L.rr_flat_search(None, 7, 10, 768, None, 1, 5, None, None, 0, None, 0, None, 0, None) == -1      # dtype
    assert L.rr_flat_search(None, 0, 10, 700, None, 1, 5, None, None, 0, None, 0, None, 0, None) == -2      # unpadded dim
    assert L.rr_flat_search(None, 0, 10, 1000, None, 1, 5, None, None, 0, None, 0, None, 0, None) == -2
    assert L.rr_flat_search(None, 0, 10, 768, None, 0, 5, None, None, 0, None, 0, None, 0, None) == 0       # nq = 0 is a no-op
    assert L.rr_merge_topk(None, None, 1, 9000, 5, 1, None, None, None) == -2
    assert L.rr_merge_topk(None, None, 0, 5, 5, 1, None, None, None) == 0
    # rr_merge_topk_gathered: [rank][D f32[slots][nq][k_in] | pad | I i64[slots][nq][k_in]]
    assert L.rr_merge_topk_gathered(None, 8, 256 * 32 * 12, 256 * 32 * 4, 1, 256, 32, 32, 1, None, None, None) == -1     # null buffer
    assert L.rr_merge_topk_gathered(None, 8, 256 * 32 * 12, 256 * 32 * 4, 1, 0, 32, 32, 1, None, None, None) == 0        # nq = 0
    assert L.rr_merge_topk_gathered(None, 8, 256 * 32 * 12 + 4, 256 * 32 * 4, 1, 256, 32, 32, 1, None, None, None) == -1  # stride not 8-aligned
    assert L.rr_merge_topk_gathered(None, 8, 256 * 32 * 12, 256 * 32 * 2, 1, 256, 32, 32, 1, None, None, None) == -1     # ids overlap the scores
    assert L.rr_merge_topk_gathered(None, 8, 256 * 32 * 8, 256 * 32 * 4, 1, 256, 32, 32, 1, None, None, None) == -1      # rank block too short
    assert L.rr_merge_topk_gathered(None, 100, 256 * 100 * 12, 256 * 100 * 4, 1, 256, 100, 10, 1, None, None, None) == -2  # > 8192 candidates
    assert L.rr_rows_to_half(None, 1, 8, 4, None, 0, 8, 0, None) == -1
    assert L.rr_l2_normalize_f32(None, 0, 8, None) == 0
    assert L.rr_router_mlp(None, None, 1, None, None, None) == -1
    assert L.rr_flat_search_workspace_bytes(0) == 0


def test_segment_row_counts_cannot_overflow():
    from ragroute_amd import _lib
    L, S = _lib.lib(), _lib.SegmentStruct
    for seg in (S(256, (1 << 63) - 1, 0, 0, 0), S(1 << 62, 10, 0, 0, 0), S(0, 1001, 0, 0, 0)):
        assert L.rr_flat_search_segments(None, 0, 1000, 768, (S * 1)(seg), 1, None, 1, 5, None, None, None, 0, None, 64, None) == -1
        assert b"inside the matrix" in L.rr_last_error()


def test_product_library_ignores_tuning_variables():
    """RR_WIDE_* / RR_SCAN_VARIANT / RR_CHUNK_GROWTH re-route kernels and schedules in DEVELOPMENT builds only (rr::tuning_env):
    with every one of them set, a product library names the same kernels as without."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\nfrom ragroute_amd import _lib\nL = _lib.lib()\n"
            "print([L.rr_flat_scan_kernel_name(d, q).decode() for d in (768, 1024, 2048, 4096) for q in (1, 100, 160, 208, 256)])" % ROOT)
    base = {k: v for k, v in os.environ.items() if not k.startswith("RR_")}
    tuned = dict(base, RR_WIDE_RS="0", RR_WIDE_RS_MAXD="1024", RR_WIDE_RS_MINQ="1", RR_WIDE_WAVES="4", RR_WIDE_PD="0", RR_CHUNK_GROWTH="2",
                 RR_SCAN_VARIANT="15", RR_GENERIC_TALL="4", RR_WIDE_MIN_QUERIES="1", RR_SAMPLE_ROWS="1024", RR_SCAN_TIMELINE="1")
    plain = subprocess.run([sys.executable, "-c", code], env=base, capture_output=True, text=True, timeout=120)
    other = subprocess.run([sys.executable, "-c", code], env=tuned, capture_output=True, text=True, timeout=120)
    assert plain.returncode == 0 and other.returncode == 0, (plain.stderr[-500:], other.stderr[-500:])
    assert plain.stdout == other.stdout and "flat_scan_wide_rs_kernel" in plain.stdout
    src = "".join(open(os.path.join(ROOT, "ragroute_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "ragroute_amd", "csrc"))
                  if f.endswith((".hip", ".h")) and f != "flat_scan_dev.hip")
    assert len(re.findall(r"\bgetenv\(", src)) == 1, "every RR_* variable goes through rr::tuning_env (rr_common.h)"


def test_library_under_test_is_a_product_build():
    """rr_build_flags(): no -D switch, i.e. neither the development kernels nor the timing-only ablations (which produce wrong
    scores) are compiled in; RR_LIB_OVERRIDE (A/B libraries) must not be set when the suite runs."""
    from ragroute_amd import _build, _lib
    assert not os.environ.get("RR_LIB_OVERRIDE")
    flags = _lib.lib().rr_build_flags().decode()
    assert flags == " ".join(_build.PRODUCT_FLAGS), flags
    assert "-D" not in flags


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ragroute_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.lib()
        raise AssertionError("expected RagrouteHipError")
    except _lib.RagrouteHipError as e:
        assert "no CPU fallback" in str(e)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "ragroute_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().lower(), f


def test_isa_check_flags_a_prefetch_register_reused_before_the_drain():
    """The build refuses a wide-row kernel whose tail writes a query-prefetch register before s_waitcnt vmcnt(0)
    (the cause of intermittent wild stores in round 2), and accepts the same tail behind the wait."""
    from ragroute_amd import _build
    head = ["0000000000001000 <_ZN2rr24flat_scan_wide_pd_kernelIfLb0ELb0ELi1ELi2EEEvNS_8ScanArgsEi>:",
            "\tglobal_load_dwordx4 v[2:5], v41, s[4:5]",
            "\tglobal_load_dwordx4 v[6:9], v41, s[4:5] offset:64",
            "\tv_mfma_f32_16x16x32_f16 a[0:3], v[60:63], v[2:5], a[0:3]",
            "\ts_cbranch_scc1 65000"]
    bad = head + ["\tv_add_u32_e32 v2, v1, v40", "\ts_waitcnt vmcnt(0)", "\tglobal_store_dword v[2:3], v54, off", "\ts_endpgm"]
    good = head + ["\ts_waitcnt vmcnt(0)", "\tv_add_u32_e32 v2, v1, v40", "\tglobal_store_dword v[2:3], v54, off", "\ts_endpgm"]
    in_loop = head[:3] + ["\tv_mov_b32_e32 v6, v10"] + head[3:] + ["\ts_endpgm"]
    assert len(_build.check_prefetch_registers("\n".join(bad))) == 1
    assert _build.check_prefetch_registers("\n".join(good)) == []
    assert len(_build.check_prefetch_registers("\n".join(in_loop))) == 1
    other = "\n".join(bad).replace("flat_scan_wide_pd_kernel", "flat_scan16_kernel__")
    assert _build.check_prefetch_registers(other) == []
