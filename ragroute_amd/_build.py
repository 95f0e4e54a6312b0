"""Build libragroute_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build()."""
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libragroute_hip.so")
SOURCES = ["capi.hip", "flat_scan.hip", "flat_scan_dev.hip", "select.hip", "prep.hip", "router.hip", "screen.hip"]


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in ("rr_common.h", "rr_kernels.h", "rr_sort.h", "flat_scan_common.h")] + [
        os.path.join(os.path.dirname(CSRC), "..", "include", "ragroute_hip.h")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-o", LIB_PATH] + srcs
    if os.environ.get("RR_DEV_VARIANTS") or os.environ.get("RR_ABLATION_VARIANTS"):
        cmd.insert(1, "-DRR_DEV_VARIANTS")      # the measured alternatives of flat_scan_dev.hip (RR_SCAN_VARIANT / RR_GENERIC_TALL)
    if os.environ.get("RR_ABLATION_VARIANTS"):
        cmd.insert(1, "-DRR_ABLATION_VARIANTS")  # + timing-only ablations of the 32x32x16 loop
    if os.environ.get("RR_EXTRA_DEFINES"):  # development A/B builds, e.g. "-DRR_DMA_SCHED=1"
        cmd[1:1] = os.environ["RR_EXTRA_DEFINES"].split()
    if os.environ.get("RR_LIB_SUFFIX"):
        cmd[cmd.index("-o") + 1] = LIB_PATH.replace(".so", os.environ["RR_LIB_SUFFIX"] + ".so")
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout, res.stderr)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed building libragroute_hip.so:\n" + res.stderr[-4000:])
    return LIB_PATH
