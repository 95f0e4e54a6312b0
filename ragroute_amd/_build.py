"""Build libragroute_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build().

Every translation unit is compiled to its own object (in parallel, only when it or a header changed), the objects are linked
into the shared library, and the device code of flat_scan_wide.hip is then disassembled and checked (check_isa): the wide-row
kernel keeps its accumulators in hard-wired AGPRs that the compiler only knows as asm clobbers, so a compiler-made
AGPR copy or a scratch spill inside that kernel would silently corrupt scores — the build fails instead."""
import concurrent.futures
import os
import re
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(PKG, "libragroute_hip.so")
SOURCES = ["capi.hip", "flat_scan.hip", "flat_scan_wide.hip", "flat_scan_dev.hip", "select.hip", "prep.hip", "router.hip", "screen.hip"]
HEADERS = ["rr_common.h", "rr_kernels.h", "rr_sort.h", "flat_scan_common.h"]
LLVM_BIN = os.environ.get("RR_LLVM_BIN", "/opt/rocm/lib/llvm/bin")

# kernels whose accumulators live in hard-wired AGPRs named only inside asm text
FIXED_AGPR_KERNELS = ("flat_scan_wide",)   # flat_scan_wide_kernel, _pd_kernel, wide8_kernel
# Scalar-register pressure tripwire for the hand-scheduled kernels: the largest .sgpr_spill_count per kernel family as of round 4
# (only the L2 four-block instances of the 4-wave kernel spill, 2 SGPRs, outside the counted-wait step).  Spilled SGPRs go to
# lanes of a compiler-owned VGPR (v_writelane / v_readlane), which is correct but costs issue slots inside a loop whose waits
# are counted by hand: a count above these is a build failure, to be looked at, not waved through.
SGPR_SPILL_MAX = {"flat_scan_wide_pd_kernel": 2, "flat_scan_wide8_kernel": 0, "flat_scan_wide_rs_kernel": 0, "flat_scan_wide_kernel": 0}


PRODUCT_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17"]   # what rr_build_flags() of a product library reports


def _flags():
    flags = list(PRODUCT_FLAGS)
    if os.environ.get("RR_DEV_VARIANTS") or os.environ.get("RR_ABLATION_VARIANTS"):
        flags.append("-DRR_DEV_VARIANTS")      # the measured alternatives of flat_scan_dev.hip (RR_SCAN_VARIANT / RR_GENERIC_TALL)
    if os.environ.get("RR_ABLATION_VARIANTS"):
        flags.append("-DRR_ABLATION_VARIANTS")  # + timing-only ablations of the 32x32x16 loop
    if os.environ.get("RR_EXTRA_DEFINES"):      # development A/B builds, e.g. "-DRR_DMA_SCHED=1"
        flags += os.environ["RR_EXTRA_DEFINES"].split()
    return flags


def _run(cmd, what):
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"{what} failed:\n{' '.join(cmd)}\n{res.stdout[-2000:]}{res.stderr[-4000:]}")
    return res


def device_code_object(obj_path, out_path):
    """Extract the gfx950 code object embedded in a HIP object file."""
    fat = out_path + ".fatbin"
    _run([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj_path], "llvm-objcopy")
    _run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", f"--input={fat}", "--type=o",
          "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out_path}"], "clang-offload-bundler")
    os.remove(fat)
    return out_path


def check_isa(obj_path):
    """Fail if a kernel with hard-wired AGPR accumulators contains anything the compiler could only have put there by
    treating those AGPRs (or scratch) as its own: v_accvgpr_write / v_accvgpr_mov (AGPR spills and copies; the kernel's own
    asm only ever READS accumulators with v_accvgpr_read), scratch_ instructions, a private segment, or spill counts."""
    co = device_code_object(obj_path, obj_path + ".gfx950.co")
    notes = _run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", co], "llvm-readelf").stdout
    problems, seen = [], 0
    for block in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name or not any(k in name.group(1) for k in FIXED_AGPR_KERNELS):
            continue
        seen += 1
        # (.sgpr_spill_count may be non-zero: scalar spills go to lanes of a VGPR the compiler owns - never to the hard-wired AGPRs,
        # and never to memory while private_segment_fixed_size stays 0 - but not above the recorded bound of the kernel family)
        m = re.search(r"\.sgpr_spill_count:\s+(\d+)", block)
        family = max((f for f in SGPR_SPILL_MAX if f in name.group(1)), key=len, default=None)
        if m and family is not None and int(m.group(1)) > SGPR_SPILL_MAX[family]:
            problems.append(f"{name.group(1)}: .sgpr_spill_count = {m.group(1)} > {SGPR_SPILL_MAX[family]} (SGPR_SPILL_MAX: scalar pressure grew in a hand-scheduled kernel)")
        for key in (".private_segment_fixed_size", ".vgpr_spill_count"):
            m = re.search(re.escape(key) + r":\s+(\d+)", block)
            if m and int(m.group(1)) != 0:
                problems.append(f"{name.group(1)}: {key} = {m.group(1)}")
    dis = _run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", co], "llvm-objdump").stdout
    current = None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            current = m.group(1) if any(k in m.group(1) for k in FIXED_AGPR_KERNELS) else None
            continue
        if current and re.search(r"\b(v_accvgpr_write|v_accvgpr_mov|scratch_load|scratch_store)", line):
            problems.append(f"{current}: {line.strip()}")
    problems += check_prefetch_registers(dis)
    os.remove(co)
    if seen == 0:
        problems.append("no kernel named " + " / ".join(FIXED_AGPR_KERNELS) + " found in the device code")
    if problems:
        raise RuntimeError("ISA check of the fixed-AGPR kernels failed (the compiler used AGPRs or scratch of its own):\n  " +
                           "\n  ".join(problems[:40]))
    return seen


# kernels whose query prefetch (inline-asm global_load_dwordx4 with an SGPR base) stays in flight across loop iterations
PREFETCH_KERNELS = ("flat_scan_wide_pd_kernel", "flat_scan_wide8_kernel")
_QLOAD = re.compile(r"^\s*global_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[")
_VDEST = re.compile(r"^\s*(?:v_(?!cmp|mfma|accvgpr_write)\w+|ds_read\w*|global_load\w*|buffer_load\w*)\s+v(?:(\d+)\b|\[(\d+):(\d+)\])")


def check_prefetch_registers(dis):
    """The prefetched query loads land asynchronously in registers the compiler only sees as asm operands.  Anything else
    that writes one of those registers is safe only after the loop, behind an s_waitcnt vmcnt(0): a tail that computed store
    addresses in them while the last prefetch was still landing produced wild stores (round 2, d = 2048, few queries)."""
    problems, current, body = [], None, []

    def finish():
        if not current or not body:
            return
        loads = [(i, range(int(m.group(1)), int(m.group(2)) + 1)) for i, l in enumerate(body) for m in [_QLOAD.match(l)] if m]
        if not loads:
            problems.append(f"{current}: no asm query loads found (pattern out of date?)")
            return
        qregs = {r for _, rs in loads for r in rs}
        first, last = loads[0][0], loads[-1][0]
        for i, l in enumerate(body):
            m = _VDEST.match(l)
            if not m or i <= first or _QLOAD.match(l):
                continue
            regs = {int(m.group(1))} if m.group(1) else set(range(int(m.group(2)), int(m.group(3)) + 1))
            if not regs & qregs:
                continue
            drained, j = False, i - 1          # a drain earlier in the same straight-line run (back to the previous branch)
            while i > last and j > last and not re.match(r"^\s*(s_branch|s_cbranch|s_endpgm|s_setpc)", body[j]):
                if re.match(r"^\s*s_waitcnt vmcnt\(0\)", body[j]):
                    drained = True
                    break
                j -= 1
            if not drained:
                problems.append(f"{current}: query-prefetch register written while loads may be in flight: {l.strip()}")

    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            finish()
            current = m.group(1) if any(k in m.group(1) for k in PREFETCH_KERNELS) else None
            body = []
        elif current:
            body.append(line.split("//")[0])
    finish()
    return problems


def build(force=False, verbose=False):
    suffix = os.environ.get("RR_LIB_SUFFIX", "")
    lib_path = LIB_PATH.replace(".so", suffix + ".so") if suffix else LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = _flags()
    # objects of a suffixed (development / A/B) library live in their own directory: "flat_scan" + "_dev" must not meet the
    # product object of flat_scan_dev.hip
    obj_dir = os.path.join(OBJ_DIR, suffix.strip("_")) if suffix else OBJ_DIR
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(PKG, "..", "include", "ragroute_hip.h")]
    hdr_time = max(os.path.getmtime(h) for h in headers)
    stamp = os.path.join(obj_dir, "flags.txt")
    flags_changed = not os.path.exists(stamp) or open(stamp).read() != " ".join(flags)
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, s.replace(".hip", ".o"))
        stale = force or flags_changed or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time)
        jobs.append((src, obj, stale))
    todo = [(src, obj) for src, obj, stale in jobs if stale]
    if not todo and os.path.exists(lib_path) and all(os.path.getmtime(lib_path) >= os.path.getmtime(o) for _, o, _ in jobs):
        return lib_path

    # rr_build_flags(): capi.hip includes the flag string as a C literal
    inc = os.path.join(obj_dir, "rr_build_flags.inc")
    literal = '"' + " ".join(flags).replace("\\", "\\\\").replace('"', '\\"') + '"\n'
    if not os.path.exists(inc) or open(inc).read() != literal:
        open(inc, "w").write(literal)

    def compile_one(job):
        src, obj = job
        res = _run([hipcc] + flags + [f"-I{obj_dir}", "-c", src, "-o", obj], f"hipcc -c {os.path.basename(src)}")
        return res.stderr

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as pool:
        for err in pool.map(compile_one, todo):
            if verbose and err:
                print(err)
    open(stamp, "w").write(" ".join(flags))
    if not os.environ.get("RR_SKIP_ISA_CHECK"):   # diagnostic builds only
        check_isa(os.path.join(obj_dir, "flat_scan_wide.o"))
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + [o for _, o, _ in jobs], "hipcc -shared")
    return lib_path


def build_asan_host(out_dir=None):
    """an AddressSanitizer build of the binding's HOST code (capi.hip: argument validation, workspace carving, the
    chunk / segment schedules) linked with the ordinary kernel objects - for the CPU argument-validation tests only (no GPU ASan
    on this pool).  Returns (library path, path of the ASan runtime to LD_PRELOAD)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    build()                                                   # the ordinary objects
    out_dir = out_dir or OBJ_DIR
    obj = os.path.join(out_dir, "capi_asan.o")
    lib_path = os.path.join(out_dir, "libragroute_hip_asan.so")
    _run([hipcc] + PRODUCT_FLAGS + ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan",
                                    f"-I{OBJ_DIR}", "-c", os.path.join(CSRC, "capi.hip"), "-o", obj], "hipcc -fsanitize=address -c capi.hip")
    others = [os.path.join(OBJ_DIR, s.replace(".hip", ".o")) for s in SOURCES if s != "capi.hip"]
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-shared-libsan", "-o", lib_path, obj] + others,
         "hipcc -shared (asan)")
    rt = _run([os.path.join(LLVM_BIN, "clang"), "-print-file-name=libclang_rt.asan-x86_64.so"], "clang -print-file-name").stdout.strip()
    return lib_path, rt
