"""Build profiles/r02 from what tools/r02_profiles.sh left under gpurun_out/r02_prof (run here, after the GPU call).

    python tools/r02_collect.py [gpurun_out/r02_prof] [profiles/r02]

Copies the bench / shape lines and the rocprofv3 kernel-stats CSVs, derives the PMC summaries (FETCH_SIZE in KiB and x2 on
gfx950, WRITE_SIZE in KiB, as MI355X_MICROARCH.md prescribes; SQ counters of the largest scan launch), refreshes
profiles/traffic.json and prints the shape table of profiles/README.md."""
import csv
import json
import os
import shutil
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r02_prof"
DST = sys.argv[2] if len(sys.argv) > 2 else "profiles/r02"


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def counters(path):
    """{dispatch: {"kernel": name, counter: value summed over the XCD rows}} for the scan kernels of a --pmc pass."""
    out = {}
    for r in csv.DictReader(open(path)):
        if "flat_scan" not in r["Kernel_Name"]:
            continue
        d = out.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"]})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return out


def sq_summary(path, label=None):
    disp = counters(path)
    if not disp:
        return None
    big = max(disp.values(), key=lambda d: d.get("SQ_WAVE_CYCLES", 0.0))
    wc = big["SQ_WAVE_CYCLES"]
    res = {"kernel": label or big["kernel"],
           "SQ_WAVE_CYCLES_sum": wc,
           "SQ_WAIT_ANY_share": round(big.get("SQ_WAIT_ANY", 0.0) / wc, 4),
           "SQ_WAIT_INST_ANY_share": round(big.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
           "SQ_ACTIVE_INST_ANY_share": round(big.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4),
           # SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD; the launch lasted GRBM_GUI_ACTIVE / 8 XCDs cycles on 256 CUs x 4 SIMDs
           "MFMA_busy_share_of_SIMD_time": round(big.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1.0, big.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 256 * 4), 4),
           "SQ_LDS_BANK_CONFLICT": big.get("SQ_LDS_BANK_CONFLICT", 0.0),
           "GRBM_GUI_ACTIVE_per_XCD": big.get("GRBM_GUI_ACTIVE", 0.0) / 8.0}
    return res


def copy(name, to=None):
    src = os.path.join(SRC, name)
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(DST, to or name))
        return True
    print("missing", src)
    return False


def main():
    os.makedirs(DST, exist_ok=True)
    copy("bench_n1.json")
    copy("bench_n1_under_rocprof.json")
    copy("bench_trace/t_kernel_stats.csv", "bench_n1_kernel_stats.csv")
    copy("config5_80M_bf16_k100.json")
    if copy("traffic.json", "traffic_10M_768.json"):
        shutil.copyfile(os.path.join(SRC, "traffic.json"), "profiles/traffic.json")
    shapes = ["1000000_768", "4000000_1024", "2000000_4096", "2000000_2048", "10000000_768_1", "4000000_1024_1", "2000000_4096_1"]
    for tag in shapes:
        copy(f"shape_{tag}.json")
        if os.path.exists(os.path.join(SRC, f"trace_{tag}", "t_kernel_stats.csv")):
            copy(f"trace_{tag}/t_kernel_stats.csv", f"shape_{tag}_kernel_stats.csv")
            copy(f"shape_{tag}_under_rocprof.json")

    s16 = os.path.join(SRC, "pmc_sq", "t_counter_collection.csv")
    if os.path.exists(s16):
        res = sq_summary(s16, "flat_scan16_kernel<f16,768,filter,nt>, largest chunk launch of a 10M-row search")
        json.dump(res, open(os.path.join(DST, "scan16_pmc_summary.json"), "w"), indent=1)

    wide = {}
    for tag in ("4000000_1024", "2000000_4096"):
        rows, dim = (int(x) for x in tag.split("_"))
        f = os.path.join(SRC, f"wide_fetch_{tag}", "t_counter_collection.csv")
        if not os.path.exists(f):
            continue
        fetch = counters(f)
        write = counters(os.path.join(SRC, f"wide_write_{tag}", "t_counter_collection.csv"))
        l2 = counters(os.path.join(SRC, f"wide_l2_{tag}", "t_counter_collection.csv"))
        searches = 13  # shape_bench.py ... 4: 5 warm-up + 4 searches with per-search events + 4 back to back
        entry = {"kernel": max(fetch.values(), key=lambda d: d["FETCH_SIZE"])["kernel"],
                 "scan_launches_profiled": len(fetch),
                 "hbm_fetch_bytes_per_search_corrected": sum(d["FETCH_SIZE"] for d in fetch.values()) * 1024 * 2 / searches,
                 "hbm_write_bytes_per_search": sum(d["WRITE_SIZE"] for d in write.values()) * 1024 / searches,
                 "algorithmic_corpus_bytes_per_search": rows * dim * 2}
        entry["fetch_over_algorithmic"] = round(entry["hbm_fetch_bytes_per_search_corrected"] / entry["algorithmic_corpus_bytes_per_search"], 4)
        hit = sum(d.get("TCC_HIT_sum", 0.0) for d in l2.values())
        miss = sum(d.get("TCC_MISS_sum", 0.0) for d in l2.values())
        entry.update({"L2_hit_requests": hit, "L2_miss_requests": miss, "L2_hit_rate": round(hit / max(1.0, hit + miss), 4)})
        sq = os.path.join(SRC, f"wide_sq_{tag}", "t_counter_collection.csv")
        if os.path.exists(sq):
            entry["largest_launch"] = sq_summary(sq)
        wide[tag] = entry
    if wide:
        json.dump(wide, open(os.path.join(DST, "wide_pmc_summary.json"), "w"), indent=1)

    print("| shape | scan launches | whole search (event pair per search) | back to back |\n|---|---|---|---|")
    b = last_json(os.path.join(DST, "bench_n1.json"))
    r = b["roofline"]
    print(f"| 10M x 768, 256 queries (bench.py) | {r['achieved'] / 1000:.2f} TB/s = {r['frac']:.3f} | {b['ms_per_step']:.2f} ms, {r['end_to_end_frac']:.3f} |")
    for tag in shapes:
        p = os.path.join(DST, f"shape_{tag}.json")
        if not os.path.exists(p):
            continue
        j = last_json(p)
        r = j["roofline"]
        print(f"| {tag} | {r['achieved'] / 1000:.2f} TB/s = {r['frac']:.3f} | {j['median_ms']:.3f} ms, {j.get('end_to_end_frac_of_8TBps')} | "
              f"{j.get('back_to_back_ms')} ms, {j.get('back_to_back_frac_of_8TBps')} |")
    c5 = os.path.join(DST, "config5_80M_bf16_k100.json")
    if os.path.exists(c5):
        j = last_json(c5)
        print(f"| 80M x 768 bf16, k = 100 | - | {j['ms_per_batch']} ms, {j['hbm_frac_of_8TBs']} |")


if __name__ == "__main__":
    main()
