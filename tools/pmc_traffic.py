"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into profiles/traffic.json.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps_in_run> <out.json> [rows dim batch k dtype lib_version]

HBM bytes per launch of the scan kernel, as MI355X_MICROARCH.md prescribes: separate --pmc passes; FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of a wide coalesced stream, so it is doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores."""
import csv
import json
import sys


def per_dispatch(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if "flat_scan" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            out[int(r["Dispatch_Id"])] = out.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return out


def main():
    fetch_csv, write_csv, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f = per_dispatch(fetch_csv, "FETCH_SIZE")
    w = per_dispatch(write_csv, "WRITE_SIZE")
    n = len(f)
    fetch_bytes = sum(f.values()) * 1024 * 2      # gfx950: FETCH_SIZE = 1/2 of the streamed bytes
    write_bytes = sum(w.values()) * 1024
    extra = sys.argv[5:]
    rows, dim, batch, k = (int(x) for x in (extra + ["10000000", "768", "256", "32"])[:4]) if len(extra) < 4 else (int(x) for x in extra[:4])
    dtype = extra[4] if len(extra) > 4 else "fp16"
    lib_version = int(extra[5]) if len(extra) > 5 else None
    res = {"rows": rows, "dim": dim, "batch": batch, "k": k, "dtype": dtype, "lib_version": lib_version,
           "scan_launches": n, "steps_profiled": steps,
           "fetch_bytes_per_launch_corrected": fetch_bytes / n, "write_bytes_per_launch": write_bytes / max(1, len(w)),
           "hbm_bytes_per_launch": fetch_bytes / n + write_bytes / max(1, len(w)),
           "hbm_bytes_per_search": (fetch_bytes / n + write_bytes / max(1, len(w))) * n / steps,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over bench.py; KiB units; FETCH_SIZE x2 (gfx950)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
