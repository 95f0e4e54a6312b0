"""Per-search kernel timeline from a rocprofv3 --kernel-trace CSV of tools/shape_bench.py: for the searches of the back-to-back block,
the median duration of every kernel in launch order and the median gap in front of it (start - previous end), i.e. where one search's
time goes beyond its kernels' own durations.

    python tools/trace_gaps.py <kernel_trace.csv> <kernels per search> [searches to use]"""
import csv
import json
import statistics
import sys


def main():
    path, per = sys.argv[1], int(sys.argv[2])
    use = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    rows = [r for r in csv.DictReader(open(path)) if "rr::" in r["Kernel_Name"] or "_ZN2rr" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-per * use:]                                   # the last `use` searches = the back-to-back block
    names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rr::", "")[:48] for r in rows[:per]]
    dur = [[] for _ in range(per)]
    gap = [[] for _ in range(per)]
    span = []
    for s in range(use):
        blk = rows[s * per: (s + 1) * per]
        assert [r["Kernel_Name"] for r in blk] == [r["Kernel_Name"] for r in rows[:per]], "kernel sequence differs between searches: wrong `kernels per search`"
        for i, r in enumerate(blk):
            dur[i].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            prev_end = int(blk[i - 1]["End_Timestamp"]) if i else (int(rows[s * per - 1]["End_Timestamp"]) if s else None)
            if prev_end is not None:
                gap[i].append((int(r["Start_Timestamp"]) - prev_end) / 1e3)
        if s:
            span.append((int(blk[-1]["End_Timestamp"]) - int(rows[s * per - 1]["End_Timestamp"])) / 1e3)
    out = {"searches": use, "kernels_per_search": per, "search_span_us_median": round(statistics.median(span), 2),
           "sum_of_kernel_durations_us": round(sum(statistics.median(d) for d in dur), 2),
           "sum_of_gaps_us": round(sum(statistics.median(g) for g in gap if g), 2),
           "timeline": [{"kernel": n, "duration_us": round(statistics.median(d), 2), "gap_before_us": round(statistics.median(g), 2) if g else None}
                        for n, d, g in zip(names, dur, gap)]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
