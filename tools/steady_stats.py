"""Aggregate a rocprofv3 kernel trace over its steady-state tail.

    python tools/steady_stats.py <kernel_trace.csv> <out.csv> [--window-ms 300] [--header "text"] [--by-shape] [--include REGEX] [--skip-first N]

MIOpen's find pass, JIT compiles and warm-up pollute whole-run statistics; this keeps only the
dispatches that START inside the last `window` milliseconds of the trace and writes per-kernel
calls / total / average (the same columns as rocprofv3's *_kernel_stats.csv) plus VGPR and LDS use.

Round 6: `--window-ms 0` = the WHOLE trace (the operator workloads: a "last 30 ms" window had caught only the
cache-flush copies); `--by-shape` keeps launches of one kernel with different grids apart (name @ grid x block: the extractor's
direct-gather fallback calls no longer share a row with the fast path when their launch geometry differs); `--include REGEX` keeps
matching kernel names only; `--skip-first N` drops the first N launches of every row (warm-up)."""
import argparse
import csv
import collections


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("out")
    ap.add_argument("--window-ms", type=float, default=300.0)
    ap.add_argument("--header", default="")
    ap.add_argument("--end-offset-ms", type=float, default=0.0, help="end the window this long before the end of the trace")
    ap.add_argument("--by-shape", action="store_true", help="one row per (kernel, grid, workgroup) instead of per kernel")
    ap.add_argument("--include", default="", help="keep only kernels whose name matches this regular expression")
    ap.add_argument("--skip-first", type=int, default=0, help="drop the first N launches of every row (warm-up)")
    a = ap.parse_args()
    import re
    inc = re.compile(a.include) if a.include else None
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if inc is not None and not inc.search(name):
                continue
            if a.by_shape and "Grid_Size_X" in r:
                name = "%s @ grid %sx%sx%s block %s" % (name[:140], r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "1"), r.get("Grid_Size_Z", "1"),
                                                       r.get("Workgroup_Size_X", "?"))
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name,
                         r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")), r.get("LDS_Block_Size", "")))
    rows.sort()
    t_end = max(r[1] for r in rows) - int(a.end_offset_ms * 1e6)
    t0 = t_end - int(a.window_ms * 1e6) if a.window_ms > 0 else min(r[0] for r in rows)
    sel = [r for r in rows if t0 <= r[0] < t_end]
    if a.window_ms <= 0:
        a.window_ms = (t_end - t0) / 1e6
    agg = collections.OrderedDict()
    busy = 0
    seen = collections.Counter()
    for s, e, name, vg, lds in sel:
        seen[name] += 1
        if seen[name] <= a.skip_first:
            continue
        d = agg.setdefault(name, [0, 0, vg, lds])
        d[0] += 1
        d[1] += e - s
        busy += e - s
    short = sum(1 for s, e, *_ in sel if e - s < 10000)
    total = sum(v[1] for v in agg.values())
    with open(a.out, "w") as f:
        if a.header:
            f.write("# %s\n" % a.header)
        f.write("# steady-state window = last %.0f ms of the trace: %d dispatches, GPU busy %.1f ms (%.0f%% of the window), "
                "%d dispatches shorter than 10 us\n" % (a.window_ms, len(sel), busy / 1e6, 100.0 * busy / (a.window_ms * 1e6), short))
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "VGPR", "LDS"])
        for name, (calls, ns, vg, lds) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([name[:220], calls, ns, ns // calls, "%.2f" % (100.0 * ns / total), vg, lds])
    print("window %.0f ms: %d dispatches, busy %.1f ms" % (a.window_ms, len(sel), busy / 1e6))


if __name__ == "__main__":
    main()
