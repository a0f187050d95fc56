"""GPU idle gaps in a rocprofv3 kernel trace (steady-state tail): how much of a step the GPU waits for the host, and
after which kernels.   python tools/gap_stats.py <kernel_trace.csv> [--window-ms 200]"""
import argparse, csv, collections
ap = argparse.ArgumentParser(); ap.add_argument("trace"); ap.add_argument("--window-ms", type=float, default=200.0); a = ap.parse_args()
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(a.trace)))
t_end = max(r[1] for r in rows); t0 = t_end - int(a.window_ms * 1e6)
rows = [r for r in rows if r[0] >= t0]
busy = sum(e - s for s, e, _ in rows)
gaps = []
last_end, last_name = rows[0][1], rows[0][2]
for s, e, n in rows[1:]:
    if s > last_end: gaps.append((s - last_end, last_name, n))
    if e > last_end: last_end, last_name = e, n
tot = sum(g[0] for g in gaps)
print("window %.0f ms: busy %.1f ms, idle %.1f ms in %d gaps" % (a.window_ms, busy / 1e6, tot / 1e6, len(gaps)))
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e6), (1e6, 1e12)):
    sel = [g for g in gaps if lo <= g[0] < hi]
    print("  gaps %6.0f-%-8.0f us: %5d, total %.2f ms" % (lo / 1e3, hi / 1e3, len(sel), sum(g[0] for g in sel) / 1e6))
print("largest gaps (us, after -> before):")
for g in sorted(gaps, reverse=True)[:25]:
    print("  %8.1f  %-70s -> %s" % (g[0] / 1e3, g[1][:70], g[2][:70]))
