"""Per-STREAM accounting of a rocprofv3 kernel trace of `bench.py` (train workload).  The captured step runs on several HIP streams
(main: flowNetF, netG, VGG19 at 128 px, the generator's backward and Adam; side: flowNetB + VGG19 64 / 32 px; the D step + LightCNN):
the step is as long as its critical stream, so what pays is what shortens THAT stream.  Steps are cut like tools/step_trace.py (every
third adam_flat launch); per queue / stream the tool prints launches, summed kernel time, the short-kernel share and the kernels
with the most time.

    python tools/stream_trace.py <kernel_trace.csv> [--steps 3] [--top 25] [--key Queue_Id|Stream_Id]"""
import argparse, collections, csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--key", default="")
    a = ap.parse_args()
    raw = list(csv.DictReader(open(a.trace)))
    cols = raw[0].keys()
    key = a.key or ("Stream_Id" if "Stream_Id" in cols else "Queue_Id")
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get(key, "?"), r.get("Queue_Id", "?")) for r in raw)
    adam = [i for i, r in enumerate(rows) if "adam_flat_kernel" in r[2] or "adam_flat_dev_kernel" in r[2]]
    ends = adam[2::3]
    if len(ends) < a.steps + 1:
        raise SystemExit("only %d step ends in the trace" % len(ends))
    lo, hi = ends[-a.steps - 1] + 1, ends[-1] + 1
    sel = rows[lo:hi]
    n = float(a.steps)
    wall = (sel[-1][1] - sel[0][0]) / n
    print("columns: %s" % ", ".join(cols))
    print("key %s; steps %d: wall %.2f ms per step under the tracer, %d launches per step" % (key, a.steps, wall / 1e6, len(sel) / n))
    per = collections.OrderedDict()
    for s, e, name, k, q in sel:
        d = per.setdefault(k, {"n": 0, "t": 0, "short": 0, "short_t": 0, "names": collections.Counter(), "cnt": collections.Counter(), "q": set(),
                               "first": s, "last": e, "union": 0, "cur": None})
        d["n"] += 1
        d["t"] += e - s
        d["q"].add(q)
        if e - s < 10000:
            d["short"] += 1
            d["short_t"] += e - s
        d["names"][name] += e - s
        d["cnt"][name] += 1
        d["last"] = max(d["last"], e)
    for k, d in sorted(per.items(), key=lambda kv: -kv[1]["t"]):
        print("\n== %s %s (queues %s): %.1f launches, kernel time %.2f ms per step; < 10 us: %.1f launches, %.2f ms"
              % (key, k, ",".join(sorted(d["q"])), d["n"] / n, d["t"] / n / 1e6, d["short"] / n, d["short_t"] / n / 1e6))
        for name, t in d["names"].most_common(a.top):
            print("  %8.3f ms %6.1f x %7.1f us  %s" % (t / n / 1e6, d["cnt"][name] / n, t / d["cnt"][name] / 1e3, name[:140]))


if __name__ == "__main__":
    main()
