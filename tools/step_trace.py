"""Per-STEP accounting of a rocprofv3 kernel trace of `bench.py` (train workload): steps are cut at every third adam_flat_kernel
launch (opt_D, opt_G, opt_F: the last launch of a step), the last `--steps` whole steps are kept, and kernel time / launches are
summed per category.  Prints wall per step (first kernel start -> last kernel end), busy, idle, and the category table.

    python tools/step_trace.py <kernel_trace.csv> [--steps 3] [--top 0]"""
import argparse, collections, csv

CATS = (("own winograd", ("winograd_conv", "winograd_weights", "conv3x3_thin")), ("vendor winograd", ("miopenSp3AsmConv",)),
        ("own wgrad 3x3", ("conv3x3_wgrad",)), ("own wgrad tiled", ("conv_wgrad_tile", "conv_wgrad_generic")),
        ("own conv_fwd (fwd + dgrad modes)", ("conv_fwd_kernel",)),
        ("igemm wrw", ("igemm_wrw",)), ("igemm bwd", ("igemm_bwd",)), ("igemm fwd", ("igemm_fwd",)), ("transposes", ("batched_transpose",)),
        ("fills / memsets", ("SubTensorOp", "FillFunctor", "fillBuffer")), ("own other", ("ffwm::",)), ("aten reduce", ("reduce_kernel",)),
        ("aten elementwise", ("at::native",)), ("hipblaslt", ("Cijk",)), ("miopen bn", ("BatchNorm",)))


def cat(n):
    for name, keys in CATS:
        if any(k in n for k in keys):
            return name
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--top", type=int, default=0, help="also list the N kernels with the most time")
    a = ap.parse_args()
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(a.trace)))
    adam = [i for i, r in enumerate(rows) if "adam_flat_kernel" in r[2] or "adam_flat_dev_kernel" in r[2]]      # eager / captured step
    ends = adam[2::3]                      # index of the last kernel of each step
    if len(ends) < a.steps + 1:
        raise SystemExit("only %d step ends in the trace" % len(ends))
    lo, hi = ends[-a.steps - 1] + 1, ends[-1] + 1
    sel = rows[lo:hi]
    n = float(a.steps)
    wall = (sel[-1][1] - sel[0][0]) / n
    busy = sum(e - s for s, e, _ in sel) / n
    # union of busy intervals (kernels may overlap)
    cur_s, cur_e, union = sel[0][0], sel[0][1], 0
    for s, e, _ in sel[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    union /= n
    print("steps %d: wall %.2f ms per step (under the tracer), kernel time %.2f ms, GPU busy %.2f ms (%.0f %%), idle %.2f ms, %d launches per step"
          % (a.steps, wall / 1e6, busy / 1e6, union / 1e6, 100.0 * union / wall, (wall - union) / 1e6, len(sel) / n))
    agg = collections.OrderedDict()
    for s, e, name in sel:
        d = agg.setdefault(cat(name), [0, 0])
        d[0] += 1
        d[1] += e - s
    print("%-36s %9s %9s %7s" % ("category", "launches", "ms/step", "share"))
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-36s %9.1f %9.2f %6.1f%%" % (k, c / n, t / n / 1e6, 100.0 * t / (busy * n)))
    if a.top:
        per = collections.Counter()
        cnt = collections.Counter()
        for s, e, name in sel:
            per[name] += e - s
            cnt[name] += 1
        for name, t in per.most_common(a.top):
            print("%8.2f ms %6.1f x  %s" % (t / n / 1e6, cnt[name] / n, name[:150]))


if __name__ == "__main__":
    main()
