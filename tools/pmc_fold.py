"""Fold rocprofv3 --pmc counter_collection CSVs of several passes into per-kernel averages.

    python tools/pmc_fold.py <out.json> <pass_dir> [<pass_dir> ...]

Per kernel name (ffwm:: kernels only, template arguments kept): dispatches, and the mean per dispatch of every
counter found.  Derived: hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (FETCH_SIZE counts a wide coalesced read
stream at half its bytes on gfx950 -- MI355X_MICROARCH.md, HBM section; profiles/r01_kbench_pmc_raw.json holds this
box's calibration), lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, l2_hit = TCC_HIT / (HIT + MISS).
A pass that produced no rows is reported as missing -- nothing is written for a counter that was not measured."""
import collections
import csv
import glob
import json
import sys


def main():
    out = sys.argv[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for d in sys.argv[2:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "ffwm::" not in k:
                    continue
                a = acc[k][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    res = {}
    for k, cs in acc.items():
        row = {"dispatches": max(n for n, _ in cs.values())}
        for c, (n, v) in cs.items():
            row[c] = v / n
        if "FETCH_SIZE" in row and "WRITE_SIZE" in row and row["FETCH_SIZE"] > 0:
            row["hbm_bytes"] = (2 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024
        if row.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_conflict_frac"] = row.get("SQ_LDS_BANK_CONFLICT", 0.0) / row["SQ_LDS_IDX_ACTIVE"]
        if "TCC_HIT_sum" in row:
            row["l2_hit"] = row["TCC_HIT_sum"] / max(1.0, row["TCC_HIT_sum"] + row.get("TCC_MISS_sum", 0.0))
        res[k[:160]] = row
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, row in sorted(res.items()):
        print(k[:110])
        print("   ", {c: (round(v, 3) if isinstance(v, float) and v < 10 else int(v)) for c, v in sorted(row.items())})


if __name__ == "__main__":
    main()
