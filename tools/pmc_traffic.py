"""Fold rocprofv3 --pmc counter_collection CSVs into per-kernel HBM traffic.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>

Each directory holds one rocprofv3 run of the SAME command with --pmc FETCH_SIZE resp. --pmc WRITE_SIZE
(they do not fit one pass: FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2).  Units: KiB per dispatch.
Calibration (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE counts a wide coalesced 16 B/lane
read stream at HALF its bytes.  The kbench run includes a 1.2 GB torch copy whose byte count is known;
its measured/expected ratio is reported so the correction for this box is explicit."""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in acc.items()}


def main():
    fd, wd, out = sys.argv[1:4]
    fetch, write = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, (0, 0.0))
        w = write.get(k, (0, 0.0))
        res[k[:120]] = {"dispatches": max(f[0], w[0]), "fetch_KiB": round(f[1], 1), "write_KiB": round(w[1], 1)}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["fetch_KiB"] + kv[1]["write_KiB"]))[:25]:
        print("%10.1f MB fetch %10.1f MB write  x%d  %s" % (v["fetch_KiB"] / 1024 * 1.048576, v["write_KiB"] / 1024 * 1.048576,
                                                          v["dispatches"], k[:80]))


if __name__ == "__main__":
    main()
