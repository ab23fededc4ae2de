"""Derive profiles/r01_dominant_kernel_traffic.json's per-launch HBM bytes from the two rocprofv3 PMC passes
(`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate runs of `python3 bench.py --roofline-only`).

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [--last 30]

Keeps the last N k_conv3h / k_conv3p dispatches (one network evaluation's launch set), sums the counters (KiB), doubles
FETCH_SIZE (gfx950 tallies 128-byte requests at 64 B -- MI355X_MICROARCH.md, HBM section) and prints the numbers;
with --write it also updates the JSON that bench.py reads for `roofline.traffic`."""
import argparse
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_sum(path, counter, n):
    rows = [r for r in csv.DictReader(open(path)) if ("k_conv3h" in r["Kernel_Name"] or "k_conv3p" in r["Kernel_Name"]) and r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    rows = rows[-n:]
    return sum(float(r["Counter_Value"]) for r in rows), len(rows), rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch")
    ap.add_argument("write")
    ap.add_argument("--last", type=int, default=29)
    ap.add_argument("--write-json", action="store_true")
    ap.add_argument("--round", default="r04", help="profiles/<round>_dominant_kernel_traffic.json is written (seeded from the previous round's)")
    a = ap.parse_args()
    f, nf, rows = last_sum(a.fetch, "FETCH_SIZE", a.last)
    w, nw, _ = last_sum(a.write, "WRITE_SIZE", a.last)
    assert nf == nw == a.last, (nf, nw)
    fetch = 2.0 * f * 1024 / a.last
    write = w * 1024 / a.last
    print(f"fetch {fetch/1e6:.1f} MB + write {write/1e6:.1f} MB = {(fetch+write)/1e6:.1f} MB per launch over {a.last} launches")
    if a.write_json:
        p = os.path.join(ROOT, "profiles", f"{a.round}_dominant_kernel_traffic.json")
        prev = [q for q in (p, os.path.join(ROOT, "profiles", "r03_dominant_kernel_traffic.json"), os.path.join(ROOT, "profiles", "r02_dominant_kernel_traffic.json"),
                            os.path.join(ROOT, "profiles", "r01_dominant_kernel_traffic.json")) if os.path.exists(q)][0]
        j = json.load(open(prev))
        j.update(fetch_size_kib_sum=f, write_size_kib_sum=w, fetch_bytes_per_launch_corrected=int(fetch),
                 write_bytes_per_launch=int(write), hbm_bytes_per_launch=int(fetch + write), launches=a.last)
        json.dump(j, open(p, "w"), indent=1)
        print("updated", p)


if __name__ == "__main__":
    main()
