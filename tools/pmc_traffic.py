"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE, then WRITE_SIZE -- the TCC block cannot hold both).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py --steps 1 --warmup 1 \
        --no-cpu-baseline --no-decode --no-kernel-timing --no-overlap --no-loss-probe
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 bench.py ... (same)
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_final

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB and, on gfx950, FETCH_SIZE tallies the
128-byte requests of wide coalesced reads at 64 bytes (MI355X_MICROARCH.md, section HBM).  The two passes run the
same deterministic launch sequence, so launches are matched by (kernel symbol, grid size, order of occurrence).
Writes <out>_pmc_fetch_write_by_kernel.json (every launch of this library's kernels) and <out>_pmc_traffic.json
(average per kernel symbol: what bench.py puts into roofline.traffic).
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name).strip()      # drop the argument list


def load(dirname: str, counter: str):
    files = glob.glob(f"{dirname}/**/*_counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit(f"no *_counter_collection.csv under {dirname}")
    rows = []
    with open(files[0], newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), int(r["Grid_Size"]), float(r["Counter_Value"])))
    rows.sort()
    seen = defaultdict(int)
    out = {}
    for _, k, g, v in rows:
        out[(k, g, seen[(k, g)])] = v
        seen[(k, g)] += 1
    return out


def main(argv):
    fetch_dir, write_dir, out = argv[1:4]
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    ours = re.compile(r"^(adamw|gemm|cnx|mdct|colsum|cast|flow|loss|gelu|axpby|randn|time_embed|sample_tr|grn|ln16|adaln|gate|copy2d|transpose)")
    by_launch, agg = {}, defaultdict(lambda: [0, 0.0])
    for key in sorted(set(fe) & set(wr), key=lambda k: (k[0], k[1], k[2])):
        k, g, i = key
        if not ours.match(k):
            continue
        hbm = (2.0 * fe[key] + wr[key]) * 1024.0
        by_launch[f"{k} grid={g} #{i}"] = {"fetch_size_kib": fe[key], "write_size_kib": wr[key], "hbm_bytes": round(hbm)}
        agg[k][0] += 1
        agg[k][1] += hbm
    unmatched = len(set(fe) ^ set(wr))
    with open(out + "_pmc_fetch_write_by_kernel.json", "w") as f:
        json.dump(by_launch, f, indent=1)
    table = {"_comment": "average HBM bytes per launch per kernel symbol = sum over launches of (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                         "/ launches, from two separate rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE) over `python3 bench.py "
                         "--steps 1 --warmup 1 --no-cpu-baseline --no-decode --no-kernel-timing --no-overlap --no-loss-probe` (literal config, 1x "
                         "MI355X; x2 on FETCH_SIZE = the gfx950 correction of MI355X_MICROARCH.md); made by tools/pmc_traffic.py. "
                         f"Launches present in only one pass: {unmatched}."}
    for k, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        table[k] = {"launches": n, "avg_hbm_bytes_per_launch": round(tot / n)}
    with open(out + "_pmc_traffic.json", "w") as f:
        json.dump(table, f, indent=1)
    for k, v in list(table.items())[1:12]:
        print(f"{k:60s} {v['launches']:4d}  {v['avg_hbm_bytes_per_launch'] / 1e9:8.3f} GB")


if __name__ == "__main__":
    main(sys.argv)
