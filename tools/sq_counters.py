"""Reduce a `rocprofv3 --pmc SQ_*` run to per-kernel averages (what profiles/rNN_cnx_sq_counters.json holds).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
        SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d <dir> --output-format csv -- python3 <repo>/tools/bench_cnx.py 64
    python tools/sq_counters.py <dir> profiles/r03_cnx_sq_counters.json

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over all waves;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles; SQ_BUSY_CYCLES is summed over the 32 shader engines.  Derived per kernel:
  kernel_cycles            = SQ_BUSY_CYCLES / 32
  valu_active_per_simd     = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * kernel_cycles)   -- share of the kernel's cycles in
                             which a SIMD's vector ALU is executing (1.0 = the pipe never idles)
  mfma_busy_per_simd       = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * kernel_cycles)
  waves_per_simd           = 4 * SQ_WAVE_CYCLES / (1024 * kernel_cycles)
  cycles_per_valu_inst     = 4 * SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU
  wait_any_frac / wait_inst_frac / active_any_frac = share of a resident wave's cycles parked at s_waitcnt / barrier,
                             stalled at issue, issuing.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

SIMDS, SES = 1024, 32


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name).strip()


def main(argv):
    d, out = argv[1], argv[2]
    pat = re.compile(argv[3]) if len(argv) > 3 else re.compile(r"^cnx_")
    files = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit(f"no *_counter_collection.csv under {d}")
    per = defaultdict(lambda: defaultdict(float))       # (kernel, dispatch) -> counter -> value
    dur = {}
    for r in csv.DictReader(open(files[0], newline="")):
        k = short(r["Kernel_Name"])
        if not pat.search(k):
            continue
        key = (k, int(r["Dispatch_Id"]))
        per[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if "Start_Timestamp" in r and r["Start_Timestamp"]:
            dur[key] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    agg = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for (k, _), cs in per.items():
        n[k] += 1
        for c, v in cs.items():
            agg[k][c] += v
    for (k, _), v in dur.items():
        agg[k]["duration_ns"] += v
    res = {}
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CYCLES", 0)):
        a = {c: v / n[k] for c, v in agg[k].items()}
        e = {"launches": n[k]}
        e.update({c: round(v) for c, v in a.items()})
        kc = a.get("SQ_BUSY_CYCLES", 0.0) / SES
        if kc > 0:
            e["kernel_cycles"] = round(kc)
            if "SQ_ACTIVE_INST_VALU" in a:
                e["valu_active_per_simd"] = round(4 * a["SQ_ACTIVE_INST_VALU"] / (SIMDS * kc), 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
                e["mfma_busy_per_simd"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * kc), 3)
            if "SQ_WAVE_CYCLES" in a:
                e["waves_per_simd"] = round(4 * a["SQ_WAVE_CYCLES"] / (SIMDS * kc), 2)
        if a.get("SQ_INSTS_VALU"):
            e["cycles_per_valu_inst"] = round(4 * a.get("SQ_ACTIVE_INST_VALU", 0.0) / a["SQ_INSTS_VALU"], 2)
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        if wc > 0:
            for c, nm in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_frac"), ("SQ_ACTIVE_INST_ANY", "active_any_frac")):
                if c in a:
                    e[nm] = round(a[c] / wc, 3)
        res[k] = e
    json.dump(res, open(out, "w"), indent=1)
    for k, e in res.items():
        print(f"{k[:60]:60s} n={e['launches']:3d} valu/simd {e.get('valu_active_per_simd')}  mfma/simd {e.get('mfma_busy_per_simd')}  "
              f"waves/simd {e.get('waves_per_simd')}  wait_any {e.get('wait_any_frac')} wait_inst {e.get('wait_inst_frac')}")


if __name__ == "__main__":
    main(sys.argv)
