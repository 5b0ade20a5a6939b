"""Regenerate the two measured tables of DESIGN.md section 6 (per-kernel times of the default bench line, SQ counters of the
ConvNeXt kernels) from the committed profiles/<tag>_*.json, and print the sums the prose quotes.

    python tools/gen_design_tables.py r03_final          # rewrites DESIGN.md in place
"""
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r03_final"
d = json.loads((ROOT / "profiles" / f"{tag}_bench_default.json").read_text())
sq = json.loads((ROOT / "profiles" / f"{tag}_cnx_sq_counters.json").read_text())

rows = ["| kernel | ms / step | launches / step | ms / launch | bound, frac |", "|---|---|---|---|---|"]
for r in d["top_kernels"]:
    if r["per_step_ms"] < 0.27:
        continue
    fr = f"{r['bound']} {r['frac']:.3f}" if r.get("frac") is not None else "-"
    rows.append(f"| `{r['kernel']}` | {r['per_step_ms']:.2f} | {r['launches']} | {r['avg_ms']:.3f} | {fr} |")
t1 = "\n".join(rows)

rows = ["| kernel | ms | VALU-active / SIMD | MFMA-busy / SIMD | waves / SIMD | wait-any share |", "|---|---|---|---|---|---|"]
for k, e in sq.items():
    rows.append(f"| `{k}` | {e['duration_ns'] / 1e6:.3f} | {e['valu_active_per_simd']:.2f} | {e['mfma_busy_per_simd']:.2f} | "
                f"{e['waves_per_simd']:.2f} | {e['wait_any_frac']:.2f} |")
t2 = "\n".join(rows)

p = ROOT / "DESIGN.md"
lines = p.read_text().split("\n")


def splice(header_prefix, table, occurrence=0):
    idx = [i for i, l in enumerate(lines) if l.startswith(header_prefix)][occurrence]
    j = idx
    while j < len(lines) and lines[j].startswith("|"):
        j += 1
    lines[idx:j] = table.split("\n")


splice("| kernel | ms / step | launches / step", t1, 0)
splice("| kernel | ms | VALU-active / SIMD", t2, 0)
p.write_text("\n".join(lines))

cat = {"gemm_adamw": 0.0, "cnx": 0.0, "gemm": 0.0, "rest": 0.0}
for r in d["top_kernels"]:
    k = r["kernel"]
    key = "gemm_adamw" if k.startswith("gemm_adamw") else "cnx" if k.startswith("cnx") else "gemm" if k.startswith("gemm") else "rest"
    cat[key] += r["per_step_ms"]
print({k: round(v, 2) for k, v in cat.items()}, "sum", d["sum_kernel_ms_per_step"])
