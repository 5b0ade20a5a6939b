#!/bin/bash
# Effective shader clock per kernel = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (MI355X_MICROARCH.md, DVFS section).
# usage (GPU box, from the repo root): tools/probe_clock.sh <out dir> tools/bench_cnx.py 64
# The profiled program is always `python3 <absolute script path> <args>`: under --pmc only the program itself may follow
# `--` (no env / bash -c / launcher hop), and the script path is resolved against the repo root BEFORE the cd to /tmp.
OUT=$1; shift
SCRIPT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
case "$SCRIPT" in
  /*) ;;
  *) SCRIPT="$R/$SCRIPT" ;;
esac
case "$SCRIPT" in
  *.py) ;;
  *) echo "usage: tools/probe_clock.sh <out dir> <python script> [args]   (got: $SCRIPT)" >&2; exit 2 ;;
esac
[ -f "$SCRIPT" ] || { echo "no such script: $SCRIPT" >&2; exit 2; }
case "$OUT" in
  /*) ;;
  *) OUT="$R/$OUT" ;;
esac
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d "$OUT" --output-format csv -- python3 "$SCRIPT" "$@" > "$OUT/run.log" 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
cc = glob.glob(out + "/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])      # ns
    if dur < 200e3:                                                      # the quotient reads high on short dispatches
        continue
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
    a = agg[k]; a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += dur
for k, (n, c, d) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print(f"{k[:70]:70s} n={n:3d} avg {d / n / 1e6:7.3f} ms  clock {c / 8 / d:5.2f} GHz")
PY
