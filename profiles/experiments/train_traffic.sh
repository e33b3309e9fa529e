#!/usr/bin/env bash
# HBM traffic of the training step (bs 512 x 10, BBB, eager) per kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
# bytes per step by kernel name (gfx950: FETCH_SIZE counts 1/2 of wide coalesced reads -- the "x2" column applies the correction).
#   gpurun -- 'bash profiles/experiments/train_traffic.sh r06'
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt_$C && mkdir -p /tmp/pt_$C
  TRAIN_STEPS_LONG=0 rocprofv3 --kernel-trace --pmc $C -d /tmp/pt_$C -o pt -- python $R/profiles/experiments/train_steps.py bbb 512 10 > /tmp/pt_$C/log.txt 2>&1
done
python - > "$OUT/${TAG}_pmc_train_traffic.txt" <<'PY'
import sqlite3, glob, re
print("# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python profiles/experiments/train_steps.py bbb 512 10   (15 eager steps; MB per step by kernel)")
tot = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    db = glob.glob('/tmp/pt_%s/**/*.db' % C, recursive=True)[0]
    c = sqlite3.connect(db)
    for n, v in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (C,)):
        n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)[:60]
        t = tot.setdefault(n, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
        t[C] += v
        if C == "FETCH_SIZE":
            t["n"] += 1
steps = 15.0
rows = sorted(tot.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"]))
print("%-62s %8s %12s %12s %12s" % ("kernel", "calls", "fetch MB", "fetch x2 MB", "write MB"))
sf = sw = 0.0
for n, t in rows[:30]:
    f, w = t["FETCH_SIZE"] / 1024 / steps, t["WRITE_SIZE"] / 1024 / steps
    sf += f; sw += w
    print("%-62s %8.1f %12.1f %12.1f %12.1f" % (n, t["n"] / steps, f, 2 * f, w))
print("%-62s %8s %12.1f %12.1f %12.1f" % ("TOTAL (rows shown)", "", sf, 2 * sf, sw))
PY
cat "$OUT/${TAG}_pmc_train_traffic.txt"
