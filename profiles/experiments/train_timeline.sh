#!/usr/bin/env bash
# One training step (bbb 512 10 by default) as a timeline: every kernel of the last step with its queue, start offset and duration.
#   gpurun -- 'bash profiles/experiments/train_timeline.sh "bbb 512 10"'
set -u
ARGS=${1:-"bbb 512 10"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && mkdir -p /tmp/kt
TRAIN_STEPS_LONG=0 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/profiles/experiments/train_steps.py $ARGS > /tmp/kt/log.txt 2>&1
tail -1 /tmp/kt/log.txt
python - <<'PY'
import sqlite3, glob, re
db = glob.glob('/tmp/kt/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else "1")
rows = c.execute(f"select name, start, end, {q}, {gx}, {wx} from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if 'adam_step' in r[0]]
lo, hi = adam[-3] + 1, adam[-1] + 1          # two adam launches per step: the last step
rows = rows[lo:hi]
t0 = rows[0][1]
qs = {}
for n, s, e, qq, g, w in rows:
    qi = qs.setdefault(qq, len(qs))
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\(.*', '', n)[:60]
    print("%8.1f us  +%7.1f us  q%d  wg %6d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, qi, (g or 0) // max(w or 1, 1), n))
print("step span %.1f us" % ((rows[-1][2] - t0) / 1e3))
PY
