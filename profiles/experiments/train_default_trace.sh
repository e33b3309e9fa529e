cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt && mkdir -p /tmp/kt
rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/profiles/experiments/train_default_trace.py > /tmp/kt/log.txt 2>&1
tail -1 /tmp/kt/log.txt
python - <<'PY'
import sqlite3, glob, re
db = glob.glob('/tmp/kt/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# the last 40 steps: cut at adam_step kernels (2 per step)
adam = [i for i, r in enumerate(rows) if 'adam_step' in r[0]]
lo = adam[-81] + 1 if len(adam) > 81 else 0
rows = rows[lo:]
agg = {}
for n, s, e in rows:
    n = re.sub(r'\(anonymous namespace\)::', '', n)[:90]
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values()); span = (rows[-1][2] - rows[0][1]) / 1e3
print("kernels per step %.1f, kernel time per step %.1f us, span per step %.1f us" % (len(rows) / 40, tot / 40, span / 40))
for n, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%6.1f calls/step %8.1f us/step  %s" % (k / 40, t / 40, n))
PY
