#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) from a rocprofv3 rocpd sqlite database.
usage: summarize_rocpd.py results.db [--after-spin N [--steps K]]   (the same numbers as `rocprofv3 --stats`, as text)
--after-spin N: only the kernels between the N-th at::cuda::spin_kernel (1-based) and the next one (or the end) --
bench.py parks the GPU behind such a kernel before its HIP-event-bracketed eager pass, so this isolates exactly the
launches the JSON's roofline block was timed on.
--last-steps K: only the last K Monte-Carlo steps of the trace (a step ends with its mc_tail kernel): with
`bench.py --no-extras --no-roofline` these are the hipGraph replays of the timed region.  --by-grid: one row per
(kernel, grid size), i.e. per layer."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) < 110 else name[:107] + "..."


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall() \
        if "grid_x" in [r[1] for r in c.execute("pragma table_info(kernels)")] else \
        [(r[0], r[1], r[2], 0, 0, 0, 0) for r in c.execute("select name, start, end from kernels order by start")]
    if "--after-spin" in sys.argv:
        n = int(sys.argv[sys.argv.index("--after-spin") + 1])
        spins = [i for i, r in enumerate(rows) if "spin_kernel" in r[0]]
        lo = spins[n - 1] + 1
        hi = spins[n] if n < len(spins) else len(rows)
        if "--steps" in sys.argv:        # a step ends with its mc_tail kernel
            want, seen = int(sys.argv[sys.argv.index("--steps") + 1]), 0
            for i in range(lo, hi):
                if "mc_tail" in rows[i][0]:
                    seen += 1
                    if seen == want:
                        hi = i + 1
                        break
        rows = rows[lo:hi]
    if "--last-steps" in sys.argv:
        want = int(sys.argv[sys.argv.index("--last-steps") + 1])
        tails = [i for i, r in enumerate(rows) if "mc_tail" in r[0]]
        if len(tails) > want:
            rows = rows[tails[-want - 1] + 1:tails[-1] + 1]
    by_grid = "--by-grid" in sys.argv
    agg = {}
    for name, s, e, gx, gy, gz, wx in rows:
        key = short(name) + (f"  [grid {gx // max(wx, 1)}]" if by_grid and "pconv" in name else "")
        a = agg.setdefault(key, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    span = rows[-1][2] - rows[0][1] if rows else 0
    print(f"# {db}: {len(rows)} dispatches, kernel time {tot/1e6:.3f} ms, span {span/1e6:.3f} ms")
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'share':>7}  kernel")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{a[0]:7d} {a[1]/1e3:12.1f} {a[1]/a[0]/1e3:10.2f} {a[2]/1e3:9.2f} {a[3]/1e3:9.2f} {100*a[1]/tot:6.2f}%  {k}")


if __name__ == "__main__":
    main()
