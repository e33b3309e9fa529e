#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd sqlite database.
usage: summarize_pmc.py results.db [name-substring] [--steps N]
With --steps N (the number of MC steps the profiled command ran, warm-up included) the summary ends with STEP_TOTAL rows:
a counter summed over ALL launches of a kernel family ("pconv_gemm", "reparam") divided by N = per-step totals, which
bench.py parses (FETCH_SIZE / WRITE_SIZE are in KB)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:70]


def main():
    args = sys.argv[1:]
    steps = None
    if "--steps" in args:
        i = args.index("--steps")
        steps = int(args[i + 1])
        del args[i:i + 2]
    c = sqlite3.connect(args[0])
    filt = args[1] if len(args) > 1 else ""
    rows = c.execute("select kernel_name, grid_size_x, grid_size_y, grid_size_z, workgroup_size_x, dispatch_id, counter_name, "
                     "value, duration from counters_collection").fetchall()
    per = {}
    fam = {}
    for n, gx, gy, gz, wx, did, cn, v, dur in rows:
        for f in ("pconv_gemm", "pconv_bf16", "pconv_c8x3", "reparam", "maxpool", "mc_tail", "s2d_c8s3"):
            if f in n:
                t = fam.setdefault((f, cn), [0.0, 0])
                t[0] += v
                t[1] += 1
        if filt and filt not in n:
            continue
        key = (short(n), gx // max(wx, 1), gy, gz)
        d = per.setdefault(key, {}).setdefault(cn, [0.0, 0])
        d[0] += v
        d[1] += 1
        dd = per[key].setdefault("_dur_ns", [0.0, 0])
        dd[0] += dur
        dd[1] += 1
    for key, cs in per.items():
        print(key)
        for cn, (s, n) in sorted(cs.items()):
            print(f"    {cn:32s} avg {s / n:16.1f}   (n={n})")
        avg = {cn: s / n for cn, (s, n) in cs.items()}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg and avg["GRBM_GUI_ACTIVE"] > 0:
            # MFMA_BUSY is summed over the 1024 SIMDs (256 CUs x 4); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so /8 = the
            # shader-clock cycles the launch took
            cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
            util = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
            clk = cyc / avg["_dur_ns"]
            print(f"    -> matrix-pipe utilisation {100 * util:5.1f} % of the launch's cycles  (clock {clk:.2f} GHz over the launch)")


    if steps:
        for (f, cn), (tot, cnt) in sorted(fam.items()):
            print(f"STEP_TOTAL {f} {cn} KB_per_step={tot / steps:.1f} launches_per_step={cnt / steps:.2f}")


if __name__ == "__main__":
    main()
