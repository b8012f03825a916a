#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) result into a small text summary for profiles/.

    python scripts/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/<name>.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print("# per kernel: calls, total us, average us, % of GPU time")
    for name, calls, total, avg, pct in cur.execute(
        "select name, total_calls, total_duration, average, percentage from top_kernels"
    ):
        print(f"{calls:6d} {total:14.3f} {avg:12.3f} {pct:7.2f}%  {name}")
    print("# per kernel: grid, workgroup, vgpr, accum_vgpr, sgpr, lds, scratch (first dispatch)")
    seen = set()
    for row in cur.execute(
        "select name, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size from kernels"
    ):
        if row[0] in seen:
            continue
        seen.add(row[0])
        print("  grid=%d wg=%d vgpr=%d agpr=%d sgpr=%d lds=%d scratch=%d  %s" % (row[1:] + (row[0],)))
    try:
        rows = list(cur.execute(
            "select k.name, p.name, sum(e.value), count(*) from rocpd_pmc_event e "
            "join rocpd_info_pmc p on e.pmc_id = p.id join kernels k on e.event_id = k.id group by k.name, p.name"))
        if rows:
            print("# PMC counters: kernel, counter, sum over dispatches, dispatches")
            for r in rows:
                print(f"  {r[1]:32s} {r[2]:20.1f} {r[3]:5d}  {r[0][:80]}")
    except sqlite3.Error as e:
        print(f"# (no PMC events: {e})")


if __name__ == "__main__":
    main(sys.argv[1])
