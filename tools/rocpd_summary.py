#!/usr/bin/env python3
"""
Turn a rocprofv3 result database (rocpd sqlite, the default output of ROCm 7.2's
`rocprofv3 --kernel-trace --stats`) into the plain-text per-kernel summary that is committed
under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print(f"# source: {path}")
    print("# rocprofv3 --kernel-trace --stats ; durations in microseconds")
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>12} {'pct':>7}  kernel")
    for name, calls, total, avg, pct in rows:
        print(f"{calls:6d} {total:12.3f} {avg:12.3f} {pct:7.2f}  {name}")
    try:
        rows = db.execute("select k.name, d.vgpr_count, d.accum_vgpr_count, d.sgpr_count, d.lds_size, d.scratch_size, "
                          "d.workgroup_size, d.grid_size, count(*) from kernels d join kernel_symbols k on 1=0").fetchall()
    except Exception:
        rows = []
    try:
        cur = db.execute("select * from kernels limit 1")
        cols = [c[0] for c in cur.description]
        want = [c for c in ('name', 'vgpr_count', 'accum_vgpr_count', 'sgpr_count', 'lds_size', 'lds_block_size',
                            'scratch_size', 'workgroup_size', 'workgroup_size_x', 'grid_size', 'grid_size_x') if c in cols]
        if want:
            print("\n# per-dispatch launch parameters (distinct)")
            for r in db.execute(f"select distinct {', '.join(want)} from kernels"):
                print("  " + ", ".join(f"{c}={v}" for c, v in zip(want, r)))
    except Exception as e:  # schema differs between ROCm versions: the table above is what matters
        print(f"# (no launch-parameter view: {e})")


if __name__ == '__main__':
    main(sys.argv[1])
