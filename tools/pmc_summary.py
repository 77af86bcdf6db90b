#!/usr/bin/env python3
"""
Per-kernel means of the counters in a `rocprofv3 --pmc ... --kernel-trace --output-format csv` result
(`*_counter_collection.csv`), for the kernels of this library.

    python tools/pmc_summary.py gpurun_out/r2/prof/sq1 [more result directories ...]
"""
import collections
import csv
import glob
import re
import sys


def main(dirs):
    for d in dirs:
        for f in glob.glob(f'{d}/**/*_counter_collection.csv', recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            meta = {}
            for r in csv.DictReader(open(f)):
                m = re.search(r'(logl_\w+<[^>]*>|walk_kernel<[^>]*>|pass_[abc]_kernel|draw_kernel|reduce_partials_kernel|validate_kernel)', r['Kernel_Name'])
                if not m:
                    continue
                k = m.group(1)
                agg[k][r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
                meta[k] = (r['Grid_Size'], r['Workgroup_Size'], r['VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'], r['Scratch_Size'])
            print(f"# {f}")
            for k, v in agg.items():
                g, wg, vg, sg, lds, scr = meta[k]
                print(f"{k}: grid={g} workgroup={wg} vgpr={vg} sgpr={sg} lds={lds} scratch={scr}")
                for c, vals in sorted(v.items()):
                    xs = [x[0] for x in vals]
                    ds = [x[1] for x in vals]
                    print(f"    {c:22s} dispatches={len(xs):3d}  mean={sum(xs) / len(xs):16.1f}   mean duration {sum(ds) / len(ds) / 1e3:9.1f} us")


if __name__ == '__main__':
    main(sys.argv[1:])
