#!/usr/bin/env python3
"""
Writes the counters of tools/profile_r04.sh (gpurun_out/r4/r04_pmc.txt: FETCH_SIZE / WRITE_SIZE per dispatch of the two kernels
of a step) into profiles/r04_hbm_traffic.json and stamps it with the hash of the kernel sources they were measured at
(bench.py quotes `roofline.traffic` only while that hash matches).    python tools/restamp_traffic.py [pmc summary]
"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'r4', 'r04_pmc.txt')
vals, kernel = {}, None
for line in open(src):
    if line.startswith('walk_kernel<'):
        kernel = 'walk_kernel'
    elif line.startswith('logl_kernel<'):
        kernel = 'frame_loop_over_work_lists' if ('grid=131072' in line or 'grid=65536' in line) else None   # the listed launch of the headline
    m = re.match(r'\s+(FETCH_SIZE|WRITE_SIZE)\s+dispatches=\s*(\d+)\s+mean=\s*([\d.]+)', line)
    if m and kernel and int(m.group(2)) > 8:
        vals.setdefault(kernel, {})[m.group(1) + '_KiB'] = float(m.group(3))
path = os.path.join(ROOT, 'profiles', 'r04_hbm_traffic.json')
d = json.load(open(path))
for k in ('walk_kernel', 'frame_loop_over_work_lists'):
    assert set(vals[k]) == {'FETCH_SIZE_KiB', 'WRITE_SIZE_KiB'}, vals
    d[k] = vals[k]
total = sum(sum(v.values()) for v in (vals['walk_kernel'], vals['frame_loop_over_work_lists'])) * 1024
d['traffic_bytes_corrected'] = int(round(total))
d['traffic_bytes_upper'] = int(round(total + vals['frame_loop_over_work_lists']['FETCH_SIZE_KiB'] * 1024))
d['kernel_source_hash'] = bench.kernel_source_hash()
json.dump(d, open(path, 'w'), indent=1)
print(d['kernel_source_hash'], d['traffic_bytes_corrected'], d['traffic_bytes_upper'], vals)
