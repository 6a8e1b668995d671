"""Per-stream timeline of the LAST bench step in a rocprofv3 kernel trace (t_kernel_trace.csv): per-kernel time, launch gaps.
usage: python tools/trace_gaps.py t_kernel_trace.csv [anchor_kernel_substring]   (development aid)"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else 'bbox_init'
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
i0 = starts[-1]
seq = rows[i0:]
t0 = int(seq[0]['Start_Timestamp'])
by_stream = defaultdict(list)
for r in seq:
    by_stream[r['Stream_Id']].append(r)
for sid, rs in by_stream.items():
    prev = None
    gaps = busy = 0
    per = defaultdict(lambda: [0, 0.0])
    gap_after = defaultdict(float)
    for r in rs:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if prev is not None and s > prev[1]:
            gaps += s - prev[1]
            gap_after[prev[0][:36]] += (s - prev[1]) / 1e3
        busy += e - s
        k = r['Kernel_Name'][:36]
        per[k][0] += 1
        per[k][1] += (e - s) / 1e3
        prev = (r['Kernel_Name'], max(e, prev[1]) if prev else e)
    first, last = int(rs[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rs)
    print('stream %s: %d kernels, span %.1f us (from +%.1f us), busy %.1f us, gaps %.1f us' % (sid, len(rs), (last - first) / 1e3, (first - t0) / 1e3, busy / 1e3, gaps / 1e3))
    for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:24]:
        print('   %-38s n=%3d  %8.1f us   gap after: %7.1f us' % (k, n, t, gap_after.get(k, 0.0)))
