#!/usr/bin/env python3
"""Kernel statistics (the `--stats` summary) from a rocprofv3 rocpd SQLite database.
usage: tools/rocpd_stats.py trace_results.db [out.md]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("""
  select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start),
         max(d.grid_size_x), max(d.workgroup_size_x), max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count),
         max(d.group_segment_size)
  from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
  group by s.kernel_name order by 3 desc""").fetchall()
total = sum(r[2] for r in rows) or 1
lines = ["| kernel | calls | total ms | avg us | min us | max us | % | grid | wg | vgpr | agpr | sgpr | lds B |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    name = r[0].split("(")[0]
    name = name.replace("pcoa::(anonymous namespace)::", "")
    lines.append("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f | %d | %d | %s | %s | %s | %s |" %
                 (name[:70], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10], r[11]))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
