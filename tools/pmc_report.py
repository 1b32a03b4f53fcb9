"""Prints per-dispatch PMC counters from a rocprofv3 rocpd database (gpurun_out/.../*_results.db)."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
rows = cur.execute("select dispatch_id, kernel_name, counter_name, value, duration, vgpr_count, lds_block_size, scratch_size "
                   "from counters_collection order by dispatch_id").fetchall()
by = {}
for d, k, c, v, dur, vg, lds, sc in rows:
    by.setdefault((d, k, dur, vg, lds, sc), {})[c] = v
for (d, k, dur, vg, lds, sc), cs in by.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print("%4d %-28s %9.3f ms vgpr %s lds %s scratch %s | " % (d, k[:28], (dur or 0) / 1e6, vg, lds, sc)
          + " ".join("%s=%.4g" % (c.replace("SQ_", ""), v) for c, v in sorted(cs.items())))
