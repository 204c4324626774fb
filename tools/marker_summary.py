#!/usr/bin/env python3
"""Summarise the roctx ranges (one per C-ABI call, csrc/capi.hip ZK_API_RANGE) of a `rocprofv3 --marker-trace --kernel-trace` run:
  marker_summary.py <rocprofv3 output dir> <out.txt>
Lists every range name with its count and total / average host-side duration, and -- by containment of the kernels' dispatch timestamps in the
ranges of the same process -- which kernels each entry point launched."""
import glob, os, sqlite3, sys
from collections import defaultdict

d, out = sys.argv[1], sys.argv[2]
dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
lines = []
if not dbs:
    lines.append("no rocpd database under " + d)
else:
    db = sqlite3.connect(dbs[0])
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    def cols(t): return [r[1] for r in db.execute(f"pragma table_info('{t}')")]
    # rocpd: `regions` view (marker + API regions) with name / start / end; `kernels` view with name / start / end
    reg_t = next((t for t in ("regions", "rocpd_region", "markers") if t in tables), None)
    ranges = []
    if reg_t:
        c = cols(reg_t)
        name_c = next((x for x in ("name", "message") if x in c), None)
        cat_c = next((x for x in ("category", "kind") if x in c), None)
        import json
        ext_c = "extdata" if "extdata" in c else None
        q = f"select {name_c}, start, end" + (f", {ext_c}" if ext_c else ", NULL") + f" from {reg_t}"
        for row in db.execute(q):
            name = str(row[0] or "")
            if row[3]:                            # rocpd keeps a roctx range as name = 'roctxThreadRangeA', extdata = {"message": "<range name>"}
                try: name = json.loads(row[3]).get("message", name)
                except Exception: pass
            if "zkhip_" in name: ranges.append((name, int(row[1]), int(row[2])))
    lines.append(f"rocprofv3 --marker-trace --kernel-trace: {len(ranges)} roctx ranges named zkhip_* (tables: {', '.join(t for t in tables if 'region' in t or 'kernel' in t or 'marker' in t)})")
    per = defaultdict(list)
    for n, s, e in ranges: per[n].append(e - s)
    for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"  {n:44s} calls {len(v):5d}   host time total {sum(v) / 1e6:9.3f} ms   avg {sum(v) / len(v) / 1e3:9.1f} us")
    if "kernels" in tables and ranges:
        kc = cols("kernels")
        # a kernel belongs to the range that was open on the host when it was enqueued: rocpd keeps the enqueue as the dispatch's `start` of the
        # corresponding API record when present; fall back to the kernel's own start time (asynchronous: may lie after the range closed)
        tcol = next((x for x in ("dispatch_time", "enqueue_time", "start") if x in kc), "start")
        ranges.sort(key=lambda r: r[1])
        import bisect
        starts = [r[1] for r in ranges]
        by = defaultdict(lambda: defaultdict(int))
        for name, t in db.execute(f"select name, {tcol} from kernels"):
            i = bisect.bisect_right(starts, int(t)) - 1
            owner = ranges[i][0] if i >= 0 and (int(t) <= ranges[i][2] or tcol == "start") else "(outside any range)"
            by[owner][str(name).split("(")[0][-60:]] += 1
        lines.append("")
        lines.append(f"kernels per entry point (attributed by {tcol}; with `start` a kernel is attributed to the last range opened before it began to run):")
        for owner, ks in sorted(by.items()):
            lines.append(f"  {owner}")
            for k, cnt in sorted(ks.items(), key=lambda kv: -kv[1])[:14]:
                lines.append(f"      {cnt:5d} x {k}")
open(out, "w").write("\n".join(lines) + "\n")
