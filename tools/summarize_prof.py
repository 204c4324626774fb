#!/usr/bin/env python3
"""Turn rocprofv3 output directories into the summaries committed under profiles/.
  summarize_prof.py stats <dir> <out.csv>            copy the --stats kernel table (largest first)
  summarize_prof.py pmc <fetch_dir> <write_dir> <out.txt> <out.json>   per-kernel averages of FETCH_SIZE / WRITE_SIZE (KiB per dispatch)
  summarize_prof.py sq <dir_a> <dir_b> <out.txt>     per-kernel averages of the SQ issue / wait counters (tools/collect_profiles.sh)
FETCH_SIZE gets the gfx950 x2 correction the MI355X guide prescribes for wide coalesced streams; both figures are listed."""
import csv, glob, json, os, re, sqlite3, sys
from collections import defaultdict

def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits: raise SystemExit(f"no *{suffix} under {d}")
    return hits[0]

def short(name):
    m = re.match(r"(?:void )?(zkhip::\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]

def counters(d, counter):
    """per kernel: (dispatches, average counter value); reads rocprofv3's rocpd database (default output) or its csv output"""
    acc = defaultdict(list)
    dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    if dbs:
        for name, value in sqlite3.connect(dbs[0]).execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            acc[short(name)].append(float(value))
    else:
        with open(find(d, "counter_collection.csv")) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == counter: acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}

def kernel_stats(d):
    dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    if not dbs: return list(csv.reader(open(find(d, "kernel_stats.csv"))))
    per = defaultdict(list)
    for name, dur in sqlite3.connect(dbs[0]).execute("select name, duration from kernels"): per[name].append(int(dur))
    total = sum(sum(v) for v in per.values())
    # MedianNs (round 5) beside rocprofv3's own columns: the first launches of a process run at the clocks the card idled at (30-40 ms of work
    # to ramp: profiles/r03_clock_ramp.txt), so the average of a short profiled run sits a few per cent above the steady state the bench line times
    rows = [["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "MedianNs"]]
    for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        sv = sorted(v)
        med = sv[len(sv) // 2] if len(sv) % 2 else (sv[len(sv) // 2 - 1] + sv[len(sv) // 2]) / 2.0
        rows.append([name, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / total, 2), min(v), max(v), med])
    return rows

if sys.argv[1] == "pmc_ntt":
    # summarize_prof.py pmc_ntt <out.json> <log_n> <fetch_dir> <write_dir> [<log_n> <fetch_dir> <write_dir> ...]: FETCH / WRITE per k_ntt_pass launch
    out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace), python3 tools/ntt_prof.py <log_n>",
           "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported; counters are KiB", "ntt": {}}
    args = sys.argv[3:]
    for i in range(0, len(args), 3):
        L, fd, wd = int(args[i]), args[i + 1], args[i + 2]
        f, w = counters(fd, "FETCH_SIZE"), counters(wd, "WRITE_SIZE")
        rows = {}
        for name in sorted(set(f) | set(w)):
            if "ntt" not in name: continue
            nf, vf = f.get(name, (0, 0.0)); nw, vw = w.get(name, (0, 0.0))
            rows[name] = {"dispatches": max(nf, nw), "fetch_bytes_raw": round(vf * 1024), "fetch_bytes_corrected": round(2 * vf * 1024), "write_bytes": round(vw * 1024),
                          "hbm_bytes_per_launch": round(2 * vf * 1024 + vw * 1024), "algorithmic_bytes_per_launch": 64 * (1 << L),
                          "traffic_over_algorithmic": round((2 * vf * 1024 + vw * 1024) / (64.0 * (1 << L)), 3)}
        out["ntt"][f"2^{L}"] = rows
    json.dump(out, open(sys.argv[2], "w"), indent=1)
elif sys.argv[1] == "stats":
    rows = kernel_stats(sys.argv[2])
    with open(sys.argv[3], "w", newline="") as f: csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerows(rows)
elif sys.argv[1] == "sq":
    names = {sys.argv[2]: ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"], sys.argv[3]: ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS"]}
    tab = defaultdict(dict)
    for d, cs in names.items():
        for c in cs:
            for k, (n, v) in counters(d, c).items(): tab[k][c] = v; tab[k]["dispatches"] = n
    lines = ["rocprofv3 --kernel-trace --pmc <4 SQ counters> (two separate passes), workload tools/prof_msm.py 20 3 24 (3 prepared MSMs 2^20, 3 NTTs 2^24).",
             "Per-dispatch averages.  SQ_WAVE_CYCLES = SQ_ACTIVE_INST_ANY + SQ_WAIT_INST_ANY (issue stall: the pipe is taken by another wave) + SQ_WAIT_ANY",
             "(parked at s_waitcnt / barrier), in quad-cycles summed over waves (MI355X_MICROARCH.md).  active% x (waves per SIMD) ~ VALU occupancy.", ""]
    for k, v in sorted(tab.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:26]:
        wc = v.get("SQ_WAVE_CYCLES", 0) or 1
        lines.append(f"  {k:40s} n={v.get('dispatches', 0):3d} waves={v.get('SQ_WAVES', 0):9.0f} valu_insts={v.get('SQ_INSTS_VALU', 0):13.0f} vmem={v.get('SQ_INSTS_VMEM', 0):10.0f} lds={v.get('SQ_INSTS_LDS', 0):10.0f}"
                     f"  active={100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.1f}% issue_stall={100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:5.1f}% parked={100 * v.get('SQ_WAIT_ANY', 0) / wc:5.1f}%")
    open(sys.argv[4], "w").write("\n".join(lines) + "\n")
else:
    fetch, write = counters(sys.argv[2], "FETCH_SIZE"), counters(sys.argv[3], "WRITE_SIZE")
    lines = ["rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), same command as the bench line:",
             "python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-general-path.  Counter unit: KiB per dispatch (average over the kernel's",
             "dispatches).  gfx950 (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts half the bytes of a wide coalesced stream; x2 = corrected.", "", "FETCH_SIZE"]
    for k, (n, v) in sorted(fetch.items(), key=lambda kv: -kv[1][1])[:16]:
        lines.append(f"  {k:44s} dispatches={n:4d}  avg = {v:12.1f} KiB = {v*1024/1e6:9.1f} MB   x2 = {2*v*1024/1e6:9.1f} MB")
    lines += ["", "WRITE_SIZE"]
    for k, (n, v) in sorted(write.items(), key=lambda kv: -kv[1][1])[:16]:
        lines.append(f"  {k:44s} dispatches={n:4d}  avg = {v:12.1f} KiB = {v*1024/1e6:9.1f} MB")
    open(sys.argv[4], "w").write("\n".join(lines) + "\n")
    def pick(table, k):   # exact name, or the template instance with the most dispatches (k_accumulate<true> on the prepared path)
        hits = [v for name, v in table.items() if name == k or name.startswith(k + "<")]
        return max(hits, key=lambda v: v[0]) if hits else (0, 0.0)
    def entry(k):
        fr = pick(fetch, k)[1] * 1024; wr = pick(write, k)[1] * 1024
        return {"fetch_bytes_raw": round(fr), "fetch_bytes_corrected": round(2 * fr), "write_bytes": round(wr), "hbm_bytes_per_launch": round(2 * fr + wr)}
    out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace), python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-general-path",
           "workload": "prepared MSM 2^20 (bench headline configuration)",
           "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported; counters are KiB",
           "k_accumulate": entry("zkhip::k_accumulate"), "k_coarse_sorted": entry("zkhip::k_coarse_sorted"), "k_fine_sorted": entry("zkhip::k_fine_sorted")}
    # every kernel of one MSM (round 4): launches per MSM = the kernel's dispatches / k_accumulate's dispatches in the same pass (one-time
    # kernels -- table build, synthetic bases -- have fewer dispatches than MSMs and are left out); whole-MSM traffic = sum over kernels
    n_msm_f, n_msm_w = pick(fetch, "zkhip::k_accumulate")[0], pick(write, "zkhip::k_accumulate")[0]
    per = {}
    for name in sorted(set(fetch) | set(write)):
        nf, vf = fetch.get(name, (0, 0.0)); nw, vw = write.get(name, (0, 0.0))
        if n_msm_f == 0 or n_msm_w == 0 or (nf < n_msm_f and nw < n_msm_w): continue
        lf, lw = nf / n_msm_f, nw / n_msm_w
        per[name] = {"launches_per_msm": round(max(lf, lw), 2), "fetch_bytes_raw_per_launch": round(vf * 1024), "write_bytes_per_launch": round(vw * 1024),
                     "hbm_bytes_per_msm_raw_fetch": round(vf * 1024 * lf + vw * 1024 * lw), "hbm_bytes_per_msm_x2_fetch": round(2 * vf * 1024 * lf + vw * 1024 * lw)}
    out["kernels_per_msm"] = per
    out["whole_msm"] = {"hbm_bytes_raw_fetch": sum(v["hbm_bytes_per_msm_raw_fetch"] for v in per.values()),
                        "hbm_bytes_x2_fetch": sum(v["hbm_bytes_per_msm_x2_fetch"] for v in per.values()), "algorithmic_bytes": 96 * (1 << 20),
                        "note": "sum over every kernel of one prepared 2^20 MSM; raw = FETCH_SIZE as counted, x2 = with the guide's gfx950 correction applied to every kernel"}
    json.dump(out, open(sys.argv[5], "w"), indent=1)
