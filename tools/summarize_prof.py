#!/usr/bin/env python3
"""Turns rocprofv3 CSV output into the small summaries committed under profiles/ (development tool).

  python tools/summarize_prof.py stats  <rocprof_dir> <out.csv> "<command line that was profiled>"
      per-kernel and per-(kernel, grid) launch statistics of the llmie kernels from *_kernel_trace.csv
  python tools/summarize_prof.py pmc    <fetch_dir> <write_dir> <out.csv> <out.json> <kernel substring> <grid>
      HBM bytes per launch from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units; FETCH_SIZE x2 on gfx950
      for wide coalesced streaming reads, MI355X_MICROARCH.md HBM section)
  python tools/summarize_prof.py sq     <pmc_dir> <out.csv> "<command line>" [kernel substring ...]
      per-kernel averages of every SQ counter of one --pmc pass, plus the derived ratios (MFMA busy, waits, LDS conflicts)
  python tools/summarize_prof.py roofline <out.json> <key> <trace_dir> <kernel substring> <grid or 0> <stats csv> [<fetch_dir> <write_dir>]
      adds/replaces entry <key> of the JSON bench.py reads for its *_rocprof_trace fractions: the kernel's average duration
      in the kernel trace and, from the two PMC pass directories, its HBM bytes per launch -- only from counter rows of exactly the
      same kernel name and grid (anything else is refused)
"""
import csv, glob, json, os, sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, d))
    return hits[0]


def short(name):
    return name if len(name) < 160 else name[:157] + "..."


def stats(d, out, cmd):
    by_name, by_grid = defaultdict(list), defaultdict(list)
    with open(find(d, "kernel_trace.csv")) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if "llmie" not in n:
                continue
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            by_name[n].append(dur)
            by_grid[(n, grid)].append(dur)
    total = sum(sum(v) for v in by_name.values())
    with open(out, "w") as f:
        f.write("# %s\n# llmie kernels only; durations in ns (rocprofv3 kernel trace)\n" % cmd)
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for n, v in sorted(by_name.items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (short(n), len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / total, min(v), max(v)))
        f.write("# per (kernel, total grid size in threads)\nName,Grid,Calls,AverageNs,MinNs,MaxNs\n")
        for (n, g), v in sorted(by_grid.items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%d,%d,%.1f,%d,%d\n' % (short(n), g, len(v), sum(v) / len(v), min(v), max(v)))
    print("wrote", out)


def pmc(dfetch, dwrite, out, out_json, kernel_sub, grid):
    def collect(d, counter):
        acc = defaultdict(list)
        with open(find(d, "counter_collection.csv")) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == counter and "llmie" in r["Kernel_Name"]:
                    acc[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
        return acc
    fe, wr = collect(dfetch, "FETCH_SIZE"), collect(dwrite, "WRITE_SIZE")
    rows, pick = [], None
    for key in sorted(fe, key=lambda k: -sum(fe[k])):
        n, g = key
        fk = sum(fe[key]) / len(fe[key])
        wk = sum(wr[key]) / len(wr[key]) if key in wr else 0.0
        rd, wb = 2.0 * fk * 1024.0, wk * 1024.0
        rows.append((n, g, len(fe[key]), fk, wk, rd, wb, rd + wb))
        if kernel_sub in n and g == int(grid):
            pick = dict(kernel="%s|grid=%d" % (short(n), g), hbm_bytes_per_launch=rd + wb,
                        source="%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH_SIZE x2 gfx950 correction)" % out)
    with open(out, "w") as f:
        f.write("# HBM traffic per launch from rocprofv3 PMC counters, two separate passes (eager launches, --no-graph)\n")
        f.write("# counters in KiB; read bytes = 2 * FETCH_SIZE * 1024 (gfx950 wide-read correction), write bytes = WRITE_SIZE * 1024\n")
        f.write("Name,Grid,Launches,FETCH_SIZE_KiB,WRITE_SIZE_KiB,ReadBytes,WriteBytes,HbmBytesPerLaunch\n")
        for r in rows:
            f.write('"%s",%d,%d,%.2f,%.2f,%.0f,%.0f,%.0f\n' % ((short(r[0]),) + r[1:]))
    if pick:
        with open(out_json, "w") as f:
            json.dump(pick, f, indent=1)
    print("wrote", out, "and" if pick else "(kernel not found for)", out_json)


def sq(d, out, cmd, subs):
    acc, grids = defaultdict(lambda: defaultdict(list)), {}
    with open(find(d, "counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if "llmie" not in n or (subs and not any(x in n for x in subs)):
                continue
            key = (n, int(r["Grid_Size"]))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for v in acc.values() for c in v})
    with open(out, "w") as f:
        f.write("# %s\n# per (kernel, grid): mean counter value per launch; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles\n"
                "# summed over waves, SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (MI355X_MICROARCH.md, counter table)\n" % cmd)
        f.write("# mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES * 32): SQ_BUSY_CYCLES is summed over the 32 shader engines\n"
                "#   (8 XCDs x 4), so SQ_BUSY_CYCLES / 32 = kernel duration in shader cycles and x 1024 SIMDs = all SIMD cycles;\n"
                "#   wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES;\n"
                "#   issue_stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE\n")
        f.write("# valu_busy_frac = SQ_ACTIVE_INST_VALU (quad-cycles, summed over waves) * 4 / (SQ_BUSY_CYCLES / 32 * 1024 SIMDs)\n")
        f.write("Name,Grid,Launches," + ",".join(names) + ",mfma_busy_frac,wait_frac,issue_stall_frac,lds_conflict_frac,valu_busy_frac\n")
        for key in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_WAVE_CYCLES", [0]))):
            v = {c: sum(x) / len(x) for c, x in acc[key].items()}
            n_l = max(len(x) for x in acc[key].values())

            def ratio(a, b, k=1.0):
                return "%.4f" % (v[a] / (v[b] * k)) if a in v and b in v and v[b] > 0 else ""
            f.write('"%s",%d,%d,%s,%s,%s,%s,%s,%s\n' % (short(key[0]), key[1], n_l, ",".join("%.0f" % v.get(c, 0.0) for c in names),
                                                      ratio("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", 32.0), ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"),
                                                      ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"), ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
                                                      ratio("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", 8.0)))
    print("wrote", out)


def roofline(out_json, key, d, kernel_sub, grid, stats_csv, dfetch=None, dwrite=None):
    """entry <key>: the kernel's average duration in the kernel trace of <d>; kernel = the one whose demangled name contains
    <kernel_sub>, at grid <grid> (threads; 0 = the grid with the largest total time among the matches).  With the two PMC pass
    directories, `hbm_bytes_per_launch` is taken from the rows of EXACTLY that kernel name and grid -- a counter pass of another
    instantiation or grid is refused (round 2 carried round-1 counters of a different GEMV instantiation in this field)."""
    by = defaultdict(list)
    with open(find(d, "kernel_trace.csv")) as f:
        for r in csv.DictReader(f):
            if kernel_sub in r["Kernel_Name"]:
                g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
                by[(r["Kernel_Name"], g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    cand = {k: v for k, v in by.items() if int(grid) in (0, k[1])}
    if not cand:
        raise SystemExit("kernel %s (grid %s) not in %s" % (kernel_sub, grid, d))
    name, g = max(cand, key=lambda k: sum(cand[k]))
    durs = cand[(name, g)]
    try:
        with open(out_json) as f:
            data = json.load(f)
    except (OSError, ValueError):
        data = {}
    e = dict(kernel=short(name), grid=g, calls=len(durs), avg_us=round(sum(durs) / len(durs) / 1e3, 3),
             source="%s (rocprofv3 --kernel-trace, eager launches)" % stats_csv)
    if dfetch and dwrite:
        def collect(dd, counter):
            vals = []
            with open(find(dd, "counter_collection.csv")) as f:
                for r in csv.DictReader(f):
                    if r["Counter_Name"] == counter and r["Kernel_Name"] == name and int(r["Grid_Size"]) == g:
                        vals.append(float(r["Counter_Value"]))
            return vals
        fe, wr = collect(dfetch, "FETCH_SIZE"), collect(dwrite, "WRITE_SIZE")
        if not fe or not wr:
            raise SystemExit("REFUSED: no FETCH_SIZE / WRITE_SIZE rows for exactly kernel %s grid %d in %s / %s -- traffic of another "
                             "instantiation or grid is not attached" % (short(name), g, dfetch, dwrite))
        rd, wb = 2.0 * sum(fe) / len(fe) * 1024.0, sum(wr) / len(wr) * 1024.0
        e["hbm_bytes_per_launch"] = round(rd + wb)
        e["hbm_read_bytes"], e["hbm_write_bytes"] = round(rd), round(wb)
        e["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of the same command (%d / %d launches of this kernel and grid; KiB units, FETCH_SIZE x2 gfx950 wide-read correction)" % (len(fe), len(wr))
    data[key] = e
    with open(out_json, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print("wrote", out_json, key, e["avg_us"], "us x", len(durs), "grid", g, "hbm", e.get("hbm_bytes_per_launch"))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "sq":
        sq(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5:])
    elif sys.argv[1] == "roofline":
        roofline(*sys.argv[2:10])
    else:
        pmc(*sys.argv[2:8])
