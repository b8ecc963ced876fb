#!/usr/bin/env python3
"""Turns rocprofv3 CSV output into the small summaries committed under profiles/ (development tool).

  python tools/summarize_prof.py stats  <rocprof_dir> <out.csv> "<command line that was profiled>"
      per-kernel and per-(kernel, grid) launch statistics of the llmie kernels from *_kernel_trace.csv
  python tools/summarize_prof.py pmc    <fetch_dir> <write_dir> <out.csv> <out.json> <kernel substring> <grid>
      HBM bytes per launch from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units; FETCH_SIZE x2 on gfx950
      for wide coalesced streaming reads, MI355X_MICROARCH.md HBM section)
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


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        pmc(*sys.argv[2:8])
