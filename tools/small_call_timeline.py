"""GPU timeline of one host-level call out of a rocprofv3 --kernel-trace --memory-copy-trace run of tools/perf_host_phases.py:
   python tools/small_call_timeline.py <dir with *_kernel_trace.csv / *_memory_copy_trace.csv (+ .hdr when the files were cut)> [pair|parser]"""
import csv, glob, os, sys
d = sys.argv[1]; which = sys.argv[2] if len(sys.argv) > 2 else "pair"
def load(pat):
    f = glob.glob(os.path.join(d, pat))[0]
    rows = list(csv.reader(open(f)))
    hdr = rows[0] if rows[0][0] == "Kind" else open(f + ".hdr").read().strip().replace('"', "").split(",")
    return [dict(zip(hdr, r)) for r in rows if r[0] != "Kind"]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("flbgpu::", "")[:48]) for r in load("*kernel_trace.csv")]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"]) for r in load("*memory_copy_trace.csv")]
ev.sort()
key = "k_pg_emit" if which == "pair" else "k_parser_emit"
idx = [i for i, e in enumerate(ev) if key in e[2]]
i = idx[-5]
j = i
while j > 0 and "k_finish" not in ev[j - 1][2]: j -= 1
k = i
while "k_finish" not in ev[k][2]: k += 1
t0 = ev[j][0]
for e in ev[j:k + 1]:
    print("%9.1f %9.1f  dur %7.1f  %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2]))
