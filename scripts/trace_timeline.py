"""Summarise a rocprofv3 --kernel-trace CSV: timeline of the LAST eigensolve (from its first eig_init_q_kernel on).
usage: python3 scripts/trace_timeline.py <kernel_trace.csv> [out.txt]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
def nm(r): 
    n = r["Kernel_Name"]; n = n.replace("flgp::", "").replace("void ", ""); return n.split("(")[0][:40]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "bs_seed_kernel" in r["Kernel_Name"] or "eig_init_q" in r["Kernel_Name"]]
# a solve begins at bs_seed (block sparse set-up) when present
seeds = [i for i, r in enumerate(rows) if "bs_seed_kernel" in r["Kernel_Name"] or "bsg_scan_kernel" in r["Kernel_Name"]]
first = seeds[-1] if seeds else starts[-1]
sel = rows[first:]
t0 = int(sel[0]["Start_Timestamp"])
tend = max(int(r["End_Timestamp"]) for r in sel)
print("solve span %.3f ms, %d kernels" % ((tend - t0) / 1e6, len(sel)), file=out)
busy = collections.Counter(); cnt = collections.Counter()
prev_end = t0; gaps = 0; covered = 0; cur_end = t0
for r in sel:
    s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy[nm(r)] += e_ - s_; cnt[nm(r)] += 1
    if s_ > cur_end: gaps += s_ - cur_end
    cur_end = max(cur_end, e_)
print("idle (no kernel running) %.3f ms" % (gaps / 1e6), file=out)
for k, v in busy.most_common(40):
    print("%-42s %5d %9.3f ms  avg %7.2f us" % (k, cnt[k], v / 1e6, v / 1e3 / cnt[k]), file=out)
print("---- timeline (start us, dur us, gap-before us, queue, name)", file=out)
cur_end = t0
for r in sel:
    s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %7.1f %6.1f q%s %s" % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, (s_ - cur_end) / 1e3, r.get("Queue_Id", "?"), nm(r)), file=out)
    cur_end = max(cur_end, e_)
