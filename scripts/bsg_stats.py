"""Prints the product kernels' rows of the newest rocprofv3 kernel-stats file under each directory given."""
import csv, glob, os, sys
for d in sys.argv[1:]:
    fs = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if not fs: print(d, "no stats"); continue
    for r in csv.reader(open(fs[-1])):
        if any(k in r[0] for k in ("bsg_gemm", "bsg_pre")):
            print(d, r[0][:28].ljust(28), r[1], "avg %.1f us" % (float(r[3]) / 1e3), "min %.1f max %.1f" % (float(r[5]) / 1e3, float(r[6]) / 1e3))
