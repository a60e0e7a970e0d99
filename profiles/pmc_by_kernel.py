"""Per-kernel sums of the counters in a rocprofv3 --pmc output directory (all kernels, not only the trace kernel).
    python profiles/pmc_by_kernel.py <dir with pass subdirectories>"""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-70:]
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[name].add((path, row["Dispatch_Id"]))
for name, c in sorted(acc.items()):
    if "mirror" not in name and "traceKernel" not in name:
        continue
    print(name)
    for k in sorted(c):
        print("   %-28s %.6g" % (k, c[k]))
    if c.get("SQ_INSTS_VALU") and c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
        # lanes busy per VALU instruction: thread-cycles / (4 cycles per wave64 instruction x 64 lanes x instructions)
        print("   lane utilisation %.3f" % (c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])))
