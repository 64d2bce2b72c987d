#!/usr/bin/env python3
"""Launches and kernel time of the LAST training step in a rocprofv3 kernel trace (step = the dispatches between the last two
vnp_adam_kernel launches), by kernel; library (vnp_*) and other (ATen / runtime copy and fill) kernels separately.
usage: launch_census.py <dir with *_kernel_trace.csv>"""
import csv, glob, re, sys
from collections import Counter
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
ends = [i for i, n in enumerate(names) if "vnp_adam" in n]
lo, hi = ends[-2] + 1, ends[-1] + 1


def short(n):
    n = n.replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", "")
    m = re.search(r"(CatArrayBatchedCopy\w*|FillFunctor<\w+>|CUDAFunctor_add<\w+>|\w+Functor<[^>]*>|reduce_kernel|_scatter_gather_elementwise_kernel|"
                  r"arange\w*|radixSort\w*|copyBuffer|fillBuffer\w*|multi_tensor_apply_kernel|direct_copy_kernel\w*|sqrt_kernel\w*|index_\w+)", n)
    return m.group(1) if (m and "vnp_" not in n) else n.split("(")[0][:80]


cnt, tim = Counter(), Counter()
for i in range(lo, hi):
    k = short(names[i])
    cnt[k] += 1
    tim[k] += (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1000.0
lib = [k for k in cnt if "vnp_" in k]
oth = [k for k in cnt if "vnp_" not in k]
print(f"step: {hi - lo} launches, {sum(tim.values()) / 1000:.3f} ms of kernel time; library {sum(cnt[k] for k in lib)} launches "
      f"{sum(tim[k] for k in lib) / 1000:.3f} ms; other {sum(cnt[k] for k in oth)} launches {sum(tim[k] for k in oth) / 1000:.3f} ms")
for group in (oth, lib):
    for k in sorted(group, key=lambda k: -tim[k]):
        print(f"{cnt[k]:5d} {tim[k]:9.1f} us  {k}")
    print()
