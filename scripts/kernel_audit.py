#!/usr/bin/env python3
"""Compiler-side audit of every kernel of libvnpcc (no GPU needed): registers, scratch, occupancy and the 64-bit VECTOR address arithmetic
left in its code — the round-5 find was that a wave-uniform index the compiler cannot prove uniform (threadIdx.x >> 6) turns every access
of a kernel into per-lane 64-bit multiplies / adds held in VGPRs; `__builtin_amdgcn_readfirstlane` on the index moved them to the scalar unit
(DESIGN.md §0).  usage: python scripts/kernel_audit.py [--all]      (default: kernels with scratch, at an occupancy step, or address-heavy)"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vn_pointcloudcompletion_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FILT = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
show_all = "--all" in sys.argv


def demangle(n):
    try:
        return subprocess.check_output([FILT, n]).decode().strip().split("(")[0]
    except Exception:
        return n


with tempfile.TemporaryDirectory() as tmp:
    procs = []
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
        procs.append((src, out, subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src],
                                                 stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
    rows = []
    for src, out, p in procs:
        if p.wait() != 0:
            print("failed to compile", src)
            continue
        text = open(out).read()
        for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", text, re.S | re.M):
            name, body = m.group(1), m.group(0)

            def setv(key):
                mm = re.search(r"\.set " + re.escape(name) + r"\." + key + r", (\d+)", text)
                return int(mm.group(1)) if mm else 0
            vg, ag, scratch = setv("num_vgpr"), setv("num_agpr"), setv("private_seg_size")
            occ = re.search(re.escape(name) + r":.*?; Occupancy: (\d+)", text, re.S)
            valu = len(re.findall(r"^\s+v_", body, re.M))
            addr64 = len(re.findall(r"^\s+(v_lshl_add_u64|v_mad_u64_u32|v_mad_i64_i32|v_addc_co_u32|v_add_co_u32)", body, re.M))
            rows.append((os.path.basename(src), demangle(name), vg + ag, scratch, int(occ.group(1)) if occ else -1, valu, addr64))
    steps = ((65, 72), (97, 104), (129, 136), (169, 176), (257, 272))
    print(f"{'file':18s} {'kernel':86s} {'regs':>5s} {'scratch':>7s} {'occ':>3s} {'VALU':>6s} {'64-bit addr ops':>15s}")
    for f, k, regs, scratch, occ, valu, a64 in sorted(rows, key=lambda r: (-r[3], -r[6] / max(r[5], 1))):
        flag = scratch > 0 or any(lo <= regs <= hi for lo, hi in steps) or (valu > 50 and a64 / valu > 0.12)
        if show_all or flag:
            print(f"{f:18s} {k[:86]:86s} {regs:5d} {scratch:7d} {occ:3d} {valu:6d} {a64:7d} ({100.0 * a64 / max(valu, 1):.0f} %)")
