#!/usr/bin/env python3
"""Dev tool (CPU only): what hipcc made of the main loops.  Compiles one .hip file of csrc/ to gfx950 assembly and prints, for every
loop that contains MFMAs, the instruction mix and every wait the compiler (or the source) placed in it — the things no test sees:
a dynamic index into a byte array of the kernel arguments becomes `global_load_sbyte` + `s_waitcnt vmcnt(0)` (which drains the LDS-DMA
ring in front of it), a control-flow join in front of an MFMA batch becomes `lgkmcnt(0..1)` (which waits for the fragment reads that
were meant to stay in flight).
usage: python tools/asm_loops.py [file.hip] [--kernel SUBSTR] [--dump]   (--dump prints the loop bodies)"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yolo_dual_amd", "csrc")


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "ds_read"
    if op.startswith("ds_"):
        return "ds_write"
    if op.startswith("buffer_load") or op.startswith("global_load") or op.startswith("flat_load") or op.startswith("scratch_load"):
        return "vmem_ld"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem_st"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    # quarter-rate vector instructions (16 cycles per wave instead of 4): 32-bit integer multiplies, the 64-bit multiply-add the
    # compiler uses for every `a * b + c` on ints, f64 and transcendentals.  A loop whose MFMA count is N x 16 cycles per wave can be
    # VALU-bound on a handful of these (round 5: wgrad3's per-DMA address arithmetic — 10 of them per stage — cost more than its MFMAs)
    if re.match(r"v_(mul_lo_u32|mul_hi_u32|mul_hi_i32|mad_u64_u32|mad_i64_i32|exp_|log_|rcp_|rsq_|sqrt_|sin_|cos_|.*_f64)", op):
        return "valu_slow"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file", nargs="?", default="igemm.hip")
    ap.add_argument("--kernel", default="")
    ap.add_argument("--dump", action="store_true")
    ap.add_argument("--asm", default="", help="use this .s file instead of compiling")
    ap.add_argument("-D", action="append", default=[])
    a = ap.parse_args()
    if a.asm:
        text = open(a.asm).read()
    else:
        out = "/tmp/_asm_loops.s"
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-unused-result",
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only", os.path.join(CSRC, a.file), "-o", out]
        cmd += ["-D" + d for d in a.D]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    lines = text.split("\n")
    # split into kernels
    kstart = [(i, m.group(1)) for i, l in enumerate(lines) if (m := re.match(r"^(_Z[A-Za-z0-9_]+):", l))]
    kstart.append((len(lines), None))
    names = subprocess.run(["c++filt"], input="\n".join(n for _, n in kstart[:-1]), capture_output=True, text=True).stdout.split("\n")
    for ki in range(len(kstart) - 1):
        s, e = kstart[ki][0], kstart[ki + 1][0]
        name = names[ki]
        if a.kernel and a.kernel not in name:
            continue
        body = lines[s:e]
        labels = {}
        for i, l in enumerate(body):
            m = re.match(r"^(\.LBB[0-9_]+):", l)
            if m:
                labels[m.group(1)] = i
        loops = []
        for i, l in enumerate(body):
            m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB[0-9_]+)", l) or re.match(r"^\s+s_branch\s+(\.LBB[0-9_]+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        # keep the innermost loops that contain MFMAs
        shown = False
        for (ls, le) in sorted(set(loops)):
            seg = body[ls:le + 1]
            if not any("v_mfma" in x for x in seg):
                continue
            if any(ls < s2 and e2 < le and any("v_mfma" in x for x in body[s2:e2 + 1]) for (s2, e2) in loops if (s2, e2) != (ls, le)):
                continue
            cnt = {}
            waits = []
            vm_in_loop = []
            for l in seg:
                t = l.strip()
                if not t or t.startswith(";") or t.startswith("."):
                    continue
                op = t.split()[0]
                c = classify(op)
                cnt[c] = cnt.get(c, 0) + 1
                if c == "wait":
                    waits.append(t.replace("s_waitcnt ", ""))
                if c == "vmem_ld" and " lds" not in t:
                    vm_in_loop.append(op)
            if not shown:
                print(f"== {name}")
                shown = True
            print(f"   loop @{ls}-{le} ({le - ls} lines): " + " ".join(f"{k}={v}" for k, v in sorted(cnt.items())))
            print(f"      waits: {waits}")
            mf, vs, vf = cnt.get("mfma", 0), cnt.get("valu_slow", 0), cnt.get("valu", 0)
            # (16 cycles per 16x16x32 / 32 per 32x32x16 matrix instruction is not told apart here: the estimate uses 16)
            print(f"      vector-ALU cycles per wave and iteration ~ {4 * vf + 16 * vs} (of which quarter-rate {16 * vs}) against ~{16 * mf} matrix cycles")
            if vm_in_loop:
                print(f"      !! register-destination VMEM loads inside the loop: {vm_in_loop}")
            if a.dump:
                print("\n".join(seg))


if __name__ == "__main__":
    main()
