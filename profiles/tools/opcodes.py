#!/usr/bin/env python3
"""Opcode histograms of the SHIPPED gfx950 code objects: python3 profiles/tools/opcodes.py [libfelics.so] > profiles/rNN/opcodes.txt

Extracts the code objects from the library (llvm-objdump --offloading), disassembles them and prints, per kernel, the static count of
every opcode with its issue class (profiles/r04/valu_rate.txt): C = 1.0 ns per wave64 instruction and SIMD (32-bit add / sub / logic /
shift-right, v_bitop3, the 16-bit VOP2 instructions), E = 1.7 ns (everything else vector), S = scalar, L = LDS, M = vector memory.
Static counts: the hot loops are unrolled straight-line code, so for k_pack_g / k_hist / k_scatter / k_assign_serial the histogram of
the kernel is close to the mix it executes; felics_amd/_build/libfelics.so by default.  Writes nothing into the repository but its stdout.
"""
import collections, glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin/"
CHEAP = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32",
         "v_ashrrev_i32", "v_bitop3_b32", "v_min_u16", "v_max_u16", "v_min_i16", "v_max_i16", "v_add_u16", "v_sub_u16", "v_lshlrev_b16",
         "v_lshrrev_b16", "v_ashrrev_i16", "v_mul_lo_u16", "v_mov_b64"}


def klass(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.startswith("v_"):
        if op.endswith(("_sdwa", "_dpp")):
            return "E"
        if op.endswith("_e64") and base not in ("v_bitop3_b32",):
            return "E"  # (VOP3 forms of the cheap instructions measured 1.7 ns with an SGPR operand; counted expensive)
        return "C" if base in CHEAP else "E"
    if op.startswith("s_"):
        return "S"
    if op.startswith("ds_"):
        return "L"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "M"
    return "?"


def histograms(lib):
    """{demangled kernel name: Counter(opcode -> static count)} of the felics:: kernels in the library's gfx950 code objects"""
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        subprocess.check_call(["cp", lib, local])
        subprocess.check_call([LLVM + "llvm-objdump", "--offloading", local], stdout=subprocess.DEVNULL, cwd=tmp)
        kernels = collections.OrderedDict()
        for co in sorted(glob.glob(local + ".*gfx950*")):
            dis = subprocess.check_output([LLVM + "llvm-objdump", "-d", "--demangle", co], text=True)
            cur = None
            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
                if m:
                    name = m.group(1)
                    cur = kernels.setdefault(name, collections.Counter()) if "felics::" in name else None
                    continue
                if cur is None:
                    continue
                m = re.match(r"^\s+([a-z][a-z0-9_]+)\b", line)
                if m and not line.strip().startswith("//"):
                    cur[m.group(1)] += 1
    return kernels


def cheap_share(ctr):
    """(vector instructions, share of them in the 1.0 ns class)"""
    cls = collections.Counter()
    for op, n in ctr.items():
        cls[klass(op)] += n
    valu = cls["C"] + cls["E"]
    return valu, cls["C"] / max(valu, 1)


def main():
    lib = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "felics_amd", "_build", "libfelics.so"))
    want = sys.argv[2:] or ["k_pack_t<unsigned char>", "k_front<unsigned char, unsigned char>", "k_spine3<unsigned char>", "k_assign3<unsigned char>", "k_enum"]
    kernels = histograms(lib)
    print("# static opcode histograms of %s" % os.path.relpath(lib, ROOT))
    print("# class: C = 1.0 ns, E = 1.7 ns per wave64 instruction and SIMD (profiles/r04/valu_rate.txt); S scalar, L LDS, M vector memory")
    for name, ctr in kernels.items():
        short = name.split("felics::")[1].split("(")[0]
        if not any(short.startswith(w.split("(")[0]) for w in want):
            continue
        cls = collections.Counter()
        for op, n in ctr.items():
            cls[klass(op)] += n
        valu = cls["C"] + cls["E"]
        print("\n== %s ==" % short)
        print("vector %d (cheap %d = %.0f %%, expensive %d), scalar %d, LDS %d, vector memory %d; additive price %.2f ns per vector instruction"
              % (valu, cls["C"], 100.0 * cls["C"] / max(valu, 1), cls["E"], cls["S"], cls["L"], cls["M"],
                 (cls["C"] * 1.0 + cls["E"] * 1.72) / max(valu, 1)))
        for op, n in ctr.most_common(40):
            print("  %5d  %s  %s" % (n, klass(op), op))


if __name__ == "__main__":
    main()
