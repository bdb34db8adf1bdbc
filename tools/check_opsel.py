"""Lint of the built library's gfx950 code object for the packed-fp32 op_sel hazard (DESIGN.md section 4d, round 4;
reproducer tools/pk_opsel_hazard.hip): a v_pk_{mul,add,fma}_f32 whose op_sel takes the LOW result's operand from the HIGH
dword of a VGPR pair in src1 computes lanes 48-63 with that operand read as zero now and then while the SIMD's matrix pipe is
busy - with ANY wave's matrix instructions, so every kernel of the code object is checked (the library is built with
-fno-slp-vectorize, which is what keeps the form out of the kernels without matrix instructions).  Lists the packed-fp32
instructions with an op_sel bit on a VGPR source.
Usage: python tools/check_opsel.py [path/to/libfnoengine.so] [--all]      (exit status 1 if a hazardous form is present)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PK = re.compile(r"(v_pk_(?:mul|add|fma)_f32) (\S+), (.+?) op_sel:\[([0-9,]+)\]")


def disassemble(lib):
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        os.symlink(os.path.abspath(lib), local)
        subprocess.check_call([OBJDUMP, "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        assert co, "no gfx950 code object in " + lib
        return subprocess.check_output([OBJDUMP, "-d", os.path.join(tmp, co[0])], text=True)


def scan(lib):
    """{kernel: (n_mfma, n_src1_vgpr_crossed, n_other_vgpr_crossed, example)}"""
    res = collections.OrderedDict()
    cur = None
    for line in disassemble(lib).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = m.group(1)
            res[cur] = [0, 0, 0, ""]
            continue
        if cur is None:
            continue
        if "v_mfma_" in line:
            res[cur][0] += 1
        m = PK.search(line)
        if m:
            srcs = [s.strip() for s in m.group(3).split(", ")]
            sel = [int(x) for x in m.group(4).split(",")]
            crossed = [i for i, b in enumerate(sel) if b and srcs[i].startswith("v")]
            if 1 in crossed:
                res[cur][1] += 1
                res[cur][3] = res[cur][3] or line.split("//")[0].strip()
            elif crossed:
                res[cur][2] += 1
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pde_policylearning_amd", "libfnoengine.so")
    res = scan(lib)
    bad = 0
    for k, (nm, n1, no, ex) in res.items():
        if n1 or ("--all" in sys.argv and no):
            print(f"{nm:5d} mfma {n1:5d} src1-crossed {no:5d} src0/2-crossed  {k[:110]}   {ex}")
        bad += n1      # every kernel counts: the partner wave's matrix instructions may belong to another kernel on the same CU
    print(f"{sum(1 for v in res.values() if v[0])} kernels with matrix instructions of {len(res)}; hazardous (src1-crossed, VGPR) packed-fp32 instructions in the code object: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
