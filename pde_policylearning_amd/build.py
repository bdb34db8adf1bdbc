"""Build libfnoengine.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "fno_abi.hip")
OUT = os.environ.get("FNO_LIB_PATH") or os.path.join(HERE, "libfnoengine.so")      # (FNO_LIB_PATH: experiment builds of tools/)
DEPS = [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))] + [
    os.path.join(os.path.dirname(HERE), "include", "fnoengine.h"), os.path.abspath(__file__)]      # (the flags live in this file)


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build fnoengine)")


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(d) <= t for d in DEPS)


def build(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    # -fno-slp-vectorize: the SLP vectorizer is what turns scalar complex arithmetic into v_pk_*_f32 with crossed op_sel
    # operands - the form that misreads lanes 48-63 beside another wave's matrix instructions on gfx950 (fno_dev.h,
    # tools/pk_opsel_hazard.hip).  Without it NO kernel of the library carries the form (tools/check_opsel.py, linted in
    # tests/test_abi_and_host.py); cost measured in round 5: +0.5 % (config 2) to -1.4 % (PINO fine-tuning) fields/s.
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
           "-Wno-unused-value", "-Wno-unused-result"] + os.environ.get("FNO_EXTRA_FLAGS", "").split() + ["-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
