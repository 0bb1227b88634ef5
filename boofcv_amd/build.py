"""Builds boofcv_amd/libboofhip.so (hand-written HIP kernels + the C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU.  -ffp-contract=off keeps Java's "no fused multiply-add" arithmetic in every kernel;
fp32 divide/sqrt stay correctly rounded.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libboofhip.so")
LIB_EXPERIMENTS = os.path.join(HERE, "libboofhip_exp.so")   # -DBHIP_EXPERIMENTS: ablation / stamp / tile-variant switches (scripts/ only, never shipped)
ARCH = "gfx950"

FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + sorted(glob.glob(os.path.join(CSRC, "*.cpp")))


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "boofhip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, experiments=False):
    """experiments=True builds libboofhip_exp.so with the timing-experiment switches compiled in (select it with BHIP_LIB=...);
    the default product library contains none of them (tests/test_cabi_symbols.py)."""
    lib = LIB_EXPERIMENTS if experiments else LIB
    extra = []
    if experiments and os.environ.get("BHIP_EXP_DEFS"):
        # compile-time variants for A/B runs: BHIP_EXP_DEFS="-DBHIP_TAP_AUX=1" BHIP_EXP_TAG=sc0 -> libboofhip_exp_sc0.so
        extra = os.environ["BHIP_EXP_DEFS"].split()
        lib = os.path.join(HERE, "libboofhip_exp_%s.so" % os.environ.get("BHIP_EXP_TAG", "variant"))
        force = True
    if not force and not needs_build(lib):
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    objdir = os.path.join(HERE, ("build_exp_" + os.environ.get("BHIP_EXP_TAG", "variant")) if extra else "build_exp" if experiments else "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [hipcc, "--offload-arch=" + ARCH] + FLAGS + (["-DBHIP_EXPERIMENTS"] if experiments else []) + extra + ["-c", src, "-o", obj]
        if src.endswith(".cpp"):
            cmd.insert(1, "-x")
            cmd.insert(2, "hip")
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, experiments="--experiments" in sys.argv))
