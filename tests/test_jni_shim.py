"""The JNI shim (integration/) is source for a BoofCV maintainer -- there is no JDK here -- but it must be COMPLETE and current:
generated from include/boofhip.h, one native per export, and syntactically valid C against the JNI signatures."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_generated_files_are_current_and_complete():
    import gen_jni
    fns = gen_jni.parse_header(open(gen_jni.HEADER).read())
    c_src, j_src = gen_jni.gen(fns)
    assert open(gen_jni.OUT_C).read() == c_src, "integration/jni/boofhip_jni.c is stale: run scripts/gen_jni.py"
    assert open(gen_jni.OUT_JAVA).read() == j_src, "integration/java/boofcv/hip/BoofHip.java is stale: run scripts/gen_jni.py"
    from boofcv_amd import _lib
    names = {n for _, n, _ in fns}
    assert names == set(_lib.SIGNATURES), names ^ set(_lib.SIGNATURES)           # every export of the header ...
    for n in names:                                                                # ... is called by exactly one JNI function and declared native
        assert len(re.findall(r"\b%s\(" % n, c_src)) == 1, n
        assert ("// %s\n" % n) in j_src, n


def test_shim_is_valid_c_against_the_jni_signatures():
    for src in ("boofhip_jni.c", "boofhip_jni_buffers.c"):   # the generated shim and its hand-written companion (page-locked direct buffers)
        cmd = ["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Werror", "-Wno-unused-parameter", "-I", os.path.join(ROOT, "tests", "jni_syntax"), "-I", os.path.join(ROOT, "include"),
               os.path.join(ROOT, "integration", "jni", src)]
        p = subprocess.run(cmd, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
    # every native of the companion's Java class has its C function
    j = open(os.path.join(ROOT, "integration", "java", "boofcv", "hip", "PinnedBuffersHip.java")).read()
    c = open(os.path.join(ROOT, "integration", "jni", "boofhip_jni_buffers.c")).read()
    natives = set(re.findall(r"native \S+ (\w+)\(", j))
    assert natives == set(re.findall(r"^BHIP_JNI\(\w+, (\w+)\)", c, flags=re.M)), natives   # (line starts: not the macro's own definition)


def test_provider_sources_call_existing_natives():
    """every BoofHip.xxx( call in the hand-written provider classes names a generated native (or check)"""
    j_src = open(os.path.join(ROOT, "integration", "java", "boofcv", "hip", "BoofHip.java")).read()
    natives = set(re.findall(r"native \S+ (\w+)\(", j_src)) | {"check"}
    d = os.path.join(ROOT, "integration", "java", "boofcv", "hip")
    used = set()
    sources = sorted(f for f in os.listdir(d) if f.endswith(".java"))
    for f in sources:
        if f != "BoofHip.java":
            used |= set(re.findall(r"BoofHip\.(\w+)\(", open(os.path.join(d, f)).read()))
    assert used and used <= natives, used - natives
    # every provider class INTEGRATION.md names exists as a source file, and balances its braces (there is no javac here)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    named = set(re.findall(r"\b(\w+Hip\w*)\b", doc)) - {"BoofHip"}
    have = {f[:-5] for f in sources}
    for f in sources:   # nested classes count (NonMaxHip lives inside BoofHipOverrides)
        have |= set(re.findall(r"\bclass (\w+)", open(os.path.join(d, f)).read()))
    assert named <= have, "INTEGRATION.md names provider classes that have no source: %s" % sorted(named - have)
    for f in sources:
        src = open(os.path.join(d, f)).read()
        assert src.count("{") == src.count("}") and src.count("(") == src.count(")"), f
    # the install() of the overrides assigns every hook field of the four BOverride classes the library implements
    ov = open(os.path.join(d, "BoofHipOverrides.java")).read()
    for hook in ("BOverrideConvolveImage.horizontal", "BOverrideConvolveImage.vertical", "BOverrideConvolveImage.convolve", "BOverrideConvolveImageNormalized.horizontal",
                 "BOverrideConvolveImageNormalized.vertical", "BOverrideBlurImageOps.mean", "BOverrideBlurImageOps.median", "BOverrideBlurImageOps.gaussian",
                 "BOverrideFactoryFeatureExtractor.nonmax"):
        assert (hook + " =") in ov, hook
