#!/usr/bin/env python3
"""Builds integration/_build/hyphy_hip: the reference HyPhy with its ComputeBlock routed through
libhyphy_hip.so.  Needs /root/reference (build container only); the result is a binary (git-ignored)
that travels to the GPU box.  Re-uses the reference objects already compiled by oracle/Makefile.ref —
only likefunc.cpp and tree.cpp are recompiled, from patched copies that are written to integration/_build/ and deleted
again after linking (HYPHY_HIP_KEEP_PATCHED=1 keeps them for debugging)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("HYPHY_REF", "/root/reference")
OUT = os.path.join(HERE, "_build")
sys.path.insert(0, HERE)
import adapter_blocks as AB  # noqa: E402


def splice(text, anchor, block, before=False, count=1):
    n = text.count(anchor)
    if n < count:
        raise SystemExit(f"anchor not found ({n}x): {anchor!r}")
    idx = text.index(anchor)
    if before:
        return text[:idx] + block + text[idx:]
    idx += len(anchor)
    return text[:idx] + block + text[idx:]


def replace_once(text, old, new):
    if text.count(old) != 1:
        raise SystemExit(f"anchor not unique ({text.count(old)}x): {old!r}")
    return text.replace(old, new)


def main():
    os.makedirs(OUT, exist_ok=True)
    # likefunc.cpp copy with the four adapter blocks.  _TheTree::flatParents is protected and
    # _LikelihoodFunction is not a friend; an upstream patch would add a one-line public accessor to
    # tree.h.  Headers are found next to their includers first, so a shadow copy of
    # tree.h cannot be injected from here: this translation unit relaxes `protected` while it reads the
    # project headers instead (no layout change, one TU only).
    lf = open(os.path.join(REF, "src/core/likefunc.cpp")).read()
    lf = splice(lf, '#include "batchlan.h"\n', "#define protected public\n", before=True)
    lf = splice(lf, '#include "vector.h"\n', "#undef protected\n" + AB.HELPERS)
    lf = splice(lf, "#ifdef MDSOCL\n    OCLEval[i].init(", AB.SETUP, before=True)
    lf = splice(lf, "void _LikelihoodFunction::DeleteCaches(bool all) {\n", AB.TEARDOWN)
    # The branch-cache state machine (computedLocalUpdatePolicy) keeps running on the host; its two hooks —
    # "build the cache of branch n after this pass" and "evaluate through the cache" — go to the device
    # (AB.COMPUTE).  The host-side scaling-factor backup/restore around it is harmless: the device path keeps
    # no sticky scalers.
    lf = splice(lf, "      hyFloat sum = 0.;\n\n      if (doCachedComp >= 3) {", AB.COMPUTE, before=True)
    lf = splice(lf, "_Matrix *_LikelihoodFunction::Optimize(_AssociativeList const *options) {\n", AB.OPTIMIZE)
    lf = splice(lf, "  if (lf->GetThreadCount() != 0)\n    return logL;\n", AB.BENCHMARK)
    # Compute(): all device partitions are enqueued before the first is collected (pre-pass + the loop's own call)
    lf = splice(lf, "    for (unsigned long partID = 0; partID < theTrees.lLength; partID++) {\n      if (blockDependancies.list_data[partID]) {\n        // has category variables",
                AB.PREPASS, before=True)
    lf = replace_once(lf, AB.LOOPCALL_OLD, AB.LOOPCALL_NEW)
    # r06: the category branch of Compute() takes the summed log-likelihood from the batched rate-class evaluation
    lf = splice(lf, AB.CATWANT_ANCHOR, AB.CATWANT, before=True)
    lf = replace_once(lf, AB.CATSUM_OLD, AB.CATSUM_NEW)
    src = os.path.join(OUT, "likefunc_hip.cpp")
    open(src, "w").write(lf)
    # tree.cpp copy: ExponentiateMatrices offers its queue to the adapter before the OpenMP exponentiation loop (mode B)
    tr = open(os.path.join(REF, "src/core/tree.cpp")).read()
    tr = splice(tr, "using namespace hyphy_global_objects;\n", AB.TREE_HOOK_DEF)
    tr = splice(tr, "  if (parallel.lLength) {\n    if (parallel.lLength == 1) {", AB.TREE_HOOK_CALL, before=True)
    tr = replace_once(tr, AB.TREE_SKIP_OLD, AB.TREE_SKIP_NEW)
    tsrc = os.path.join(OUT, "tree_hip.cpp")
    open(tsrc, "w").write(tr)
    # likefunc2.cpp copy (r06): PopulateConditionalProbabilities' weighted-sum loop collects its classes for ONE device evaluation
    l2 = open(os.path.join(REF, "src/core/likefunc2.cpp")).read()
    l2 = splice(l2, "using namespace hy_global;\n", AB.CAT_DECL) if "using namespace hy_global;\n" in l2 else splice(l2, '#include "likefunc.h"\n', AB.CAT_DECL)
    l2 = replace_once(l2, AB.CAT_BEGIN_ANCHOR, AB.CAT_BEGIN + AB.CAT_BEGIN_ANCHOR[len("  scalers.Populate(arrayDim, 0, 0);\n\n"):])
    l2 = replace_once(l2, AB.CAT_SKIP_ANCHOR, AB.CAT_SKIP)
    l2 = splice(l2, AB.CAT_END_ANCHOR, AB.CAT_END, before=True)
    l2src = os.path.join(OUT, "likefunc2_hip.cpp")
    open(l2src, "w").write(l2)
    # 3. compile that one file with the reference's flags (oracle/Makefile.ref) + -DHYPHY_HIP
    refobj = os.path.join(ROOT, "oracle", "_ref", "obj")
    if not os.path.isdir(refobj):
        subprocess.check_call(["make", "-f", os.path.join(ROOT, "oracle", "Makefile.ref"), "-j8"])
    flags = ("-std=c++17 -fsigned-char -O3 -fopenmp -w -mavx -mavx2 -mfma -D_SLKP_USE_AVX_INTRINSICS "
             "-D_SLKP_USE_FMA3_INTRINSICS -D__AFYP_REWRITE_BGM__ -D__UNIX__ -D__MP__ -D__MP2__ -DHYPHY_HIP "
             f"-D_HYPHY_LIBDIRECTORY_=\"/nonexistent\" -I{ROOT}/include "
             f"-I{REF}/src/core/include -I{REF}/src/contrib -I{REF}/src/lib/Link -I{REF}/src/new/include").split()
    obj = os.path.join(OUT, "likefunc_hip.o")
    tobj = os.path.join(OUT, "tree_hip.o")
    l2obj = os.path.join(OUT, "likefunc2_hip.o")
    procs = [subprocess.Popen(["g++"] + flags + ["-c", src, "-o", obj]), subprocess.Popen(["g++"] + flags + ["-c", tsrc, "-o", tobj]),
             subprocess.Popen(["g++"] + flags + ["-c", l2src, "-o", l2obj])]
    if any(p.wait() != 0 for p in procs):
        raise SystemExit("compilation of the patched copies failed")
    objs = [tobj, l2obj]
    for dp, _, files in os.walk(refobj):
        for f in files:
            if f.endswith(".o") and not (f in ("likefunc.o", "likefunc2.o", "tree.o") and dp.endswith("core")):
                objs.append(os.path.join(dp, f))
    libdir = os.path.join(ROOT, "hyphy_amd", "lib")
    exe = os.path.join(OUT, "hyphy_hip")
    subprocess.check_call(["g++", "-fopenmp", "-o", exe, obj] + sorted(objs) +
                          [f"-L{libdir}", "-lhyphy_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,$ORIGIN/../../hyphy_amd/lib",
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-ldl"])
    if not os.environ.get("HYPHY_HIP_KEEP_PATCHED"):
        # the patched copies are intermediate files: nothing derived from the reference's sources stays in the tree
        for f in (src, tsrc, l2src, obj, tobj, l2obj):
            os.remove(f)
    print("built", exe)


if __name__ == "__main__":
    main()
