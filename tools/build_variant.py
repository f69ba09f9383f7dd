#!/usr/bin/env python
"""Build an experimental copy of libmsml_hip.so with extra -D flags on ONE source file:
    python tools/build_variant.py conv_halo.hip HALO_ABLATE_LOADS -> gpurun_variants/libmsml_HALO_ABLATE_LOADS.so
Select it at run time with MSML_LIB=<path> (msml_amd/_lib.py).  Used for the compile-time
ablations of DESIGN.md (loads / compute / epilogue removed one at a time)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def build_all(defs):
    """Every source with the extra defines (e.g. MSML_LDS_GUARD): variants/libmsml_<tag>.so"""
    import concurrent.futures
    outdir = os.path.join(ROOT, "variants", "obj_" + "_".join(d.replace("=", "") for d in defs))
    os.makedirs(outdir, exist_ok=True)
    srcs = sorted(f for f in os.listdir(ge.CSRC) if f.endswith(".hip"))

    def cc(f):
        obj = os.path.join(outdir, f + ".o")
        subprocess.run([ge.HIPCC] + ge.FLAGS + ["-D" + d for d in defs] + ["-c", os.path.join(ge.CSRC, f), "-o", obj], check=True)
        return obj
    with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(cc, srcs))
    lib = os.path.join(ROOT, "variants", "libmsml_%s.so" % "_".join(d.replace("=", "") for d in defs))
    subprocess.run([ge.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    print(lib)


def main():
    if sys.argv[1] == "--all":
        return build_all(sys.argv[2:])
    src, defs = sys.argv[1], sys.argv[2:]
    ge.build()
    outdir = os.path.join(ROOT, "variants")
    os.makedirs(outdir, exist_ok=True)
    tag = "_".join(d.replace("=", "") for d in defs)
    obj = os.path.join(outdir, src + "." + tag + ".o")
    subprocess.run([ge.HIPCC] + ge.FLAGS + ["-D" + d for d in defs] + ["-c", os.path.join(ge.CSRC, src), "-o", obj],
                   check=True)
    objs = [os.path.join(ge.OBJDIR, f) for f in sorted(os.listdir(ge.OBJDIR))
            if f.endswith(".o") and f != src + ".o"] + [obj]
    lib = os.path.join(outdir, "libmsml_%s.so" % tag)
    subprocess.run([ge.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    print(lib)


if __name__ == "__main__":
    main()
