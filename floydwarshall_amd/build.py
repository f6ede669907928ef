"""In-tree build of libfwx.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.

    python -m floydwarshall_amd.build        # build if sources are newer than the library
    python -m floydwarshall_amd.build -f     # force

The library is git-ignored (history stays source-only) but travels with the repo snapshot to the
GPU box, where the tests load exactly this file.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libfwx.so")
CLI = os.path.join(PKG, "fwx_cli")

HIP_SOURCES = ["fwx_kernels.hip", "fwx_fused.hip", "fwx_api.hip", "fwx_multi.hip"]
CXX_SOURCES = []  # host mirror sources are appended below when present
HOST_DIR = os.path.join(CSRC, "host")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",          # never fuse; the path has no add anyway
    "-fno-fast-math", "-fno-gpu-rdc",
    "-Wall", "-Wextra", "-Wno-unused-parameter",
    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
]


def _sources():
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    if os.path.isdir(HOST_DIR):
        srcs += sorted(os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR)
                       if f.endswith(".cpp"))
    return srcs


def _deps():
    deps = _sources() + [os.path.join(ROOT, "include", f)
                         for f in os.listdir(os.path.join(ROOT, "include"))]
    deps.append(os.path.join(CSRC, "cli", "fwx_cli.cpp"))
    for d in (CSRC, HOST_DIR):
        if os.path.isdir(d):
            deps += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".h", ".hpp"))]
    return deps


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def _headers():
    hs = [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    for d in (CSRC, HOST_DIR):
        if os.path.isdir(d):
            hs += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".h", ".hpp"))]
    return hs


def build_lib(force=False, verbose=False):
    """One object per translation unit (compiled in parallel, rebuilt only when the source or a
    header is newer), then one link.  Objects live under build/obj (git-ignored, not shipped)."""
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libfwx can only be built with the ROCm toolchain")
    objdir = os.path.join(ROOT, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in _headers())
    compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    jobs, objs = [], []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj) or \
            os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time)
        if stale:
            jobs.append([hipcc] + compile_flags + ["-x", "hip", "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + objs +
        ["-o", LIB + ".tmp", "-ldl"])
    os.replace(LIB + ".tmp", LIB)
    build_cli(hipcc, verbose)
    return LIB


def build_cli(hipcc, verbose=False):
    """fwx_cli: the reference's Main loop (src/app/Main.hs) over libfwx; host-only C++."""
    src = os.path.join(CSRC, "cli", "fwx_cli.cpp")
    if not os.path.exists(src):
        return None
    cmd = [hipcc, "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-o", CLI,
           "-L" + PKG, "-lfwx", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return CLI


def build_variant(name, defines):
    """An experimental build of the same ABI with extra -D switches: build/variants/libfwx_<name>.so, selected
    at run time with FWX_LIB_PATH (floydwarshall_amd/_lib.py) -- how two kernel variants are measured
    against each other on ONE GPU box in one gpurun call.  Never shipped: build/ is git-ignored."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    outdir = os.path.join(ROOT, "build", "variants")
    objdir = os.path.join(outdir, "obj_" + name)
    os.makedirs(objdir, exist_ok=True)
    compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(defines)
    objs, jobs = [], []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        jobs.append([hipcc] + compile_flags + ["-x", "hip", "-c", src, "-o", obj])
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
        list(ex.map(lambda c: subprocess.run(c, check=True), jobs))
    lib = os.path.join(outdir, "libfwx_%s.so" % name)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", lib, "-ldl"],
                   check=True)
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith("-D")]))
    else:
        print(build_lib(force="-f" in sys.argv, verbose=True))
