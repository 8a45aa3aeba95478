#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Build oracle/_ref/libblitzar_ref.so from the reference sources.

The reference's CPU MSM path (everything reachable from oracle/ref/ref_driver.cc) is compiled
*from the sources where they lie* under /root/reference -- nothing is copied into this repo and
the reference's own build system (bazel/nix) is not used.  Recipe = SURVEY.md Appendix B:

  * closure: follow `#include "sxt/..."` recursively from ref_driver.cc, add each header's
    sibling .cc when it exists;
  * shim headers (oracle/ref/shim): <print>, <format> and boost/stacktrace.hpp (abort / logging
    paths only); for the fixed-base closure, whose headers also hold the CUDA variants of its
    functions: cuda_prelude.h (force-included: empty execution-space keywords), cuda_runtime.h /
    cuda.h (host memory, synchronous streams) and sxt/algorithm/iteration/for_each.h (a host loop
    in place of the one `<<<>>>` header of the closure);
  * /opt/rocm/lib/llvm/bin/clang++ -std=gnu++2b -O2 -DNDEBUG (gnu++2b so that
    std::signed_integral<__int128> holds, sxt/base/num/abs.h:32,43).

Outputs go only to oracle/_ref/ (git-ignored, but shipped to the GPU box by gpurun).
If /root/reference is absent (GPU box) this script is a no-op and the prebuilt .so is used.
"""
import concurrent.futures
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("BLITZAR_REFERENCE_DIR", "/root/reference")
OUT = os.path.join(os.path.dirname(HERE), "_ref")
OBJ = os.path.join(OUT, "obj")
CXX = os.environ.get("ORACLE_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-std=gnu++2b", "-O2", "-DNDEBUG", "-w", "-fPIC", "-I" + os.path.join(HERE, "shim"),
         "-I" + REF, "-include", "cstdint", "-include", "concepts", "-include",
         "initializer_list", "-include", "cstddef", "-include", "compare", "-include", "utility", "-include", "memory_resource",
         # the fixed-base closure's headers also hold CUDA variants of its functions: execution-space
         # keywords and a host-memory stand-in for the runtime (shim/cuda_prelude.h)
         "-include", os.path.join(HERE, "shim", "cuda_prelude.h")]
INC = re.compile(r'^\s*#\s*include\s+"(sxt/[^"]+)"', re.M)
# translation units of the closure that hold CUDA kernels (`__global__`, `<<<>>>`) and cannot be
# compiled on the host; the drivers define what they need from them (ref_inner_product.cc)
SKIP = {"sxt/scalar25/operation/inner_product.cc", "sxt/base/device/state.cc",
        "sxt/base/device/property.cc", "sxt/base/log/log_impl.cc", "sxt/base/log/setup.cc"}
SHIM_MTIME = max(os.path.getmtime(os.path.join(d, f))
                 for d, _, fs in os.walk(os.path.join(HERE, "shim")) for f in fs)
DRIVERS = ["ref_driver.cc", "ref_inner_product.cc", "ref_sumcheck.cc", "ref_fixed_base.cc"]


def closure(root_file):
    seen_h, srcs, todo = set(), [], [root_file]
    while todo:
        f = todo.pop()
        with open(f) as fh:
            text = fh.read()
        for h in INC.findall(text):
            if h in seen_h:
                continue
            seen_h.add(h)
            hp = os.path.join(REF, h)
            if not os.path.exists(hp):
                raise SystemExit(f"missing reference header {h}")
            todo.append(hp)
            cc = hp[:-2] + ".cc"
            if os.path.exists(cc) and os.path.relpath(cc, REF) not in SKIP:
                srcs.append(cc)
                todo.append(cc)
    return sorted(set(srcs))


def compile_one(src):
    tag = hashlib.sha1(src.encode()).hexdigest()[:12]
    obj = os.path.join(OBJ, os.path.basename(src)[:-3] + "." + tag + ".o")
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), SHIM_MTIME):
        return obj
    subprocess.run([CXX, *FLAGS, "-c", src, "-o", obj], check=True)
    return obj


def main():
    if not os.path.isdir(REF):
        print(f"[oracle/_ref] {REF} not present; keeping prebuilt artefacts (if any)")
        return 0
    os.makedirs(OBJ, exist_ok=True)
    drivers = [os.path.join(HERE, d) for d in DRIVERS]
    srcs = sorted(set(s for d in drivers for s in closure(d)))
    lib = os.path.join(OUT, "libblitzar_ref.so")
    shims = [os.path.join(d, f) for d, _, fs in os.walk(os.path.join(HERE, "shim")) for f in fs]
    if os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(s)
                                   for s in srcs + drivers + shims + [__file__]):
        print(f"[oracle/_ref] up to date ({len(srcs)} reference TUs)")
        return 0
    with concurrent.futures.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        objs = list(ex.map(compile_one, srcs + drivers))
    subprocess.run([CXX, "-shared", "-Wl,-z,defs", "-o", lib, *objs], check=True)
    print(f"[oracle/_ref] built {lib} from {len(srcs)} reference TUs")
    return 0


if __name__ == "__main__":
    sys.exit(main())
