"""Build libchatts_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.  hipcc cross-compiles
without a GPU, so this also runs in the CPU-only build container (`__graft_entry__.build()`).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libchatts_amd.so")
SOURCES = ["api.hip", "ts_frontend.hip", "gemv.hip", "gemm.hip", "gemm_ring.hip", "gemm_f16q.hip", "gemm_fp8.hip", "elementwise.hip", "sampler.hip", "attention.hip", "tp.hip", "decoder.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-file extra flags (none at the moment)
EXTRA_FLAGS = {}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _digest():
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in sorted(os.listdir(root)):
            if f.endswith((".hip", ".h")):
                h.update(f.encode())
                h.update(open(os.path.join(root, f), "rb").read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


# AddressSanitizer on the HOST half of the library only ("ASan host build of the shim"): argument validation, the
# options table, error strings, geometry choices, handle bookkeeping - everything the C-ABI does before a kernel is enqueued.  GPU ASan
# needs xnack+ code objects, which this pool refuses; -fno-gpu-sanitize keeps the device code as shipped.
ASAN_FLAGS = ["--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fsanitize=address",
              "-fno-gpu-sanitize", "-shared-libsan", "-fno-omit-frame-pointer"]
ASAN_LIBDIR = os.path.join(HERE, "lib_asan")
ASAN_LIB = os.path.join(ASAN_LIBDIR, "libchatts_amd_asan.so")


def asan_runtime():
    """path of the shared ASan runtime of the ROCm clang that compiled the library (to LD_PRELOAD into python)"""
    import glob
    c = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return c[-1] if c else None


def build(force=False, verbose=False, asan=False):
    """Compile every HIP source for gfx950 and link the shared library.  Returns the .so path.  asan=True: the host-ASan variant
    (lib_asan/libchatts_amd_asan.so; load it with CHATTS_AMD_LIB=<path> and LD_PRELOAD=asan_runtime(): tests/test_asan_host.py)."""
    libdir, lib, flags = (ASAN_LIBDIR, ASAN_LIB, ASAN_FLAGS) if asan else (LIBDIR, LIB, FLAGS)
    os.makedirs(libdir, exist_ok=True)
    stamp = os.path.join(libdir, "build.stamp")
    dig = _digest() + ("/asan" if asan else "")
    if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read() == dig:
        return lib
    hipcc = _hipcc()
    objs = []

    def compile_one(src):
        obj = os.path.join(libdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *flags, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-ldl"]
    if asan:
        cmd += ["-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, asan="--asan" in sys.argv))
