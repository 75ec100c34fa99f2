"""Build a VARIANT of libchatts_amd.so next to the shipped one: same sources, extra -D flags (A/B arms), or a source swapped for a
diagnostic twin (--replace gemm_ring.hip=tools/probes/gemm_ring_probe.hip: the prefill kernel with its timeline probe and ablations).
    python tools/build_variant.py NAME [-DFOO=1 ...] [--only a.hip,b.hip] [--replace x.hip=path]   ->  chatts_amd/lib/variants/libchatts_amd_NAME.so
Point the Python side at it with CHATTS_AMD_LIB=<that path>.  Variant libraries are git-ignored like the shipped build and travel
with the gpurun snapshot."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import build as b  # noqa: E402


def build_variant(name, defines, only=None, replace=None):
    out_dir = os.path.join(b.LIBDIR, "variants", name)
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(b.LIBDIR, "variants", f"libchatts_amd_{name}.so")
    hipcc = b._hipcc()

    def one(src):
        obj = os.path.join(out_dir, src.replace(".hip", ".o"))
        # sources a variant does not touch are taken from the shipped build's objects
        path = os.path.join(b.CSRC, src)
        if replace and src in replace:
            path = os.path.abspath(replace[src])
        elif only is not None and src not in only:
            shipped = os.path.join(b.LIBDIR, src.replace(".hip", ".o"))
            if os.path.exists(shipped):
                return shipped
        cmd = [hipcc, *b.FLAGS, *b.EXTRA_FLAGS.get(src, []), *defines, "-I", b.CSRC, "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(one, b.SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-ldl"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


if __name__ == "__main__":
    args = sys.argv[1:]
    only = None
    if "--only" in args:
        i = args.index("--only")
        only = set(args[i + 1].split(","))
        del args[i:i + 2]
    replace = {}
    while "--replace" in args:
        i = args.index("--replace")
        k, v = args[i + 1].split("=", 1)
        replace[k] = v
        del args[i:i + 2]
    if replace and only is None:
        only = set()                     # everything else from the shipped objects
    b.build()
    print(build_variant(args[0], args[1:], only, replace))
