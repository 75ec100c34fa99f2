"""CPU: the C-ABI shim under AddressSanitizer ("ASan host build of the shim"; VERDICT r5 missing #6).
chatts_amd/build.py builds lib_asan/libchatts_amd_asan.so - the HOST half instrumented (-fsanitize=address -fno-gpu-sanitize: GPU ASan
needs xnack+ code objects, which the GPU pool refuses), the device code as shipped - and the host-logic tests, which drive the options
table, the error paths, the size / geometry queries and the argument validation of the entry points without a GPU, run against it with the
ASan runtime preloaded.  Any heap / stack / global overflow or use-after-free in that host code aborts the run with a report."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_logic_under_address_sanitizer():
    sys.path.insert(0, ROOT)
    from chatts_amd import build
    rt = build.asan_runtime()
    assert rt and os.path.exists(rt), "the ROCm clang's shared ASan runtime was not found"
    lib = build.build(asan=True)
    syms = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True).stdout
    assert "__asan_init" in syms and "__asan_report_load" in syms, "the library is not instrumented"
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=1", CHATTS_AMD_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_host_logic.py"), "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "passed" in r.stdout and "AddressSanitizer" not in tail, tail
    # the run really went through the instrumented build with the runtime mapped
    probe = ("import os, sys; sys.path.insert(0, %r); from chatts_amd import _lib; l = _lib.load(); "
             "m = open('/proc/self/maps').read(); print('MAPPED', 'libchatts_amd_asan.so' in m, 'libclang_rt.asan' in m, l.chatts_abi_version())" % ROOT)
    r = subprocess.run([sys.executable, "-c", probe], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert "MAPPED True True" in r.stdout, r.stdout + r.stderr
