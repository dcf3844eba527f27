"""Sanitizer builds of the threaded / pointer-handling HOST code (SURVEY.md section 5 "race detection / sanitizers"; GPU
AddressSanitizer is not available on this pool, so these are CPU builds and run in the default suite):

  * mex/gpz_mex.cpp + tests/stubs/mex_runtime.cpp under AddressSanitizer + UndefinedBehaviorSanitizer: every gateway command with good arguments and
    with each class of bad ones, against a host-only stand-in of the library that writes every output at its documented size and
    reads every input completely (tests/stubs/gpz_stub.cpp) - an output allocated too small or a short input handed through is a
    sanitizer abort;
  * the synchronisation core of the multi-device driver (gpz_amd/csrc/gpz_mgpu_sync.h: command hand-off, poisonable barrier,
    abort gate - the file gpz_mgpu.hip is built on) under ThreadSanitizer with a stub rank function: 8 threads x 1000 commands
    with failures injected at both exchange points, in the loopback and in the RCCL-like mode."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# CPU-only harness: this module, its byte code and its drivers are listed in .gpurunignore (GPU sanitizers are not available on the
# pool and its snapshots carry no sanitizer builds), so it never travels to the GPU box; it runs in the default CPU suite.
STUBS = os.path.join(ROOT, "tests", "stubs")
BUILD = os.path.join(ROOT, "build")


def _build(out, flags, srcs, incs):
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, out)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", *flags, *["-I" + i for i in incs], *srcs, "-o", exe, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return exe


def test_mex_gateway_under_address_and_undefined_behaviour_sanitizers():
    exe = _build("gateway_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
                 [os.path.join(ROOT, "mex", "gpz_mex.cpp"), os.path.join(STUBS, "mex_runtime.cpp"), os.path.join(STUBS, "gpz_stub.cpp"),
                  os.path.join(STUBS, "gateway_asan_driver.cpp")], [STUBS, os.path.join(ROOT, "include")])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "every command, good and bad arguments: ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_multi_device_synchronisation_core_under_thread_sanitizer():
    exe = _build("mgpu_sync_tsan", ["-fsanitize=thread"], [os.path.join(STUBS, "mgpu_sync_tsan.cpp")],
                 [os.path.join(ROOT, "gpz_amd", "csrc")])
    r = subprocess.run([exe, "1000"], capture_output=True, text=True, timeout=600)   # a deadlock shows up as the timeout
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert r.stdout.count(": ok") == 4 and "ThreadSanitizer" not in r.stderr, (r.stdout, r.stderr[-4000:])
    assert "8 ranks, 1000 commands" in r.stdout
