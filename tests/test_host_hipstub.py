"""CPU box: the product's host side BEHIND the planner -- device states, table blocks and their uploads, the frame-table ring, staging of host frames, slices, batches
sharded over several GPUs, teardown -- run over tests/hipstub, a test double of the HIP runtime ("device" memory = bounds-checked host memory carrying a fake ordinal;
launches validated, logged and dropped).  NOTHING is computed and no parity claim is made here: these tests hold the host logic (what is uploaded where, what is
launched on which GPU, what is cached, what is freed), and the same driver under the sanitizer builds is the hunt of DESIGN.md 8 (tools/hipstub_hunt.py).

VERDICT r05 item 10's fake-ordinal peer test: a batch over four ordinals launches each frame pair on the GPU that owns it, every GPU gets its own copy of the tables
once, and a second call uploads nothing."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUBDIR = os.path.join(ROOT, "tests", "hipstub")


@pytest.fixture(scope="module")
def stub():
    r = subprocess.run(["make", "-s", "-C", STUBDIR, "all"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return os.path.join(STUBDIR, "libhipstub.so")


def _run(stub, code_or_args, tmp_path, ndev=4, script=None, defer=0, rccl=None):
    log = tmp_path / "hipstub.log"
    env = dict(os.environ, LD_PRELOAD=stub + (":" + os.path.join(STUBDIR, "librccl.so.1") if rccl is not None else ""), HIPSTUB_DEFER=str(defer), RCCLSTUB_FAIL=rccl or "",
               SWS_HIP_NO_TORCH="1" if rccl is not None else "0",        # (torch brings its own librccl.so.1: the double must be the one dlopen() finds)
               HIPSTUB_DEVICES=str(ndev), HIPSTUB_LOG=str(log), PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]))
    env.pop("SWS_HIP_LIBRARY", None)
    cmd = [sys.executable, script] + code_or_args if script else [sys.executable, "-c", code_or_args]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    return r, (log.read_text() if log.exists() else "")


def test_random_conversions_over_the_stub(stub, tmp_path):
    """every generator of the GPU suite's random tests, a few cases each: no call fails, every launch is well-formed, no copy leaves its allocation, the table blocks read
    back as uploaded, batches run on the GPUs that own their frames, every device block is returned"""
    r, _ = _run(stub, ["14", "20260930"], tmp_path, script=os.path.join(ROOT, "tools", "hipstub_hunt.py"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert "'failed_calls': 0" in last and "'bad_tables': 0" in last and "'wrong_gpu': 0" in last and last.endswith("closed: 0"), last


def test_random_conversions_over_the_laziest_gpu(stub, tmp_path):
    """the same with HIPSTUB_DEFER=1: nothing queued on a stream runs before the host forces it, copies from pinned memory read their source then -- the frame-table
    ring must not hand a span out again before the copy that reads it ran, nothing may be freed under queued work, every context must drain what it queued"""
    r, _ = _run(stub, ["10", "77"], tmp_path, script=os.path.join(ROOT, "tools", "hipstub_hunt.py"), defer=1)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert "'failed_calls': 0" in lines[-1] and "'bad_tables': 0" in lines[-1] and lines[-1].endswith("closed: 0"), lines[-1]
    late, pinned, left = (int(lines[-2].split(k)[0].split()[-1]) for k in (" operations ran late", " copies from pinned", " still queued"))
    assert late > 1000 and pinned > 50 and left == 0, lines[-2]


PEERS = r"""
import sys
sys.argv = ["x", "1", "1"]
import hipstub_hunt as H
from librempeg_amd import SwsContext, SWS_BICUBIC, SWS_BITEXACT
c = SwsContext(640, 360, "yuv420p", 1280, 720, "rgb24", SWS_BICUBIC | SWS_BITEXACT)
owners = [0, 1, 2, 3, 3, 2, 1, 0]
srcs = [H.StubFrame("yuv420p", 640, 360, g) for g in owners]
dsts = [H.StubFrame("rgb24", 1280, 720, g) for g in owners]
for call in range(2):
    assert c.scale_frames(srcs, dsts) == 8
    c.sync()
    print("CALL", call, H.STUB.hipstub_launches(), H.STUB.hipstub_copies(), flush=True)
bad, text = c.debug_check()
assert bad == 0, text
print("CHECK", text)
c.close()
for f in srcs + dsts:
    f.free()
print("LIVE", H.STUB.hipstub_live_blocks())
"""


def test_batch_over_four_fake_ordinals(stub, tmp_path):
    r, log = _run(stub, PEERS, tmp_path)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = dict((ln.split()[0] + ln.split()[1] if ln.startswith("CALL") else ln.split()[0], ln) for ln in r.stdout.splitlines() if ln[:4] in ("CALL", "CHEC", "LIVE"))
    l0, c0 = map(int, out["CALL0"].split()[2:4])
    l1, c1 = map(int, out["CALL1"].split()[2:4])
    assert l0 == 4 and l1 == 8, out                       # one launch per GPU per call: two frames each (grid z)
    assert c1 == c0, out                                    # the second call uploads nothing: tables and frame tables are cached per GPU
    assert out["LIVE"].split()[1] == "0", out
    assert all(f"gpu {g}: " in out["CHECK"] for g in range(4)), out
    launches = [ln for ln in log.splitlines() if ln.startswith("launch ")]
    assert sorted(int(ln.split("dev=")[1].split()[0]) for ln in launches) == [0, 0, 1, 1, 2, 2, 3, 3], launches
    assert all(ln.split("dev=")[1].split()[0] == ln.split("stream=")[1].split()[0] for ln in launches), launches      # ... each on a stream of its own GPU
    assert all("grid=225,1,2 " in ln for ln in launches), launches
    # every GPU's table copy: the same two blocks, uploaded on that GPU before its first launch
    first_launch = next(i for i, ln in enumerate(log.splitlines()) if ln.startswith("launch "))
    ups = [ln for ln in log.splitlines()[:first_launch] if ln.startswith("copy ") and "kind=1" in ln and "bytes=192" not in ln]
    per = {}
    for ln in ups:
        per.setdefault(int(ln.split("dev=")[1].split()[0]), []).append(int(ln.split("bytes=")[1]))
    assert sorted(per) == [0, 1, 2, 3] and len({tuple(v) for v in per.values()}) == 1, per


def test_single_gpu_box_takes_the_same_batch(stub, tmp_path):
    """HIPSTUB_DEVICES=1: the same eight pairs, all on GPU 0 -- one launch of eight frames"""
    code = PEERS.replace("owners = [0, 1, 2, 3, 3, 2, 1, 0]", "owners = [0] * 8")
    r, log = _run(stub, code, tmp_path, ndev=1)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    launches = [ln for ln in log.splitlines() if ln.startswith("launch ")]
    assert len(launches) == 2 and all("dev=0 " in ln and ",1,8 block=256,1,1" in ln for ln in launches), launches      # (fewer row segments per frame than the two-frame launches above: eight frames fill the chip)


def test_no_gpu_fails_loudly(stub, tmp_path):
    """HIPSTUB_DEVICES=0 is a box without a GPU: the conversion call fails (there is no CPU path to fall back to), the context itself can still be made and planned"""
    code = r"""
from librempeg_amd import SwsContext, SWS_BICUBIC
from librempeg_amd.swscale import HostFrame
c = SwsContext(64, 36, "yuv420p", 128, 72, "rgb24", SWS_BICUBIC)
r = c.scale(HostFrame("yuv420p", 64, 36), HostFrame("rgb24", 128, 72))
print("RET", r)
assert r < 0
"""
    r, log = _run(stub, code, tmp_path, ndev=0)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "launch " not in log


def test_the_stub_catches_a_copy_that_leaves_its_block(stub, tmp_path):
    """the test double's own known-answer test: one byte past a device block aborts the process with a message"""
    code = r"""
import ctypes as C
L = C.CDLL(None)
L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
p = C.c_void_p()
assert L.hipMalloc(C.byref(p), 1000) == 0
src = (C.c_uint8 * 2000)()
assert L.hipMemcpy(p, src, 1000, 1) == 0
print("IN-BOUNDS OK", flush=True)
L.hipMemcpy(p, src, 1001, 1)
print("NOT REACHED")
"""
    r, _ = _run(stub, code, tmp_path)
    assert r.returncode != 0 and "IN-BOUNDS OK" in r.stdout and "NOT REACHED" not in r.stdout
    assert "hipstub: copy writes outside a device allocation" in r.stderr, r.stderr[-500:]


def test_the_stub_catches_a_kernel_argument_into_a_freed_block(stub, tmp_path):
    """... and the launch-time argument check's: a conversion handed a frame that was freed (its bytes stay mapped in the stub's quarantine, as they may on a GPU) is
    stopped at the launch, naming the argument -- the check every launch of the hunt passes through, for its tables as for its frames"""
    code = r"""
import sys
sys.argv = ["x", "1", "1"]
import hipstub_hunt as H
from librempeg_amd import SwsContext, SWS_BICUBIC
c = SwsContext(640, 360, "yuv420p", 1280, 720, "rgb24", SWS_BICUBIC)
s, d = H.StubFrame("yuv420p", 640, 360, 0), H.StubFrame("rgb24", 1280, 720, 0)
assert c.scale(s, d) == 720
print("LIVE FRAMES OK", H.STUB.hipstub_checked_pointers(), flush=True)
d.free()
d.base = None
d2 = H.StubFrame("rgb24", 1280, 720, 0)
import ctypes as C
dp, ds = d2.ptrs()
H.STUB.hipFree(d2.base)
sp, ss = s.ptrs()
c.L.sws_scale(c.c, sp, ss, 0, 360, dp, ds)
print("NOT REACHED")
"""
    r, _ = _run(stub, code, tmp_path)
    assert r.returncode != 0 and "LIVE FRAMES OK" in r.stdout and "NOT REACHED" not in r.stdout, r.stdout + r.stderr[-800:]
    assert int(r.stdout.split("LIVE FRAMES OK")[1].split()[0]) >= 10
    assert "points into a device block that was FREED" in r.stderr and "destination plane of the frame table" in r.stderr, r.stderr[-800:]


def test_the_stub_catches_a_pinned_source_rewritten_before_its_copy_ran(stub, tmp_path):
    """HIPSTUB_DEFER=1's own known-answer test"""
    code = r"""
import ctypes as C
L = C.CDLL(None)
for f in (L.hipMalloc, L.hipHostMalloc):
    f.argtypes = [C.POINTER(C.c_void_p), C.c_size_t] + ([C.c_uint] if f is L.hipHostMalloc else [])
L.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
L.hipStreamSynchronize.argtypes = [C.c_void_p]
d, h, st = C.c_void_p(), C.c_void_p(), C.c_void_p()
assert L.hipMalloc(C.byref(d), 64) == 0 and L.hipHostMalloc(C.byref(h), 64, 0) == 0 and L.hipStreamCreateWithFlags(C.byref(st), 1) == 0
C.memset(h, 1, 64)
assert L.hipMemcpyAsync(d, h, 64, 1, st) == 0 and L.hipStreamSynchronize(st) == 0
print("QUEUED, RUN, UNCHANGED: OK", flush=True)
assert L.hipMemcpyAsync(d, h, 64, 1, st) == 0
C.memset(h, 2, 1)
L.hipStreamSynchronize(st)
print("NOT REACHED")
"""
    r, _ = _run(stub, code, tmp_path, defer=1)
    assert r.returncode != 0 and "UNCHANGED: OK" in r.stdout and "NOT REACHED" not in r.stdout, r.stdout + r.stderr[-500:]
    assert "PINNED host memory runs after the host rewrote its source" in r.stderr, r.stderr[-500:]


RCCL = PEERS.replace('c = SwsContext(640, 360, "yuv420p", 1280, 720, "rgb24", SWS_BICUBIC | SWS_BITEXACT)',
                     'c = SwsContext(640, 360, "yuv420p", 1280, 720, "rgb24", SWS_BICUBIC | SWS_BITEXACT)\nassert c.set_option("rccl_tables", 1) == 0') + r"""
import ctypes as C
R = C.CDLL(None)
R.rcclstub_broadcasts.restype = R.rcclstub_inits.restype = C.c_ulong
print("RCCL", R.rcclstub_broadcasts(), R.rcclstub_inits())
"""


@pytest.mark.parametrize("defer", [0, 1])
def test_rccl_table_delivery(stub, tmp_path, defer):
    """option rccl_tables over the RCCL test double (csrc/dev_rccl.hip): the home GPU's two table blocks are uploaded once, each goes to the three
    peers with ONE grouped ncclBroadcast over ONE communicator, the peers' copies read back equal to the upload (sws_hip_debug_check compares hashes), the second call
    sends nothing.  Immediate and laziest-GPU execution."""
    r, log = _run(stub, RCCL, tmp_path, defer=defer, rccl="")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "RCCL 2 1" in r.stdout and "LIVE 0" in r.stdout, r.stdout
    assert all(f"gpu {g}: 2 table blocks checked" in r.stdout for g in range(4)), r.stdout
    copies = [ln for ln in log.splitlines() if ln.startswith("copy ") and "bytes=192" not in ln]
    h2d = sorted(int(ln.split("dev=")[1].split()[0]) for ln in copies if "kind=1" in ln)
    d2d = sorted(int(ln.split("dev=")[1].split()[0]) for ln in copies if "kind=3" in ln)
    assert h2d == [0, 0] and d2d == [1, 1, 2, 2, 3, 3], (h2d, d2d)


@pytest.mark.parametrize("what", ["init", "bcast"])
def test_rccl_failure_falls_back_to_host_copies(stub, tmp_path, what):
    """a communicator that cannot be made, a broadcast that fails: not an error of the call -- every GPU gets its tables by the host -> device copy, and they are right"""
    r, log = _run(stub, RCCL, tmp_path, rccl=what)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "tables go by host -> device copies" in r.stderr, r.stderr[-800:]
    assert all(f"gpu {g}: 2 table blocks checked" in r.stdout for g in range(4)) and "LIVE 0" in r.stdout, r.stdout
    copies = [ln for ln in log.splitlines() if ln.startswith("copy ") and "bytes=192" not in ln]
    assert sorted(int(ln.split("dev=")[1].split()[0]) for ln in copies if "kind=1" in ln) == [0, 0, 1, 1, 2, 2, 3, 3] and not [ln for ln in copies if "kind=3" in ln], copies


def test_a_damaged_table_block_fails_the_first_conversion_loudly(stub, tmp_path):
    """the always-on integrity check (csrc/dev_state.hip verify_tables_once): a table block that no longer holds its upload when the plan is first launched fails THAT call
    with a message naming the block; the context re-plans and converts on the next call; an undamaged context reads its tables back exactly once"""
    code = r"""
import ctypes as C, sys
sys.argv = ["x", "1", "1"]
import hipstub_hunt as H
from librempeg_amd import SwsContext, SWS_BICUBIC, SWS_BITEXACT
c = SwsContext(640, 360, "yuv420p", 1280, 720, "rgb24", SWS_BICUBIC | SWS_BITEXACT)
c.L.sws_hip_plan.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
assert c.L.sws_hip_plan(c.c, (C.c_uint64 * 3)()) == 0          # planned: the tables are on the "GPU", nothing launched yet
H.STUB.hipstub_scribble.argtypes = [C.c_ulonglong]
assert H.STUB.hipstub_scribble(130688) == 1                     # (the larger of this conversion's two table blocks)
s, d = H.StubFrame("yuv420p", 640, 360, 0), H.StubFrame("rgb24", 1280, 720, 0)
r1 = c.scale(s, d)
print("FIRST", r1, H.STUB.hipstub_launches(), flush=True)
r2 = c.scale(s, d)
print("SECOND", r2, H.STUB.hipstub_launches(), flush=True)
r3 = c.scale(s, d)
print("THIRD", r3, H.STUB.hipstub_launches(), flush=True)
"""
    r, log = _run(stub, code, tmp_path)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.split() and ln.split()[0] in ("FIRST", "SECOND", "THIRD")}
    assert int(out["FIRST"][0]) < 0 and out["FIRST"][1] == "0", out           # refused, nothing launched
    assert "no longer holds what was uploaded" in r.stderr, r.stderr[-600:]
    assert out["SECOND"] == ["720", "1"] and out["THIRD"] == ["720", "2"], out  # re-planned, converts
    # read-backs (device -> host copies of the table sizes): the failed check stopped at the damaged block or before it, the re-plan read both back once; the third call none
    d2h = [ln for ln in log.splitlines() if ln.startswith("copy ") and "kind=2" in ln and ("bytes=130688" in ln or "bytes=41984" in ln)]
    assert 3 <= len(d2h) <= 4, d2h
