"""The accumulation kernel waits for "all but the youngest 17" vector-memory operations after a bucket flush
(msm_accum.hip, accum_take_point): that is only correct while the flush path issues AT LEAST 17 such operations between the
gather of the point and the wait.  This test counts them in the compiler's output for gfx950.  CPU only (hipcc cross-compiles);
the assembly is cached under csrc/build/ keyed by the hash of the sources."""
import hashlib
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kzg_poly_commit_exploration_amd", "csrc")
FLAGS = ["-DKZG_LAZY_FP", "-DKZG_FIPS_SQR", "-O3", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-S"]


def kernel_asm():
    h = hashlib.sha256()
    for f in ("msm_accum.hip", "g1_30.hip.h", "field30.hip.h", "field30_inv.hip.h", "engine.h"):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    out = os.path.join(CSRC, "build", "msm_accum_%s.s" % h.hexdigest()[:16])
    if not os.path.exists(out):
        subprocess.run(["hipcc"] + FLAGS + [os.path.join(CSRC, "msm_accum.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
    asm = open(out).read()
    i = asm.index("_ZN3kzg19k_bucket_accumulateE")
    i = asm.index(":", i)
    return asm[i:asm.index(".end_amdhsa_kernel", i)].splitlines(), asm


def is_vmem(line):
    op = line.split()[0] if line.split() else ""
    return op.startswith(("global_load", "global_store", "global_atomic", "scratch_", "buffer_", "flat_"))


@pytest.mark.skipif(subprocess.run(["which", "hipcc"], capture_output=True).returncode != 0, reason="no hipcc")
def test_flush_path_issues_the_operations_the_partial_wait_counts():
    body, _ = kernel_asm()
    waits = [n for n, l in enumerate(body) if re.search(r"s_waitcnt\s+vmcnt\(17\)", l)]
    assert len(waits) == 1, "one partial wait, in accum_take_point"
    w = waits[0]
    # walk back to the flush: the asm block of sixteen 16-byte stores
    stores = [n for n in range(w) if "global_store_dwordx4" in body[n]]
    first = max(n for n in stores if not any(m for m in stores if m == n - 1))  # start of the last run of consecutive stores
    run = [n for n in stores if n >= first]
    assert len(run) == 16 and run == list(range(first, first + 16)), "sixteen consecutive stores"
    # between the stores and the wait: the walk over empty buckets (a loop that waits for everything itself) and exactly one
    # more load on the straight path; nothing else that touches vector memory
    tail = [l for l in body[first + 16:w] if l.startswith("\t") and is_vmem(l.strip())]
    straight = [l for l in tail if "global_load_dword " in l or "global_load_dword\t" in l]
    assert all("global_load_dword" in l for l in tail), tail
    assert 1 <= len(straight) <= 2, tail  # the load inside the walk loop and the one behind it
    loop_waits = [n for n in range(first + 16, w) if re.search(r"s_waitcnt\s+vmcnt\(0\)", body[n])]
    assert len(tail) == 1 or loop_waits, "a second load is only allowed inside the walk, which waits for everything"


def test_two_workgroups_per_cu_by_register_count():
    body, asm = kernel_asm()
    body_text = "\n".join(body)
    i = asm.index(".name:           _ZN3kzg19k_bucket_accumulateE")
    m = re.search(r"\.vgpr_count:\s+(\d+)", asm[i:i + 2000])
    v = int(m.group(1))
    assert 169 <= v <= 256, "three workgroups of this kernel must not fit a CU (see the comment at its top): %d VGPRs" % v
    assert "s_waitcnt vmcnt(17)" in body_text
