"""CPU (-m "not gpu"): build check of the inter-workgroup publications.  tools/check_release_isa.py compiles the
kernels that hand data to other workgroups through memory — the BatchNorm slab row groups and the row-chain roles (csrc/meet.h),
the split-dW ticket (csrc/gemm_tiled.h), the peer-to-peer gradient exchange (csrc/xchg_ipc.hip) — to gfx950 assembly (device
side only: no GPU needed) and requires an `s_waitcnt vmcnt(0)` between a kernel's last write-through store and every arrival
(atomic add / system-scope flag store) that follows it: on gfx950 a workgroup-scope release fence emits no such wait.  Second
lint of the same script: no kernel copies its argument struct (or spills) into per-thread scratch."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_publication_is_drained_before_its_arrival():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_release_isa.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "release check: PASS" in r.stdout and "scratch check: PASS" in r.stdout
    for kernel in ("bn_linear_fwd_slab_kernel", "bn_linear_bwd_slab_kernel", "rowchain_split_kernel", "gemm_tiled_kernel", "xchg_two_shot_kernel"):
        assert kernel in r.stdout, kernel
