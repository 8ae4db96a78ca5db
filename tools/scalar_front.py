#!/usr/bin/env python3
"""How many DEPENDENT scalar-load round trips does a kernel make before its first vector-memory request?

Round 5: the fused optimiser launch read its argument records field by field where they were used — a dozen s_load / s_waitcnt
pairs, 2.2 us, in front of its first operand load.  This lint compiles the translation units to gfx950 assembly and, per kernel,
counts the `s_waitcnt lgkmcnt(0)` that follow at least one s_load before the first global / buffer load (listing order, not control
flow: a guide, not a proof).

usage: tools/scalar_front.py [unit.hip ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "goal-conditioned-rl-framework_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S"]
UNITS = sys.argv[1:] or ["rowchain.hip", "gemm_mfma.hip", "ops.hip", "ops_sac.hip", "bn_slab.hip", "dw_adam.hip", "her_ring.hip", "rowtile.hip", "xchg_ipc.hip", "normalizer.hip"]


def kernels(text):
    name, body, is_kernel = None, [], set(re.findall(r"\.amdhsa_kernel\s+(\S+)", text))
    for line in text.splitlines():
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            if name in is_kernel:
                yield name, body
            name, body = m.group(1), []
        elif name:
            body.append(line)
    if name in is_kernel:
        yield name, body


with tempfile.TemporaryDirectory() as tmp:
    for unit in UNITS:
        asm = os.path.join(tmp, unit + ".s")
        subprocess.run([HIPCC] + FLAGS + ["-x", "hip", os.path.join(CSRC, unit), "-o", asm], check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
        for name, body in kernels(open(asm).read()):
            trips, pending, loads = 0, False, 0
            for line in body:
                if re.match(r"\s*s_load", line):
                    pending, loads = True, loads + 1
                elif re.match(r"\s*s_waitcnt.*lgkmcnt\(0\)", line) and pending:
                    trips, pending = trips + 1, False
                elif re.match(r"\s*(global_load|buffer_load|flat_load|global_atomic)", line):
                    break
            full = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            short = re.sub(r"\(anonymous namespace\)::|gcrl::|void ", "", full).split("(")[0][:70]
            print(f"{unit:16s} {trips:3d} scalar round trips ({loads:3d} s_loads) before the first vector load   {short}")
