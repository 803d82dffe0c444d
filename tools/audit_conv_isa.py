#!/usr/bin/env python3
"""Audit the conv kernels' ISA: every MFMA A operand must be a register written by a hand-issued weight load
(no compiler copy of an in-flight destination), and the kernels must not spill (cdna_hip_programming.md 5.7)."""
import re
import subprocess
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "qmri_pnp_recon_poc_amd/csrc/conv_kernels.hip"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-S", "--cuda-device-only",
                "-o", "/tmp/conv_audit.s", src], check=True)
txt = open("/tmp/conv_audit.s").read()
ok = True
for m in re.finditer(r"^(_ZN12_GLOBAL__N_16k_convILi(\d)ELi(\d)EEEvNS_8ConvArgsE):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
    kind, mt, body = m.group(2), m.group(3), m.group(4)
    lines = body.split("\n")
    ring = set()
    for l in lines:
        mm = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[", l)
        if mm:
            ring.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
    n = bad = 0
    for l in lines:
        mm = re.search(r"v_mfma_f32_32x32x2_f32 v\[\d+:\d+\], v(\d+), v(\d+)", l)
        if mm:
            n += 1
            bad += int(mm.group(1)) not in ring
    spill = sum("scratch_" in l for l in lines)
    print(f"kind {kind} MT {mt}: {n} MFMAs, {bad} with an A operand outside the ring registers, {len(ring)} ring registers, {spill} scratch ops")
    ok = ok and bad == 0 and spill == 0
sys.exit(0 if ok else 1)
