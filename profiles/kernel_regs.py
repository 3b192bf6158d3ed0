#!/usr/bin/env python3
"""profiles/kernel_regs.py [asm.s] -- register / scratch / LDS budget of every kernel in the gfx950 assembly (make -C vsearch_amd/csrc asm)."""
import re
import subprocess
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "build/asm/vsx_device-hip-amdgcn-amd-amdhsa-gfx950.s"
s = open(path).read()
names = [m.group(1) for m in re.finditer(r"\.amdhsa_kernel (\S+)", s)]
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
for (m, dn) in zip(re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S), dem):
    body = m.group(2)

    def g(k):
        r = re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body)
        return r.group(1) if r else "?"
    dn = re.sub(r"\(.*", "", dn.replace("void ", ""))
    v = int(g("next_free_vgpr"))
    waves = 8 if v <= 64 else 512 // ((v + 7) // 8 * 8)
    print(f"{dn:62s} vgpr={v:>4} (waves/SIMD {min(8, waves)}) sgpr={g('next_free_sgpr'):>4} scratch={g('private_segment_fixed_size'):>5} lds={g('group_segment_fixed_size')}")
