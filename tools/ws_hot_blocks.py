#!/usr/bin/env python
"""The hot basic blocks of one instantiation of the wave-split entropy kernel in an object file: instructions, float64
FMAs, vector instructions, v_readlane / v_writelane (reloads of spilled scalar registers), waits.  The kernel's registers
are allotted over ALL of its code, so a change in a rider path can put spill reloads into the batch loop: this is the
check that needs no GPU.
    python tools/ws_hot_blocks.py pyvbmc_amd/csrc/_obj/entropy_ws_dp10.o [mangled-name fragment]"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

B = "/opt/rocm/lib/llvm/bin/"
obj = sys.argv[1]
frag = sys.argv[2] if len(sys.argv) > 2 else "entmc_ws_kernelILi10ELi13ELb1ELb0ELi1E"
with tempfile.TemporaryDirectory() as d:
    d = Path(d)
    subprocess.check_call([B + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, str(d / "f.bin")])
    tg = [t for t in subprocess.check_output([B + "clang-offload-bundler", "--list", "--type=o", f"--input={d / 'f.bin'}"]).decode().split() if "gfx950" in t][0]
    subprocess.check_call([B + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={d / 'f.bin'}", f"--targets={tg}", f"--output={d / 'k.co'}"])
    dis = subprocess.run([B + "llvm-objdump", "-d", "--symbolize-operands", str(d / "k.co")], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().splitlines()
blocks, cur, name, on = [], [], "entry", False
for l in dis:
    m = re.match(r"^[0-9a-f]+ <(.*)>:\s*$", l)
    if m:
        lab = m.group(1)
        if re.match(r"^L\d+$", lab):
            if on:
                blocks.append((name, cur)); cur = []; name = lab
        else:
            if on:
                break
            on = frag in lab
        continue
    if on and l.strip():
        cur.append(l)
blocks.append((name, cur))
cnt = lambda b, *keys: sum(any(k in x for k in keys) for x in b)
print(f"{obj}: {frag}: {len(blocks)} blocks, {sum(len(b) for _, b in blocks)} instructions")
for n, b in sorted(blocks, key=lambda nb: -cnt(nb[1], "v_fma", "v_fmac"))[:3]:
    print(f"   {n}: {len(b)} instructions, {cnt(b, 'v_fma', 'v_fmac')} FMAs, {sum(x.strip().startswith('v_') for x in b)} vector, "
          f"{cnt(b, 'v_readlane', 'v_writelane')} lane moves, {cnt(b, 'v_mov_b')} v_mov, {cnt(b, 's_waitcnt')} waits, {cnt(b, 's_load')} s_load")
