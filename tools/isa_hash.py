#!/usr/bin/env python3
"""Per-kernel hash of the gfx950 instructions inside a built library: does an edit of the sources change the device code?

    python tools/isa_hash.py [lib.so] > before.txt ; ...edit, rebuild... ; python tools/isa_hash.py | diff before.txt -

Addresses and encodings are dropped (a kernel that moves in the file keeps its hash), mnemonics and operands are kept; branch targets
are kept as offsets relative to the kernel's start."""
import hashlib, os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_lint


def hashes(lib):
    cos, tmp = isa_lint.device_code_objects(lib)
    out = {}
    for co in cos:
        dis = subprocess.run([f"{isa_lint.LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", co], check=True, capture_output=True, text=True).stdout
        dem = subprocess.run(["c++filt"], input=dis, capture_output=True, text=True).stdout
        cur, h, n = None, None, 0
        for ln in dem.split("\n"):
            m = re.match(r"^<(.*)>:$", ln.strip())
            if m:
                name = m.group(1)
                if not name.startswith("L") or "(" in name:      # a kernel (local labels look like LBB..)
                    if cur: out[cur] = (h.hexdigest()[:16], n)
                    cur, h, n = name, hashlib.sha256(), 0
                continue
            if cur and ln.strip():
                ins = ln.split("//")[0].strip()
                if ins:
                    h.update(ins.encode() + b"\n"); n += 1
        if cur: out[cur] = (h.hexdigest()[:16], n)
    if tmp:
        for f in os.listdir(tmp): os.unlink(os.path.join(tmp, f))
        os.rmdir(tmp)
    return out


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "vulkan_forge_amd", "libvf_hip.so")
    for k, (hh, n) in sorted(hashes(lib).items()):
        print(f"{hh} {n:6d} {k[:150]}")
