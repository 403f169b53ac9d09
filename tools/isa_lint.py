#!/usr/bin/env python3
"""Build-time check of the gfx950 code inside a built library for an instruction form that MI355X executes wrongly.

Finding (round 4, tools/probes/shift64_top.hip, profiles/r04_hw_shift64_probe.log): the 64-bit shifts v_lshlrev_b64,
v_lshrrev_b64 and v_ashrrev_i64 misread their 32-bit shift amount when it sits in the LAST register of the wave's VGPR
allocation (v15 of 16, v23 of 24, .. v127 of 128): about 2 % of the lanes get a result shifted by a wrong amount, varying from
run to run.  Every other operand position and every other 64-bit instruction probed is right.  The compiler does not know, so a
kernel whose register count is a multiple of the allocation granule (8) can be hit whenever register allocation happens to put a
shift amount there -- k_triangle was, under one set of code-generation flags, and drew ~8 % of its waves wrongly.

This script disassembles the library's device code and reports every such instruction.  __graft_entry__.build() runs it and
refuses the library if one is found; tests/test_build_lint.py runs it on the built library as well.

usage: tools/isa_lint.py [path/to/lib.so]        exit code 1 = hazardous instruction present"""
import os, re, subprocess, sys, tempfile

def _llvm_bin():
    """the LLVM tools of the ROCm installation whose hipcc builds the library (VF_LLVM_BIN overrides)"""
    import shutil
    cands = [os.environ.get("VF_LLVM_BIN")]
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if os.path.exists(hipcc):
        cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin"))
    cands += [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin"), "/opt/rocm/lib/llvm/bin"]
    for c in cands:
        if c and os.path.exists(os.path.join(c, "llvm-objdump")) and os.path.exists(os.path.join(c, "llvm-readelf")):
            return c
    raise RuntimeError("llvm-objdump / llvm-readelf not found (set VF_LLVM_BIN)")


LLVM = _llvm_bin()
GRANULE = 8          # VGPR allocation granule of gfx90a and later (unified register file, wave64)
HAZARD = re.compile(r"\b(v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64)\s+v\[\d+:\d+\],\s*v(\d+)\s*,")


def device_code_objects(lib):
    """the gfx950 code objects bundled in a HIP shared library (or the file itself if it already is one)"""
    head = open(lib, "rb").read(20)
    if head[:4] == b"\x7fELF" and head[18:20] == b"\xe0\x00":      # e_machine 224 = AMDGPU
        return [lib], None
    tmp = tempfile.mkdtemp(prefix="vf_lint_")
    link = os.path.join(tmp, "lib.so")
    os.symlink(os.path.abspath(lib), link)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", link], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f and "gfx950" in f), tmp


def lint_code_object(co):
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    alloc = {}
    for blk in notes.split(".agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk); vg = re.search(r"\.vgpr_count:\s+(\d+)", blk); ag = re.match(r"\s*(\d+)", blk)
        if name and vg:
            total = int(vg.group(1)) + (int(ag.group(1)) if ag else 0)   # gfx90a+: AGPRs follow the VGPRs in one allocation
            alloc[name.group(1)] = (int(vg.group(1)), -(-max(total, 1) // GRANULE) * GRANULE)
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    found, checked, cur = [], 0, None
    for ln in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            if m.group(1) in alloc: cur = m.group(1)     # (other symbols are labels inside the kernel above them)
            continue
        if cur is None:
            continue
        h = HAZARD.search(ln)
        if h:
            checked += 1
            if int(h.group(2)) == alloc[cur][1] - 1:
                found.append((cur, alloc[cur][0], alloc[cur][1], ln.split("//")[0].strip()))
    return alloc, checked, found


def lint(lib):
    cos, tmp = device_code_objects(lib)
    if not cos:
        raise RuntimeError(f"{lib}: no gfx950 code object found")
    kernels, checked, found = 0, 0, []
    for co in cos:
        a, c, f = lint_code_object(co)
        kernels += len(a); checked += c; found += f
    if tmp:
        for f in os.listdir(tmp): os.unlink(os.path.join(tmp, f))
        os.rmdir(tmp)
    return kernels, checked, found


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "vulkan_forge_amd", "libvf_hip.so")
    kernels, checked, found = lint(lib)
    print(f"{lib}: {kernels} kernels, {checked} 64-bit shifts by a register amount checked, {len(found)} with the amount in the last allocated VGPR")
    for k, vg, al, ins in found:
        print(f"  HAZARD {k}: .vgpr_count {vg} (allocation {al}): {ins}")
    sys.exit(1 if found else 0)
