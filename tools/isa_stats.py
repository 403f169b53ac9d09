#!/usr/bin/env python3
"""ISA-level record of the kernels: per kernel VGPRs, SGPRs, LDS, scratch bytes, spills, waves per SIMD, code size -- and, for the
tile kernel, where its scratch accesses sit relative to its loops.  Compiles vulkan_forge_amd/csrc/vf_hip.hip once more with the
flags of __graft_entry__.build() plus --save-temps (into build/asm/) and reads the assembler listing's per-kernel notes.

usage: tools/isa_stats.py [output file]        (default: stdout)"""
import hashlib, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g


def collect():
    """(text report, {demangled kernel name: notes dict}, {tile kernel instantiation: {(loop depth, load|store): count}})"""
    asm_dir = os.path.join(ROOT, "build", "asm")
    os.makedirs(asm_dir, exist_ok=True)
    subprocess.check_call([g._hipcc(), *g.HIPCC_FLAGS, "--save-temps", g.HIP_SRC, "-o", os.path.join(asm_dir, "libvf_isa.so")], cwd=asm_dir,
                          stderr=subprocess.DEVNULL)
    s = open(os.path.join(asm_dir, "vf_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    out = []
    lib = os.path.join(ROOT, "vulkan_forge_amd", "libvf_hip.so")
    out.append(f"lib_sha256 {hashlib.sha256(open(lib, 'rb').read()).hexdigest()}  (vulkan_forge_amd/libvf_hip.so; this listing: the same sources and flags + --save-temps)")
    out.append(f"flags: {' '.join(g.HIPCC_FLAGS)}")
    out.append("")
    out.append(f"{'kernel':34s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'LDS B':>7s} {'scratch B':>9s} {'waves/SIMD':>10s} {'code B':>7s} {'scratch ld/st':>13s}")
    def demangle(n):
        """vf::k_tile<false, false, true> from the mangled name, without a demangler: the template arguments are the Lb0E / Lb1E runs"""
        m = re.match(r"_ZN2vf(\d+)", n)
        if not m:
            return n
        ln = int(m.group(1)); base = n[len(m.group(0)):len(m.group(0)) + ln]; rest = n[len(m.group(0)) + ln:]
        t = re.match(r"I((?:Lb[01]E)+)E", rest)
        if t:
            return base + "<" + ", ".join("true" if b == "1" else "false" for b in re.findall(r"Lb([01])E", t.group(1))) + ">"
        t = re.match(r"I([fd])E", rest)
        return base + ("<float>" if t and t.group(1) == "f" else "<double>" if t else "")
    # one linear pass: a kernel's text runs from its label to .Lfunc_end; its notes ("; NumVgprs: ..") follow under "; Kernel info:"
    lines = s.split("\n")
    bodies, notes, cur, body_of, in_kernel_info = {}, {}, None, None, False
    for ln in lines:
        if ln.startswith("_ZN2vf") and ":" in ln.split(";")[0]:
            body_of = ln.split(":")[0]; bodies[body_of] = []
        elif ln.startswith(".Lfunc_end"):
            body_of = None
        elif body_of is not None:
            bodies[body_of].append(ln)
        m = re.match(r"\s*\.amdhsa_kernel (\S+)", ln)
        if m:
            cur = m.group(1); notes[cur] = {}
        if ln.startswith("; Kernel info:"): in_kernel_info = True
        elif ln.startswith("; Function info:") or ln.startswith("\t.text") or ln.startswith("\t.section"): in_kernel_info = False
        m = re.match(r"; (codeLenInByte|TotalNumSgprs|NumVgprs|NumAgprs|ScratchSize|LDSByteSize|Occupancy)\s*[:=]\s*(\d+)", ln)
        if m and cur and in_kernel_info:
            notes[cur][m.group(1)] = m.group(2)
    records = {}
    for name, nt in notes.items():
        body = "\n".join(bodies.get(name, []))
        records[demangle(name)] = dict(nt)
        ld, st = len(re.findall(r"\bscratch_load", body)), len(re.findall(r"\bscratch_store", body))
        out.append(f"{demangle(name):34s} {nt.get('NumVgprs','?'):>5s} {nt.get('NumAgprs','?'):>5s} {nt.get('TotalNumSgprs','?'):>5s} {nt.get('LDSByteSize','?'):>7s} {nt.get('ScratchSize','?'):>9s} {nt.get('Occupancy','?'):>10s} {nt.get('codeLenInByte','?'):>7s} {ld:>6d}/{st:<6d}")
    bodies = {k: "\n".join(v) for k, v in bodies.items()}
    depths = {}
    out.append("")
    out.append("Tile kernel, frame's main launch (k_tile<false, false, true>): where the scratch accesses sit (loop depth of the enclosing basic block;")
    out.append("depth 1 = the item loop, 2 = chunks / fragment rows, 3 = the block pull loop, 4 = pass A / pass B rounds, 5 = the line loop, 6 = painting):")
    for name, body in bodies.items():
        if "k_tileILb0ELb0ELb1" not in name:
            continue
        depth = 0
        hist = {}
        depths[demangle(name)] = hist
        for line in body.split("\n"):
            mm = re.search(r"Loop Header: Depth=(\d+)|Loop: Header=\S+ Depth=(\d+)", line)
            if line.startswith(".LBB") or line.startswith("; %bb."):
                depth = 0
            if mm:
                depth = int(mm.group(1) or mm.group(2))
            if "scratch_" in line:
                op = "load" if "scratch_load" in line else "store"
                hist[(depth, op)] = hist.get((depth, op), 0) + 1
        for (d, op), n in sorted(hist.items()):
            out.append(f"   loop depth {d}: {n} scratch {op}s")
        if not hist:
            out.append("   none")
    return "\n".join(out) + "\n", records, depths


if __name__ == "__main__":
    text, _records, _depths = collect()
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)
    print(text)
