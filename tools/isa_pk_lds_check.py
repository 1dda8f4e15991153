#!/usr/bin/env python3
"""Static guard for the gfx950 hazard found in round 4 (profiles/r04_gn_prologue_rootcause.txt): a packed-fp32 VALU instruction
(v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) that executes while LDS read returns of the same wave are still outstanding
(issued under a partial `s_waitcnt lgkmcnt(N)`, N > 0) returned 0.0 in its low half for lanes 48-63 in kernels that run beside
LDS-DMA traffic.  This script disassembles the device code of every object of the library and reports each packed-fp32
instruction that can execute with DS reads in flight (control flow followed to a fixed point).  Two classes:
  * SIGNATURE sites -- the form that failed on the MI355X: the LOW half of the packed result takes a HIGH source dword
    (`op_sel:[..1..]`; v_pk_mul_f32 v[52:53], v[30:31], v[64:65] op_sel:[0,1] under `s_waitcnt vmcnt(1) lgkmcnt(2)`).
    The library is kept free of them (exit status 1 otherwise): the producing code waits for lgkmcnt(0) first.
  * all other packed-fp32 instructions under outstanding LDS reads (~1100 in the r4 build: the chain kernels' epilogues) have
    never failed in any determinism probe; they are counted and listed with --all, not rejected.

    python tools/isa_pk_lds_check.py [--all] [ns2vc_amd/lib/obj/*.o]

Used by tests/test_cpu.py::test_no_packed_fp32_under_outstanding_lds_reads (runs without a GPU)."""
from __future__ import annotations

import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
PK = re.compile(r"^v_pk_(mul|fma|add)_f32\b")
DS_RET = re.compile(r"^ds_(read|load|bpermute|permute|swizzle|consume|append|ordered_count|[a-z0-9_]*_rtn)")   # DS ops that return data to VGPRs
WAIT = re.compile(r"^s_waitcnt\b(.*)")
LGKM = re.compile(r"lgkmcnt\((\d+)\)")
FUNC = re.compile(r"^([0-9a-f]+) <([^>]+)>:")
INSN = re.compile(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):")
BR = re.compile(r"^(s_cbranch_\w+|s_branch)\s+\S+\s*$|^(s_cbranch_\w+|s_branch)\b")
TARGET = re.compile(r"<[^>+]+(?:\+0x([0-9a-fA-F]+))?>\s*$")
CAP = 15


def device_disassembly(obj: str) -> str:
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return ""                                   # host-only object
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{obj}: cannot unbundle the gfx950 code object: {r.stderr.strip()}")
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout


def functions(dis: str):
    """yield (name, base address, [(address, text, raw line)])"""
    name, base, insns = None, 0, []
    for line in dis.splitlines():
        m = FUNC.match(line)
        if m:
            if name and insns:
                yield name, base, insns
            name, base, insns = m.group(2), int(m.group(1), 16), []
            continue
        m = INSN.match(line)
        if m and name:
            insns.append((int(m.group(2), 16), m.group(1).strip(), line))
    if name and insns:
        yield name, base, insns


def scan(name: str, base: int, insns):
    """forward data flow of `max DS read returns possibly outstanding` over the control-flow graph; returns the flagged sites"""
    index = {a: i for i, (a, _, _) in enumerate(insns)}
    n = len(insns)
    succ = [[] for _ in range(n)]
    for i, (a, t, raw) in enumerate(insns):
        op = t.split()[0]
        tgt = None
        if op.startswith("s_cbranch") or op == "s_branch":
            m = TARGET.search(raw.split("//")[1]) if "//" in raw else None
            if m:
                tgt = base + int(m.group(1) or "0", 16)
        if op in ("s_endpgm", "s_setpc_b64"):
            continue
        if op != "s_branch" and i + 1 < n:
            succ[i].append(i + 1)
        if tgt is not None and tgt in index:
            succ[i].append(index[tgt])
    state_in = [-1] * n
    state_in[0] = 0
    work = [0]
    while work:
        i = work.pop()
        s = state_in[i]
        t = insns[i][1]
        op = t.split()[0]
        if DS_RET.match(op):
            s = min(CAP, s + 1)
        else:
            m = WAIT.match(t)
            if m:
                g = LGKM.search(m.group(1))
                if g:
                    s = min(s, int(g.group(1)))
                elif re.match(r"^\s*(0x[0-9a-f]+|\d+)\s*$", m.group(1)):        # raw immediate form
                    v = int(m.group(1).strip(), 0)
                    s = min(s, (v >> 8) & 0xF)
        for j in succ[i]:
            if s > state_in[j]:
                state_in[j] = s
                work.append(j)
    return [(hex(insns[i][0]), insns[i][1], state_in[i]) for i in range(n) if state_in[i] > 0 and PK.match(insns[i][1])]


def is_signature(text: str) -> bool:
    m = re.search(r"\bop_sel:\[([01,]+)\]", text)
    return bool(m and "1" in m.group(1))


def check(objs):
    sites = []
    for obj in objs:
        for name, base, insns in functions(device_disassembly(obj)):
            for addr, text, outstanding in scan(name, base, insns):
                sites.append((os.path.basename(obj), name, addr, text, outstanding))
    return sites


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    show_all = "--all" in sys.argv[1:]
    args = [a for a in sys.argv[1:] if a != "--all"]
    objs = args or sorted(glob.glob(os.path.join(here, "ns2vc_amd", "lib", "obj", "*.o")))
    found = check(objs)
    sig = [x for x in found if is_signature(x[3])]
    for o, f, a, t, k in (found if show_all else sig):
        print(f"{o}: {f[:90]} {a}: {t}   [<= {k} DS read(s) may be outstanding]")
    print(f"{len(found)} packed-fp32 instruction(s) that can execute with LDS read returns outstanding in {len(objs)} object(s); "
          f"{len(sig)} of them of the failing form (low half from a high source dword)")
    sys.exit(1 if sig else 0)
