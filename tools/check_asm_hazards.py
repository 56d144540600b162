#!/usr/bin/env python3
"""Scan the gfx950 ISA of csrc/txlayer.hip for the one hazard hipcc cannot pad around an inline-asm MFMA: a VGPR written
by a VALU instruction and read as an operand of an asm v_mfma within the next two issue states (cdna_hip_programming.md
§5.7 item 2).  Exit code 1 and a listing if any is found.  usage: check_asm_hazards.py [file.s]  (default: compiles
dorado_amd/csrc/txlayer.hip with hipcc -S)."""
import os
import re
import subprocess
import sys
import tempfile


def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def scan(path):
    lines = open(path).read().split('\n')
    bad = []
    in_asm = False
    recent = []   # (instruction text, written vgprs) of the last compiler instructions
    kernel = "?"
    for n, l in enumerate(lines):
        t = l.strip()
        m = re.match(r'^(_Z\w+):', l)
        if m:
            kernel = m.group(1)
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
            continue
        op = t.split()[0]
        args = [a.strip() for a in t[len(op):].split(',')]
        if in_asm and op.startswith('v_mfma'):
            src = set()
            for a in args[1:]:
                src |= regs(a)
            for k, (txt, wr) in enumerate(reversed(recent[-2:])):
                if wr & src:
                    bad.append((n + 1, kernel + ": " + t, txt))
            recent = []
            continue
        if in_asm:
            continue
        wr = set()
        if op.startswith('v_') and not op.startswith('v_mfma') and not op.startswith('v_cmp'):
            wr = regs(args[0]) if args else set()
            if op.startswith('v_permlane') and 'swap' in op and len(args) > 1:
                wr |= regs(args[1])       # the swaps write both operands
        if op == 's_nop':
            k = int(args[0]) + 1 if args and args[0].isdigit() else 1
            recent += [("s_nop", set())] * k
        else:
            recent.append((t, wr))
        recent = recent[-4:]
    return bad


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        path = os.path.join(tempfile.mkdtemp(), "txlayer.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                               "-w", os.path.join(root, "dorado_amd", "csrc", "txlayer.hip"), "-o", path])
    bad = scan(path)
    for n, mf, prod in bad:
        print(f"line {n}: {mf}\n    operand written by: {prod}")
    print(f"{len(bad)} VALU -> asm-MFMA operand hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
