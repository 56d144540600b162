#!/usr/bin/env python3
"""Scan gfx950 ISA for the hazards hipcc cannot pad around inline asm (its hazard recogniser does not look inside an asm
string).  Exit code 1 and a listing if any is found.

  check_asm_hazards.py [file.s]            csrc/txlayer.hip (default: compiled here with hipcc -S): a VGPR written by a VALU
                                           instruction and read as an operand of an asm v_mfma within the next two issue
                                           states (cdna_hip_programming.md 5.7 item 2)
  check_asm_hazards.py --valu [file.s]     csrc/decode.hip: the VALU -> VALU rules around its asm reductions / match bits:
      * a VGPR written by a VALU instruction and read through DPP                      2 wait states
      * an SGPR pair / VCC written by a VALU instruction (v_cmp, a carry-out) and read
        as carry-in or select mask (v_addc / v_subb / v_cndmask)                       2 wait states
      * a VGPR written by a VALU instruction and read by v_permlane*_swap              2 wait states
      * a VGPR written by a VALU instruction and read by v_readlane / v_readfirstlane  1 wait state
    checked wherever producer or consumer sits inside an asm region (compiler-to-compiler pairs are hipcc's business)."""
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


def sregs(tok):
    tok = tok.strip()
    if tok == 'vcc':
        return {'vcc'}
    m = re.match(r's\[(\d+):(\d+)\]$', tok)
    if m:
        return {'s%d' % i for i in range(int(m.group(1)), int(m.group(2)) + 1)}
    m = re.match(r's(\d+)$', tok)
    return {'s' + m.group(1)} if m else set()


class _Any(set):
    """The unknown write set of a basic-block entry: intersects every register set."""

    def __and__(self, other):
        return bool(other)


def scan_valu(path):
    """-> [(line, kernel, consumer text, producer text, rule)]
    A label starts a new basic block: the instructions in front of it textually are not the only predecessors, so the
    wait-state window is replaced by an 'unknown producer' entry that covers every register — an asm consumer within its hazard
    distance of a block entry is reported as unverifiable unless s_nop / independent instructions pad it."""
    lines = open(path).read().split('\n')
    bad = []
    in_asm = False
    recent = []   # (text, vgprs written, sgprs written, inside asm) per wait state, newest last
    kernel = "?"
    for n, l in enumerate(lines):
        t = l.strip()
        m = re.match(r'^(_Z\w+):', l)
        if m:
            kernel = m.group(1)
            recent = []
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if t.endswith(':') and not t.startswith(';') and not m:
            recent = [("<basic-block entry " + t + " (producer on another edge unknown)>", _Any(), _Any(), True)]
            continue
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
            continue
        t = t.split(';')[0].strip()
        op = t.split()[0]
        rest = t[len(op):]
        args = [a.strip().split(' ')[0] for a in rest.split(',')]
        if op == 's_nop':
            k = int(args[0]) + 1 if args and args[0].isdigit() else 1
            recent += [("s_nop", set(), set(), in_asm)] * k
            recent = recent[-4:]
            continue
        is_valu = op.startswith('v_') and not op.startswith('v_mfma')
        # ---- consumer side ----
        checks = []   # (registers read, kind 'v' / 's', wait states needed, rule)
        if is_valu:
            if '_dpp' in op or ' quad_perm:' in rest or ' row_' in rest:
                src = set()
                for a in args[1:]:
                    src |= regs(a)
                checks.append((src, 'v', 2, "VALU write -> DPP read"))
            if op.startswith('v_permlane') and 'swap' in op:
                src = set()
                for a in args[:2]:
                    src |= regs(a)
                checks.append((src, 'v', 2, "VALU write -> v_permlane swap read"))
            if op.startswith('v_readlane') or op.startswith('v_readfirstlane'):
                checks.append((regs(args[1]) if len(args) > 1 else set(), 'v', 1, "VALU write -> v_readlane read"))
            if op.startswith(('v_addc_co', 'v_subb_co', 'v_subbrev_co', 'v_cndmask')):
                msk = {'vcc'} if op.endswith('_e32') else (sregs(args[-1]) if args else set())
                checks.append((msk, 's', 2, "VALU write of an SGPR mask -> carry-in / select read"))
        for src, kind, need, rule in checks:
            for dist, (txt, vw, sw, p_asm) in enumerate(reversed(recent[-need:])):
                wr = vw if kind == 'v' else sw
                if (wr & src) and (in_asm or p_asm):
                    bad.append((n + 1, kernel, t, txt, rule))
        # ---- producer side ----
        vw, sw = set(), set()
        if is_valu:
            if op.startswith('v_cmp'):
                sw = {'vcc'} if op.endswith('_e32') else (sregs(args[0]) if args else set())
            elif op.startswith(('v_readlane', 'v_readfirstlane')):
                pass                         # writes an SGPR through the scalar path, not a mask hazard source here
            else:
                vw = regs(args[0]) if args else set()
                if op.startswith(('v_addc_co', 'v_subb_co', 'v_subbrev_co', 'v_add_co', 'v_sub_co')):
                    sw = {'vcc'} if op.endswith('_e32') else (sregs(args[1]) if len(args) > 1 else set())
                if op.startswith('v_permlane') and 'swap' in op and len(args) > 1:
                    vw |= regs(args[1])
        recent.append((t, vw, sw, in_asm))
        recent = recent[-4:]
    return bad


def compile_s(src, extra=()):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(tempfile.mkdtemp(), os.path.basename(src).replace(".hip", ".s"))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                           "-w", *extra, os.path.join(root, "dorado_amd", "csrc", src), "-o", path])
    return path


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--valu":
        path = sys.argv[2] if len(sys.argv) > 2 else compile_s("decode.hip", ("-ffp-contract=off",))
        bad = scan_valu(path)
        for n, kern, cons, prod, rule in bad:
            print(f"line {n}: {kern}: {cons}\n    {rule}; written by: {prod}")
        print(f"{len(bad)} VALU -> VALU hazards around inline asm")
        return 1 if bad else 0
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
