#!/usr/bin/env python3
"""Static check of the hand-counted weight ring (camal_kernels.hpp: w_load / w_wait) in the device assembly.

The ring's loads are inline asm: hipcc believes their destination registers are written at the asm
statement, while the data lands later (before the matching hand-placed s_waitcnt).  If the compiler
copies, spills or overwrites such a register between the load and its first consumer, the result is a
silent wrong value or, worse, a late write into a register that meanwhile holds something else.
This script walks every kernel of `hipcc -S --cuda-device-only` output and reports any non-asm
instruction that touches a register with a ring load in flight, other than the MFMA that consumes it.
It also reports an asm load whose SGPR base is not produced by the s_mov_b64 in front of it
(VALU-writes-SGPR -> VMEM hazard).   usage: check_ring.py file.s   -> exit status 1 on findings
"""
import re
import sys


def regs(tok):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.add(int(a))
    return out


def _scan(lines, at_branch):
    """One pass over a kernel's lines [(line number, text)].  at_branch[label] = registers in flight at the branches
    known to reach `label` (grows during the pass; the caller feeds the result back in until it stops growing, so
    that a label reached by a LATER branch — a loop's back edge — starts with that branch's loads in flight too)."""
    findings = []
    pending, inasm, prev, falls = set(), False, "", True
    for n, t in lines:
        if t.startswith(".LBB") and ":" in t:
            label = t.split(":")[0]
            pending = (set(pending) if falls else set()) | at_branch.get(label, set())
            falls = True
            continue
        if t.startswith("s_cbranch") or t.startswith("s_branch"):
            target = t.split()[1]
            at_branch[target] = at_branch.get(target, set()) | pending
            if t.startswith("s_branch"):
                falls = False
            prev = t
            continue
        if t.startswith("s_endpgm") or t.startswith("s_setpc"):
            falls = False
        if "ASMSTART" in t:
            inasm = True
            continue
        if "ASMEND" in t:
            inasm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if inasm:
            if t.startswith("global_load_dwordx4"):
                pending |= regs(t.split(",")[0])
                sb = t.split(",")[2].split()[0]
                # a scalar base must come from the s_mov_b64 in front ("off": the address is in the lane's registers)
                if sb != "off" and not (prev.startswith("s_mov_b64") or (prev.startswith("global_load_dwordx4") and sb in prev)):
                    findings.append((n, "asm load without s_mov_b64 in front: " + t))
            elif t.startswith("s_waitcnt vmcnt(0)"):
                pending = set()
            prev = t
            continue
        prev = t
        if not pending:
            continue
        touched = regs(t) & pending
        if not touched:
            continue
        if t.startswith("v_mfma"):
            # consumer: A operand (second register tuple) = ring registers; they are free afterwards
            ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
            pending -= regs(ops[1])
            bad = (regs(ops[0]) | regs(ops[2]) | regs(ops[3] if len(ops) > 3 else "")) & pending
            if bad:
                findings.append((n, "MFMA touches in-flight ring registers as non-A operand: " + t))
            continue
        findings.append((n, "touches in-flight ring registers %s: %s" % (sorted(touched)[:4], t)))
    return findings


def check(path):
    # split the text into kernels
    kernels, cur = [], None
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        # a function's label as LLVM prints it: "_ZN5camal...E:   ; @_ZN5camal...E" (the comment hid the colon from an
        # earlier version of this test, which then checked no kernel of real compiler output at all)
        head = t.split(";")[0].strip()
        if t.startswith(".amdhsa_kernel") or (head.endswith(":") and head.startswith("_Z") and "camal" in head):
            if head.endswith(":"):
                cur = (head[:-1], [])
                kernels.append(cur)
            continue
        if cur is not None:
            cur[1].append((n, t))
    findings = []
    check.kernels_seen = len(kernels)
    for name, lines in kernels:
        # The scan follows the text.  Blocks the compiler moved out of line (cold paths at the end of a kernel) and
        # loop headers are entered with the loads in flight at EVERY branch that reaches them, forward or backward:
        # iterate to a fixed point of at_branch (a copy or spill of an in-flight register that hipcc places at the top
        # of a loop, in front of the consuming MFMA, is what the hand-placed waits cannot protect against).
        at_branch = {}
        for _ in range(16):
            before = {k: set(v) for k, v in at_branch.items()}
            found = _scan(lines, at_branch)
            if at_branch == before:
                break
        findings += [(name, n, msg) for n, msg in found]
    return findings


if __name__ == "__main__":
    f = check(sys.argv[1])
    for k, n, msg in f[:40]:
        print(f"{k}:{n}: {msg}")
    print(f"{len(f)} finding(s)")
    sys.exit(1 if f else 0)
