#!/usr/bin/env python3
"""Lint the generated gfx950 code of k_tile_render for the one thing hand-issued loads can get wrong: a register that is the
target of an inline-asm `global_load_*` must not be read or written by anything until the inline-asm statement that waits
for the load and moves its value out (the hardware has no interlock on a pending VMEM result, and the compiler does not know
these statements are loads, so it is free to copy or spill the variable anywhere).

    profiles/lint_inflight.py svgr_hip-hip-amdgcn-amd-amdhsa-gfx950.s      (from hipcc -save-temps)

Forward data flow over the control-flow graph of every k_tile_render instantiation: the state is the set of VGPRs with a
hand-issued load in flight (union at joins, iterated to a fixed point).  An asm `global_load` adds its destination; an asm
`s_waitcnt vmcnt(n)` is taken at the source's word -- the statement that contains it names, as sources of its moves, what
has landed -- and `vmcnt(0)` clears everything; any other instruction that mentions an in-flight register is reported."""
import re
import sys


def regs_of(tok):
    """registers a token names: VGPR n -> n, SGPR n -> 1000 + n (the scalar header load lands in SGPRs)"""
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", tok)
    if m:
        base = 0 if m.group(1) == "v" else 1000
        return set(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
    m = re.fullmatch(r"([vs])(\d+)", tok)
    return {(0 if m.group(1) == "v" else 1000) + int(m.group(2))} if m else set()


def parse(line):
    line = line.split(";")[0].strip()
    parts = line.split(None, 1)
    if not parts:
        return "", []
    if len(parts) < 2:
        return parts[0], []
    return parts[0], [t.strip().strip("|-") for t in re.split(r",\s*", parts[1])]


def check(name, text):
    # instructions: (op, operands, in_asm, source line); labels map to instruction indices
    # (an asm statement may park a body out of line: the lines between `.subsection 1` and `.subsection 0` are assembled
    #  behind the kernel's code, so the textual neighbours of such a region fall through to each other, not into it)
    ins, labels, in_asm, sub, is_sub, pending = [], {}, False, False, [], []
    for ln, l in enumerate(text):
        if "#ASMSTART" in l:
            in_asm = True
            continue
        if "#ASMEND" in l:
            in_asm = False
            continue
        m = re.match(r"^\s*(\.L\w+):", l)
        if m:
            pending.append((m.group(1), sub))  # (names the next instruction of ITS kind)
            continue
        m = re.match(r"^\s*\.subsection\s+(\d+)", l)
        if m:
            sub = m.group(1) != "0"
            continue
        op, ops = parse(l)
        if not op or op.startswith(".") or op.endswith(":"):
            continue
        for name_, kind_ in [x for x in pending if x[1] == sub]:
            labels[name_] = len(ins)
        pending = [x for x in pending if x[1] != sub]
        ins.append((op, ops, in_asm, ln + 1, l.strip()))
        is_sub.append(sub)
    n = len(ins)
    succ = [[] for _ in range(n)]

    def after(i):  # the instruction control falls through to: the next one of the same (in-line / out-of-line) kind
        j = i + 1
        while j < n and is_sub[j] != is_sub[i]:
            j += 1
        return [j] if j < n else []

    for i, (op, ops, _a, _ln, _l) in enumerate(ins):
        if op == "s_branch":
            succ[i] = [labels[ops[0]]]
        elif op.startswith("s_cbranch"):
            succ[i] = [labels[ops[0]]] + after(i)
        elif op == "s_endpgm":
            succ[i] = []
        else:
            succ[i] = after(i)
    state_in = [None] * n
    state_in[0] = frozenset()
    work = [0]
    bad = {}
    loads = 0
    while work:
        i = work.pop()
        st = set(state_in[i])
        op, ops, is_asm, ln, raw = ins[i]
        if is_asm and (op.startswith("global_load") or op.startswith("s_load")):
            st |= regs_of(ops[0])
        elif is_asm and op == "s_waitcnt" and "lgkmcnt(0)" in raw:
            st = {r for r in st if r < 1000}   # every scalar load has landed
            if "vmcnt(0)" in raw:
                st.clear()
        elif is_asm and op == "s_waitcnt" and "vmcnt" in raw:
            if "vmcnt(0)" in raw:
                st.clear()
            # (a counted wait: the moves that follow inside the same statement say what has landed)
        elif is_asm and op.startswith("v_mov") and len(ops) == 2:
            st -= regs_of(ops[1])  # the take: its source has landed (the wait in front of it belongs to the same statement)
        else:
            used = set()
            for t in ops:
                used |= regs_of(t)
            hit = used & st
            if hit:
                bad[ln] = f"{name}: line {ln}: `{raw}` touches {['s%d' % (r - 1000) if r >= 1000 else 'v%d' % r for r in sorted(hit)]} while a hand-issued load into it is in flight"
        out = frozenset(st)
        for j in succ[i]:
            new = out if state_in[j] is None else state_in[j] | out
            if new != state_in[j]:
                state_in[j] = new
                work.append(j)
    loads = sum(1 for op, _o, a, _ln, _l in ins if a and (op.startswith("global_load") or op.startswith("s_load")))
    for ln in sorted(bad):
        print(bad[ln])
    return len(bad), loads


def main(path):
    lines = open(path).read().split("\n")
    total_bad = 0
    funcs = [i for i, l in enumerate(lines) if re.match(r"^_Z13k_tile_render\w*:", l)]
    for f0 in funcs:
        name = lines[f0].split(":")[0]
        f1 = next(i for i in range(f0, len(lines)) if lines[i].startswith(".Lfunc_end"))
        nbad, loads = check(name, lines[f0 + 1:f1 + 1])
        total_bad += nbad
        print(f"{name}: {loads} hand-issued load instructions, {nbad} violations")
    print("lint_inflight:", "FAILED" if total_bad else "ok", f"({len(funcs)} instantiations)")
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
