#!/usr/bin/env python3
"""Build-time guard for the inline-assembly load idiom of the dense kernels (`make check-isa`).

The streaming kernels issue some global loads through inline assembly so that the compiler, which cannot count past an
LDS-DMA, does not drain the memory queue in front of their first use (csrc/sddmm_kernels.hpp, denseStream's prologue).
The price: the compiler believes the destination registers are written when the asm statement ends.  Nothing stops it
from reading, copying or re-using such a register before the data has landed - round 2's K = 512 memory fault was a
`v_mov` of a loop-carried register the load had not filled yet (profiles/r02_dense_engines.md, trap 3).

This script reads the device assembly of the library (build/bsmr_capi.s, `make asm`) and, per kernel, walks the
control-flow graph.  A register written by an asm-issued load (between ;;#ASMSTART and ;;#ASMEND) is PENDING until an
`s_waitcnt vmcnt(N)` executes that can cover the load; any instruction that names a pending register - as a source
or as a destination - is a violation.  At joins the state is the pessimistic one (pending if pending on any path).
Two rules, by --strict:
  default  a wait covers the load if N is at most the number of vector-memory operations issued behind the load on SOME
           path to it (the kernels pick N at run time from the number of images they really issued - a tree of waits
           with different N, of which the one that executes is the right one; which one executes is not visible here).
           This is the rule that catches what happened: a register named with NO wait at all behind its load.
  --strict in-order completion taken literally on every path (N <= operations behind the load on EVERY path): also
           flags a wrong count, but reports the run-time wait trees as violations - for kernels without them.

  tools/check_isa.py FILE.s [--kernels REGEX] [--expect-violation]
exit status 0: clean (or, with --expect-violation, at least one violation found - the self-test on tools/probes/isa_trap.hip).
"""
import argparse
import re
import sys
from collections import defaultdict

VMEM = re.compile(r"^(global_(load|store|atomic)|buffer_(load|store|atomic)|scratch_(load|store)|flat_(load|store|atomic))")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LABEL = re.compile(r"^(\.?[A-Za-z_][\w.$]*):")
BRANCH = re.compile(r"^s_(cbranch_\w+|branch)\s+(\S+)")
WAIT = re.compile(r"vmcnt\((\d+)\)")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def kernels(path, pattern):
    """yield (name, [(line number, text, in_asm)])"""
    name, body, in_asm = None, [], False
    want = re.compile(pattern)
    with open(path) as f:
        for no, raw in enumerate(f, 1):
            line = raw.strip()
            if name is None:
                m = re.match(r"^(_Z\w+):", line)
                if m and want.search(m.group(1)):
                    name, body, in_asm = m.group(1), [], False
                continue
            if line.startswith(".Lfunc_end"):
                yield name, body
                name = None
                continue
            if ";;#ASMSTART" in line:
                in_asm = True
                continue
            if ";;#ASMEND" in line:
                in_asm = False
                continue
            code = line.split(";")[0].strip()
            if not code or code.startswith(".") and not LABEL.match(code):
                continue
            body.append((no, code, in_asm))


def check(name, body, strict=False):
    # basic blocks
    blocks, labels, cur = [], {}, []
    for item in body:
        m = LABEL.match(item[1])
        if m:
            if cur:
                blocks.append(cur)
            cur = []
            labels[m.group(1)] = len(blocks)
            continue
        cur.append(item)
        if BRANCH.match(item[1]) or item[1].startswith("s_endpgm"):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    succ = defaultdict(list)
    for i, blk in enumerate(blocks):
        last = blk[-1][1] if blk else ""
        m = BRANCH.match(last)
        if m:
            if m.group(2) in labels:
                succ[i].append(labels[m.group(2)])
            if m.group(1) != "branch" and i + 1 < len(blocks):
                succ[i].append(i + 1)
        elif not last.startswith("s_endpgm") and i + 1 < len(blocks):
            succ[i].append(i + 1)
    # state: {load line: (frozenset of registers, fewest younger operations on a path, most on a path)}, kept per
    # KNOWN CONDITION: the compiler likes to branch twice on one condition ("if more: gather, wait N" ... "if not more:
    # wait 0"), and a path that skips both waits does not exist.  What is known about vcc / scc / exec (zero or not)
    # when a conditional branch is taken or not taken travels with the state until something writes that register;
    # a later branch on it then has one successor.
    def kills(mnem, ops):
        dead = set()
        if mnem.startswith("s_") and not re.match(r"s_(cbranch|branch|waitcnt|nop|barrier|mov_b|load|sleep|setprio|endpgm|memtime|sendmsg)", mnem):
            dead.add("scc")
        if "vcc" in ops and not mnem.startswith("s_cbranch") and not mnem.startswith("v_cndmask"):
            dead.add("vcc")
        if "exec" in ops or mnem.startswith("v_cmpx") or "saveexec" in mnem:
            dead.add("exec")
        return dead

    def branch_fact(mnem, taken):
        m = re.match(r"s_cbranch_(vcc|scc|exec)(z|nz|0|1)$", mnem)
        if not m:
            return None
        zero = m.group(2) in ("z", "0")
        return m.group(1), ("zero" if zero == taken else "nonzero")

    state_in = [dict() for _ in blocks]     # block -> {facts (frozenset of (reg, value)): pending}
    state_in[0][frozenset()] = {}
    work = [(0, frozenset())]
    violations = {}
    rounds = 0
    while work and rounds < 200000:
        rounds += 1
        b, facts_in = work.pop()
        st = dict(state_in[b][facts_in])
        facts = dict(facts_in)
        for no, code, in_asm in blocks[b]:
            mnem = code.split()[0]
            ops = code[len(mnem):]
            named = regs(ops)
            for ld, (dst, lo, hi) in st.items():
                if ld != no and named & dst:
                    violations.setdefault((ld, no), code)
            w = WAIT.search(code) if mnem == "s_waitcnt" else None
            if w:
                n = int(w.group(1))
                st = {ld: v for ld, v in st.items() if (v[1] if strict else v[2]) < n}
            if VMEM.match(mnem):
                st = {ld: (dst, lo + 1, hi + 1) for ld, (dst, lo, hi) in st.items()}
                if in_asm and "_load" in mnem and "_lds" not in mnem:
                    first = ops.split(",")[0]
                    st[no] = (frozenset(regs(first)), 0, 0)
            if not mnem.startswith("s_cbranch"):
                # if / else on a lane mask: "s_and_saveexec ; s_xor ; s_cbranch_execz ELSE ; then ; ELSE: s_andn2_saveexec ;
                # s_cbranch_execz END ; else".  When the then-part was skipped because no lane took it, every lane takes the
                # else-part: exec is not zero there.
                else_runs = mnem == "s_andn2_saveexec_b64" and facts.get("exec") == "zero"
                for reg in kills(mnem, ops):
                    facts.pop(reg, None)
                if else_runs:
                    facts["exec"] = "nonzero"
                # the structurizer's flags: s_mov_b64 s[a:b], -1 / 0 ... s_andn2_b64 vcc, exec, s[a:b] ; s_cbranch_vccnz
                parts = [x.strip() for x in ops.split(",")]
                if parts and re.match(r"s\[\d+:\d+\]$", parts[0]):
                    facts.pop(parts[0], None)
                    if mnem == "s_mov_b64" and len(parts) == 2 and parts[1] in ("-1", "0"):
                        facts[parts[0]] = "ones" if parts[1] == "-1" else "zero"
                if mnem in ("s_andn2_b64", "s_and_b64") and len(parts) == 3 and parts[0] == "vcc" and parts[1] == "exec" and parts[2] in facts:
                    flag_set = facts[parts[2]] == "ones"
                    facts["vcc"] = ("nonzero" if flag_set else "zero") if mnem == "s_and_b64" else ("zero" if flag_set else "nonzero")
        last = blocks[b][-1][1].split()[0] if blocks[b] else ""
        m = BRANCH.match(blocks[b][-1][1]) if blocks[b] else None
        edges = []       # (successor, taken?)
        if m:
            if m.group(2) in labels:
                edges.append((labels[m.group(2)], True))
            if m.group(1) != "branch" and b + 1 < len(blocks):
                edges.append((b + 1, False))
        elif not last.startswith("s_endpgm") and b + 1 < len(blocks):
            edges.append((b + 1, None))
        for s_blk, taken in edges:
            out_facts = dict(facts)
            if taken is not None and m.group(1) != "branch":
                fact = branch_fact(last, taken)
                if fact:
                    reg, val = fact
                    if out_facts.get(reg, val) != val:
                        continue                      # this edge contradicts what is known: the path does not exist
                    out_facts[reg] = val
            key = frozenset(out_facts.items())
            old = state_in[s_blk].get(key)
            if old is None:
                if len(state_in[s_blk]) >= 16:        # (too many condition sets at one block: fold into the unconditioned one)
                    key = frozenset()
                    old = state_in[s_blk].get(key)
                if old is None:
                    state_in[s_blk][key] = dict(st)
                    work.append((s_blk, key))
                    continue
            merged, changed = dict(old), False
            for ld, (dst, lo, hi) in st.items():
                if ld not in merged:
                    merged[ld] = (dst, lo, hi)
                    changed = True
                elif merged[ld][1] > lo or merged[ld][2] < min(hi, 64):
                    merged[ld] = (dst, min(lo, merged[ld][1]), min(64, max(hi, merged[ld][2])))
                    changed = True
            if changed:
                state_in[s_blk][key] = merged
                work.append((s_blk, key))
    return violations


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--kernels", default=r"dense(Stream|Tiles|Shared|Groups|Sweep|Gemm)|sparseEntries|isaTrap")
    ap.add_argument("--expect-violation", action="store_true")
    ap.add_argument("--strict", action="store_true")
    args = ap.parse_args()
    total, checked, with_asm_loads = 0, 0, 0
    for name, body in kernels(args.asm, args.kernels):
        checked += 1
        if any(a and "_load" in c.split()[0] and VMEM.match(c.split()[0]) for _, c, a in body):
            with_asm_loads += 1
        v = check(name, body, args.strict)
        for (ld, use), code in sorted(v.items())[:5]:
            print(f"{name}: the register(s) loaded by the asm statement at line {ld} are named before a covering s_waitcnt, line {use}: {code}")
        total += len(v)
    print(f"check-isa: {checked} kernels, {with_asm_loads} with asm-issued loads, {total} violation(s)")
    if args.expect_violation:
        return 0 if total else 1
    return 1 if total or not checked else 0


if __name__ == "__main__":
    sys.exit(main())
