#!/usr/bin/env python3
"""Build-time check of the fused ballot kernels' hand-waited loads (ising_ballot.hip issues them as inline assembly, so the
compiler does not know they are in flight): in the generated ISA no instruction may touch a register an inline-assembly
global_load_dwordx2 / x4 (lattice words, accept masks) wrote before the inline-assembly s_waitcnt vmcnt that covers it.
The scan is linear in text order, which is program order for the row loop these loads live in.  The looks at the completion
counters (dword loads) are part of the scan since round 4: a look either carries its wait in the same statement or is waited for
once, outside any loop, by the code that follows it in the text -- rounds 2-3 polled in a loop whose header held the wait, and
the compiler's loop-carried copy of the register sat between the load and the wait.
vmcnt counts loads AND stores, in issue order: the scan numbers every inline-assembly vector-memory operation (round 5: the split launches' word units keep several
rows of loads in flight across the stores of the rows before), and `s_waitcnt vmcnt(n)` leaves the n youngest of them out.  Operations the compiler issues itself
are not numbered -- they can only make a wait stricter than the scan assumes; inline-assembly operations must therefore not sit in branches of their own between a
load and its wait (the mirror stores of the split form are ordinary stores for that reason).
usage: check_asm_loads.py ising_ballot.s   (hipcc -S --cuda-device-only output)"""
import re, sys

def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()

def operands(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+\b", line):
        out |= regs(tok)
    return out

def check_quad(path):
    """ising_quad.hip: the word passes keep their accept masks in accumulation registers that inline assembly alone loads (global_load_dwordx4 a[..]) and
    reads (v_accvgpr_read_b32) -- the compiler must have no use of its own for any of them in those kernels (no copy of a register whose load is in flight
    can exist then), and nothing may spill.  The masks' registers are a[0 .. n): Q_DEPTH sets of MAXI quads and the spare one -- n is read off the inline
    assembly itself (its highest register), so the check follows the kernel's depth."""
    bad = 0
    lines = open(path).read().split("\n")

    def acc(body):
        return [int(x) for x in re.findall(r"\ba(\d+)\b", body)] + [int(y) for x in re.findall(r"\ba\[(\d+):(\d+)\]", body) for y in x]

    start = None
    for i, ln in enumerate(lines + [""]):
        m = re.match(r"^(_Z\w*quad_pass_kILi(\d+)E\w*):", ln)
        if m:
            start, kern = i, m.group(1)
        if start is None or "s_endpgm" not in ln:
            continue
        # one kernel: lines[start .. i]
        in_asm, mine, marks = False, 0, []
        for j in range(start, i + 1):
            if "#ASMSTART" in lines[j]:
                in_asm = True
            if "#ASMEND" in lines[j]:
                in_asm = False
            marks.append(in_asm)
            if in_asm:
                mine = max([mine] + [r + 1 for r in acc(lines[j].split(";")[0])])
        nasm = 0
        for j in range(start, i + 1):
            body = lines[j].split(";")[0].strip()
            touches = any(r < mine for r in acc(body))
            if touches and marks[j - start]:
                nasm += 1
            if touches and not marks[j - start]:  # (accumulation registers above the masks' are the compiler's to spill into)
                print(f"{path}:{j + 1}: {kern}: the compiler uses an accumulation register of the masks: {body}")
                bad += 1
            if re.match(r"\s*(scratch_|buffer_(load|store).*offen)", body):
                print(f"{path}:{j + 1}: {kern}: scratch access (a spill?): {body}")
                bad += 1
        print(f"{kern}: {nasm} inline-assembly statements on accumulation registers a[0 .. {mine}), none by the compiler")
        if nasm == 0:
            print(f"{path}: {kern}: no accumulation-register statement found: the check looks at the wrong thing")
            bad += 1
        start = None
    return bad


if len(sys.argv) > 2 and sys.argv[1] == "--quad":
    sys.exit(1 if check_quad(sys.argv[2]) else 0)

bad = 0
kern = None
lines = open(sys.argv[1]).read().split("\n")
i = 0
while i < len(lines):
    ln = lines[i]
    m = re.match(r"^(_Z\w*ballot_(?:update|split)_k\w*):", ln)
    if m:
        kern, pending, nloads, nops = m.group(1), {}, 0, 0
    if kern and "s_endpgm" in ln:
        print(f"{kern}: {nloads} inline-assembly loads checked")
        kern = None
    if kern:
        in_asm = i > 0 and "#ASMSTART" in lines[i - 1] or (i > 1 and "#ASMSTART" in lines[i - 2] and "#ASMEND" not in lines[i - 1])
        body = ln.split(";")[0].strip()
        if body and not body.endswith(":"):
            fused_wait = i + 1 < len(lines) and lines[i + 1].split(";")[0].strip().startswith("s_waitcnt vmcnt(0)")  # load and wait in ONE statement
            if in_asm and body.startswith("global_load_dword ") and fused_wait:
                used = operands(body.split(",", 1)[1])
            elif in_asm and (body.startswith("global_load_dwordx2") or body.startswith("global_load_dwordx4") or body.startswith("global_load_dword ")
                             or (body.startswith("global_atomic_add_x2") and " sc0" in body)):  # (a returning atomic: the next ticket)
                dst = regs(body.split()[1].rstrip(","))
                nops += 1
                for r in dst:
                    pending[r] = (nops, i + 1)
                nloads += 1
                used = operands(body.split(",", 1)[1])  # address operands of the load itself
            elif in_asm and body.startswith("global_store_dword"):
                nops += 1  # (a store issued as inline assembly takes a place in the queue)
                used = operands(body)
            elif in_asm and body.startswith("s_waitcnt") and "vmcnt" in body:
                n = int(re.search(r"vmcnt\((\d+)\)", body).group(1))
                # the n youngest operations may stay out: everything older has landed
                pending = {r: v for r, v in pending.items() if v[0] > nops - n}
                used = set()
            else:
                used = operands(body)
            for r in used & set(pending):
                print(f"{kern}: line {i + 1} touches v{r}, loaded by inline assembly at line {pending[r][1]} and not yet waited for: {body}")
                bad += 1
    i += 1
sys.exit(1 if bad else 0)
