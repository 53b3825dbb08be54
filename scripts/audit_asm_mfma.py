"""ISA audit for inline-asm MFMAs: hipcc's hazard recogniser does not look inside an `asm` statement, so a
compiler-generated vector instruction that WRITES a register an asm-issued MFMA is still READING (its A / B / C sources; the
MFMA fetches them over its first passes, not at issue) goes unpadded -- the write-after-read hazard that made the in-place
asm form of the stride-2 data gradient return wrong values (v_mov_b64 register shuffles of loop-carried accumulators
right behind the MFMA that read the old contents as its B operand).  For builtin MFMAs the compiler inserts the s_nops.

For every asm-issued v_mfma (between ;;#ASMSTART / ;;#ASMEND) this lists VALU instructions whose destination overlaps one
of its source registers (destination == C of the same MFMA excluded: that is the accumulate chain) and that issue fewer
than REQ wait states behind it: REQ = 5 / 11 / 19 for 4- / 8- / 16-pass MFMAs (the ISA's XDL-read-SrcC -> VALU-write
table, applied to A and B as well); counted as: another MFMA = 4 x its passes (the matrix pipe is busy that long before it
can even issue), `s_nop N` = N + 1, anything else = 1.  LDS / memory loads into those registers are not flagged (their
data arrives >= 64 cycles later).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude --cuda-device-only -S csrc/x.hip -o x.s
    python scripts/audit_asm_mfma.py x.s
"""
import re, sys

REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def passes(op):
    if "32x32x16" in op or "32x32x2" in op or "16x16x4_f32" in op or "32x32x64" in op:
        return 8 if "32x32x2_" not in op else 16
    return 4


def scan(lines, stash_in, collect):
    """One linear pass.  Control flow: the MFMAs still pending at a branch are carried to its target label (`stash`),
    nothing is pending behind an unconditional s_branch; run twice so that backward branches are covered too."""
    kernel, in_asm = None, False
    pending, stash, pend_dst = [], {}, []
    flagged, n_asm = {}, {}
    for no, raw in enumerate(lines, 1):
        s = raw.strip()
        if re.match(r"^[_A-Za-z0-9$.]+:", raw):
            name = raw.split(":")[0]
            if name.startswith(".L"):
                pending = pending + stash_in.get((kernel, name), [])
            else:
                kernel, pending, pend_dst = name, [], []
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        if op.startswith(("s_branch", "s_cbranch")):
            tgt = s.split()[-1]
            stash.setdefault((kernel, tgt), []).extend(pending)
            if op == "s_branch":
                pending = []
            continue
        if op.startswith(("v_mfma", "v_smfma")):
            ops = [t.strip() for t in s[len(op):].split(",")]
            k = 4 * passes(op)
            pending = [(a, b, c, d - k) for a, b, c, d in pending if d - k > 0]
            pend_dst = [(a, b, c, d - k) for a, b, c, d in pend_dst if d - k > 0]
            if in_asm and len(ops) >= 4:
                src = regs(ops[1]) | regs(ops[2]) | (regs(ops[3]) - regs(ops[0]))
                n_asm[kernel] = n_asm.get(kernel, 0) + 1
                pending.append((no, s, src, {4: 5, 8: 11, 16: 19}[passes(op)]))
                # the RESULT: a VALU / LDS / memory instruction that reads or overwrites the destination before the MFMA has
                # written it (passes + 3 wait states; the compiler pads builtin MFMAs, not asm ones) -- the accumulator
                # copies the register allocator makes around branches at high pressure
                pend_dst.append((no, s, regs(ops[0]), passes(op) + 4))
            continue
        if op.startswith("v_") and not in_asm:
            ops = [t.strip() for t in s[len(op):].split(",")]
            dst = regs(ops[0]) if ops else set()
            for mno, mtxt, src, _ in pending:
                if dst & src and (mno, no) not in flagged.setdefault(kernel, {}):
                    flagged[kernel][(mno, no)] = (mtxt, s)
        if not in_asm and op.startswith(("v_", "ds_write", "ds_store", "global_store", "buffer_store", "scratch_store", "flat_store")):
            touched = regs(s[len(op):])
            for mno, mtxt, dreg, _ in pend_dst:
                if touched & dreg and (mno, no) not in flagged.setdefault(kernel, {}):
                    flagged[kernel][(mno, no)] = (mtxt, s + "      ; touches a result in flight")
        k = 1
        if op == "s_nop":
            m = re.search(r"s_nop\s+(\d+)", s)
            k = int(m.group(1)) + 1 if m else 1
        pending = [(a, b, c, d - k) for a, b, c, d in pending if d - k > 0]
        pend_dst = [(a, b, c, d - k) for a, b, c, d in pend_dst if d - k > 0]
    return stash, flagged, n_asm


def main(path):
    lines = open(path).read().split("\n")
    stash, _, _ = scan(lines, {}, False)
    _, flagged, n_asm = scan(lines, stash, True)
    for k in sorted(n_asm):
        f = flagged.get(k, {})
        print(f"{k[:100]}: {n_asm[k]} asm MFMAs, {len(f)} VALU writes into live MFMA sources / accesses to results in flight")
        for (mno, no), (mtxt, s) in list(sorted(f.items()))[:6]:
            print(f"    line {mno}: {mtxt}\n      <- line {no}: {s}")
    return sum(len(v) for v in flagged.values())


if __name__ == "__main__":
    bad = main(sys.argv[1])
    print(f"{bad} hazards")
    sys.exit(1 if bad else 0)
