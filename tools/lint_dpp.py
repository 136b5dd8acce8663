"""ISA lint for the inline-asm DPP multiply-adds of stem123w.h (and any other kernel that uses them).

hipcc pads nothing for an instruction inside an `asm` statement (cdna_hip_programming.md section 5.7): a DPP read of a VGPR needs two
wait states after a VALU write of that VGPR, and five after a VALU write of EXEC (v_cmpx / v_readlane-class).  The DPP operand
of `v_fmac_f32_dpp` here is a launch-constant weight register, so the compiler has no reason to write it inside the loops -- but a
live-range split or a spill reload could.  This script walks the compiler's assembly (hipcc -save-temps) and fails if, in the
two instructions in front of a DPP instruction (fall-through order; a label in between is reported as a warning, the check then
also looks through the label), something writes its DPP source register, or a v_cmpx sits within five.

usage: python tools/lint_dpp.py file.s [kernel-name-substring]"""
import re
import sys


def regs_of(tok):
    tok = tok.strip()
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    lines = open(path).read().splitlines()
    kernel = None
    window = []          # (text, dest-regs, is_label)
    n_dpp = bad = warn = 0
    for ln in lines:
        s = ln.strip()
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kernel = m.group(1)
            window = []
            continue
        if not s or s.startswith(";") or s.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", s):
                window.append((s, set(), True))
            continue
        if kernel is None or (want and want not in kernel):
            continue
        if s.startswith(";;#ASM"):
            continue
        code = s.split(";")[0].strip()
        mnem = code.split()[0]
        ops = code[len(mnem):].split(",")
        if "_dpp" in mnem or " quad_perm:" in code or " row_shr:" in code or " row_shl:" in code:
            n_dpp += 1
            src = regs_of(ops[1].split()[0]) if len(ops) > 1 else set()
            states = 0                                    # wait states between the candidate writer and this instruction
            for text, dst, is_label in reversed(window):
                if is_label:
                    warn += 1
                    continue
                m0 = text.split()[0]
                if m0 == "s_nop":
                    states += int(text.split()[1], 0) + 1
                    continue
                if states < 2 and (dst & src) and m0.startswith("v_"):
                    print(f"{kernel}: DPP source written {states} wait state(s) before: '{text}' -> '{code}'")
                    bad += 1
                if states < 5 and m0.startswith(("v_cmpx", "v_readlane", "v_readfirstlane")) and "exec" in text.split(",")[0]:
                    print(f"{kernel}: EXEC written by VALU {states} wait state(s) before '{code}': '{text}'")
                    bad += 1
                states += 1
                if states >= 5:
                    break
        dst = set()
        if mnem.startswith(("v_", "ds_read", "buffer_load", "global_load", "scratch_load", "flat_load")) and ops and not mnem.startswith("v_cmp"):
            dst = regs_of(ops[0])
        window.append((code, dst, False))
        if len(window) > 12:
            window.pop(0)
    print(f"lint_dpp: {n_dpp} DPP instructions checked in kernels matching '{want}', {bad} hazards, {warn} labels looked through")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
