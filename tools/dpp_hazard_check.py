#!/usr/bin/env python3
"""Static check of the one hazard the compiler cannot see: a DPP operand read needs two wait states behind the VALU write of that
register (CDNA3 ISA 4.5), and the diagonal block's v_fmac_f64_dpp / v_mov_b64_dpp sit inside asm statements.

usage: dpp_hazard_check.py <device .s file>   (hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S csrc/kernels.hip)
Exit code 1 when any function holds a DPP read closer than two wait states to the instruction that wrote its source."""
import re
import sys


def regs(tok):
    m = re.match(r"-?\|?v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"-?\|?v(\d+)\b", tok)
    return {int(m.group(1))} if m else set()


def check(lines):
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    total_dpp = total_bad = 0
    for idx, (i0, name) in enumerate(starts):
        i1 = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
        body = [l.strip() for l in lines[i0:i1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        for k, l in enumerate(body):
            if "_dpp" not in l:
                continue
            total_dpp += 1
            ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
            src = regs(ops[1].split()[0])
            ws, kk = 0, k - 1
            while kk >= 0 and ws < 2:
                pl = body[kk]
                if pl.startswith("s_nop"):
                    ws += int(pl.split()[1]) + 1
                else:
                    if pl.startswith("v_"):
                        dst = regs(pl.split(None, 1)[1].split(",")[0].strip())
                        if dst & src:
                            total_bad += 1
                            print(f"{name}: HAZARD  {pl}  ->  {l}")
                    ws += 1
                kk -= 1
    return total_dpp, total_bad


if __name__ == "__main__":
    n, bad = check(open(sys.argv[1]).read().split("\n"))
    print(f"{n} DPP instructions, {bad} closer than two wait states to the write of their source")
    sys.exit(1 if bad else 0)
