#!/usr/bin/env python3
"""Lint for the hand-issued LDS reads of the conv kernels: the 8-wave Winograd kernels (conv_wino.h: conv_wino2_kernel,
conv_wino4_kernel; conv_wino44.h: conv_wino44_kernel), the direct implicit-GEMM kernel (conv_kernel.h: conv_kernel) and the vector-ALU head kernel (conv_n8.h).

The main loop issues ds_read* through inline asm and waits with `s_waitcnt lgkmcnt(N)`, so the compiler does not
know those registers are written asynchronously.  This scans gfx950 assembly (hipcc -S --cuda-device-only) and
fails if, inside a kernel, any instruction touches a register that a still-outstanding ds_read will write
(LDS returns in order: after lgkmcnt(N) only the N newest reads are outstanding).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S --cuda-device-only -o wino.s conv_inst_wino.hip
    python tools/check_async_lds.py wino.s
"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def main(path):
    bad = 0
    kernel, pending, in_wino2 = None, [], False
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            kernel, pending = m.group(1), []
            in_wino2 = ("conv_wino2_kernel" in kernel or "conv_wino4_kernel" in kernel or "conv_wino44_kernel" in kernel or
                        "11conv_kernelI" in kernel or "conv_n8_kernel" in kernel)
            continue
        if not in_wino2:
            continue
        code = line.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        parts = code.replace(",", " ").split()
        op, args = parts[0], parts[1:]
        touched = set()
        for a in args:
            touched |= regs(a)
        if op.startswith("ds_read"):
            dst = regs(args[0])
            addr = regs(args[1])
            for d in pending:
                if (addr | dst) & d:
                    print("%s:%d: %s uses a register of an outstanding ds_read: %s" % (path, ln, kernel[:60], code))
                    bad += 1
            pending.append(dst)
            continue
        if op.startswith("ds_write"):
            # LDS writes count in lgkmcnt too and complete in order with the reads: an entry without registers
            for d in pending:
                if touched & d:
                    print("%s:%d: %s stores a register of an outstanding ds_read: %s" % (path, ln, kernel[:60], code))
                    bad += 1
                    break
            pending.append(set())
            continue
        m = re.match(r"s_waitcnt.*lgkmcnt\((\d+)\)", code)
        if m:
            n = int(m.group(1))
            pending = pending[len(pending) - n:] if n else []
            continue
        if op in ("s_endpgm",):
            pending = []
            continue
        for d in pending:
            if touched & d:
                print("%s:%d: %s touches a register of an outstanding ds_read: %s" % (path, ln, kernel[:60], code))
                bad += 1
                break
    print("%s: %d hazards" % (path, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
