"""Static check of csrc/gemm_wg.hip's generated code: no instruction may touch a register that an inline-asm global load is
still writing.

The kernel requests its operands with inline-asm loads and waits with inline-asm `s_waitcnt vmcnt(N)`; the compiler does
not know those registers are in flight, so a register copy (loop-carried value, coalescing decision) placed between a
request and its wait would read stale data - silently.  This script walks the assembly of every wgrad_direct_kernel
instantiation in program order, (loops: every block is walked as it is laid out, twice, so state carries over back
edges that jump upwards), keeps the queue of outstanding asm loads, retires all but the youngest N at each asm
`s_waitcnt vmcnt(N)`, and reports any other instruction that reads or writes an in-flight register.

usage: wg_check_isa.py <gemm_wg .s file>      (hipcc -S / -save-temps output for gfx950); exit status 1 on a finding
"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_kernel(name, lines):
    findings = []
    inflight = []          # [(set of registers, text)]
    in_asm = False
    for rep in range(2):   # second pass: state carried over the loop back edges
        for ln, raw in lines:
            t = raw.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            code = t.split(";")[0].strip()
            if not code:
                continue
            op = code.split()[0]
            if in_asm and op.startswith("global_load"):
                dst = code.split(None, 1)[1].split(",")[0]
                used = regs(code.split(",", 1)[1])
                bad = used & set().union(*[r for r, _ in inflight]) if inflight else set()
                if bad and (ln, code, sorted(bad)) not in findings:
                    findings.append((ln, code, sorted(bad)))
                inflight.append((regs(dst), code))
                continue
            if in_asm and op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", code)
                if m:
                    n = int(m.group(1))
                    inflight = inflight[len(inflight) - n:] if n else []
                continue
            if op == "s_waitcnt" and "vmcnt(0)" in code:   # the compiler's own full wait retires everything too
                inflight = []
                continue
            if op == "s_endpgm":
                inflight = []
                continue
            if not inflight:
                continue
            busy = set().union(*[r for r, _ in inflight])
            bad = regs(code) & busy
            if bad and (ln, code, sorted(bad)) not in findings:
                findings.append((ln, code, sorted(bad)))
    return findings


def main(path):
    kernels = {}
    cur = None
    for i, raw in enumerate(open(path), 1):
        m = re.match(r"^(_ZN2gb19wgrad_direct_kernel\w+):", raw)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is not None:
            kernels[cur].append((i, raw))
            if "s_endpgm" in raw:
                cur = None
    if not kernels:
        print("no wgrad_direct_kernel in", path)
        return 1
    status = 0
    for name, lines in kernels.items():
        f = check_kernel(name, lines)
        print("%s: %d instructions, %s" % (name, len(lines), "ok" if not f else "%d finding(s)" % len(f)))
        for ln, code, bad in f[:20]:
            print("   line %d: %s   <- in flight: %s" % (ln, code, ", ".join("v%d" % r for r in bad)))
        if f:
            status = 1
    return status


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
