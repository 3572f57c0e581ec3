"""Static check of csrc/gemm_wg.hip's generated code: no instruction may touch a register that an inline-asm global load is
still writing.

The kernel requests its operands with inline-asm loads and waits with inline-asm `s_waitcnt vmcnt(N)`; the compiler does
not know those registers are in flight, so a register copy (loop-carried value, coalescing decision) placed between a
request and its wait would read stale data - silently.  This script walks every path of the control-flow graph of every
wgrad_direct_kernel instantiation (basic blocks at the labels, edges from s_branch / s_cbranch / fall-through; a block is
revisited for every distinct queue it is entered with), keeps the queue of outstanding asm loads, retires all but the
youngest N at each asm `s_waitcnt vmcnt(N)`, and reports any other instruction that reads or writes an in-flight register.

usage: wg_check_isa.py <.s file> [wgrad_direct | gemm_rs]   (hipcc -S output for gfx950); exit status 1 on a finding.
The same walk checks csrc/gemm_rs.hip's asm LDS reads of the B operand against their asm `s_waitcnt lgkmcnt(N)`.
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


def split_blocks(lines):
    """[(label or None, [(line number, code, in_asm)])] in layout order; inline-asm markers folded into a flag."""
    blocks = [(None, [])]
    in_asm = False
    for ln, raw in lines:
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";"):
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            blocks.append((m.group(1), []))
            continue
        if t.startswith(".") or t.endswith(":"):
            continue
        code = t.split(";")[0].strip()
        if code:
            blocks[-1][1].append((ln, code, in_asm))
    return blocks


def check_kernel(name, lines, load_op="global_load", counter="vmcnt"):
    """Walk every path of the control-flow graph (states = the queue of outstanding asm loads; each (block, state) once)."""
    blocks = split_blocks(lines)
    index = {lab: i for i, (lab, _) in enumerate(blocks) if lab}
    findings = []
    seen = set()
    work = [(0, ())]
    while work:
        bi, state = work.pop()
        key = (bi, state)
        if key in seen or len(seen) > 20000:
            continue
        seen.add(key)
        inflight = [set(r) for r in state]
        fall = True
        for ln, code, in_asm in blocks[bi][1]:
            op = code.split()[0]
            if in_asm and op.startswith(load_op):
                dst = code.split(None, 1)[1].split(",")[0]
                busy = set().union(*inflight) if inflight else set()
                bad = regs(code.split(",", 1)[1]) & busy
                if bad and (ln, code, tuple(sorted(bad))) not in findings:
                    findings.append((ln, code, tuple(sorted(bad))))
                inflight.append(regs(dst))
                continue
            if in_asm and op == "s_waitcnt":
                m = re.search(counter + r"\((\d+)\)", code)
                if m:
                    n = int(m.group(1))
                    inflight = inflight[len(inflight) - n:] if n else []
                continue
            if op == "s_waitcnt" and counter + "(0)" in code:   # the compiler's own full wait retires everything too
                inflight = []
                continue
            if op == "s_endpgm":
                fall = False
                break
            if op == "s_branch":
                tgt = code.split()[1]
                work.append((index[tgt], tuple(frozenset(r) for r in inflight)))
                fall = False
                break
            if op.startswith("s_cbranch"):
                tgt = code.split()[1]
                work.append((index[tgt], tuple(frozenset(r) for r in inflight)))
                continue
            if inflight:
                busy = set().union(*inflight)
                bad = regs(code) & busy
                if bad and (ln, code, tuple(sorted(bad))) not in findings:
                    findings.append((ln, code, tuple(sorted(bad))))
        if fall and bi + 1 < len(blocks):
            work.append((bi + 1, tuple(frozenset(r) for r in inflight)))
    return sorted(findings)


def main(path, family="wgrad_direct"):
    # family "wgrad_direct": csrc/gemm_wg.hip (asm global loads, vmcnt; no scratch allowed at all);
    # family "gemm_rs": csrc/gemm_rs.hip's fp32 MFMA loop (asm ds_read_b32 of the B operand, lgkmcnt; a few instantiations
    # spill other registers - a spill of an in-flight one is found as an instruction that touches it)
    sym = {"wgrad_direct": "_ZN2gb19wgrad_direct_kernel", "gemm_rs": "_ZN2gb14gemm_rs_kernel"}[family]
    load_op, counter = ("global_load", "vmcnt") if family == "wgrad_direct" else ("ds_read", "lgkmcnt")
    kernels = {}
    cur = None
    for i, raw in enumerate(open(path), 1):
        m = re.match(r"^(" + sym + r"\w+):", raw)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is not None:
            if raw.startswith(".Lfunc_end"):   # (a kernel may hold several s_endpgm)
                cur = None
                continue
            kernels[cur].append((i, raw))
    if not kernels:
        print("no %s kernel in" % family, path)
        return 1
    status = 0
    if family == "wgrad_direct":
        # no scratch memory either: a spill is a store of a register - possibly of one still in flight
        text = open(path).read()
        for m in re.finditer(r"\.amdhsa_kernel (" + sym + r"\w+)(.*?)\.end_amdhsa_kernel", text, flags=re.S):
            sz = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2))
            if sz and int(sz.group(1)) != 0:
                print("%s: %s bytes of scratch per lane (spilled registers)" % (m.group(1), sz.group(1)))
                status = 1
    for name, lines in kernels.items():
        n_asm = sum(1 for _, raw in lines if raw.strip().startswith(load_op) )
        f = check_kernel(name, lines, load_op, counter)
        if family == "gemm_rs":   # ... and its A operand's asm global loads against their asm `s_waitcnt vmcnt(0)`
            f = sorted(set(f) | set(check_kernel(name, lines, "global_load", "vmcnt")), key=lambda x: x[0])
        print("%s: %d instructions, %s" % (name, len(lines), "ok" if not f else "%d finding(s)" % len(f)))
        for ln, code, bad in f[:20]:
            print("   line %d: %s   <- in flight: %s" % (ln, code, ", ".join("v%d" % r for r in bad)))
        if f:
            status = 1
    return status


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "wgrad_direct"))
