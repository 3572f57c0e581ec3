"""csrc/gemm_rs.hip: what sits between the asm request of the next A chunk and the asm wait behind the MFMAs.
A compiler-placed `s_waitcnt vmcnt(N)` in that window would wait for the request at once (the compiler cannot see the
asm loads); a branch or label would mean the window is not one basic block.  usage: rs_window.py <gemm_rs .s file>"""
import re, sys
name = None; inasm = False; win = None; rep = {}
for i, l in enumerate(open(sys.argv[1]), 1):
    m = re.match(r'^_ZN2gb14gemm_rs_kernel(\w+?)EEvNS', l)
    if m: name = m.group(1); win = None; continue
    t = l.strip()
    if t.startswith(';;#ASMSTART'): inasm = True; continue
    if t.startswith(';;#ASMEND'): inasm = False; continue
    if not t or t.startswith(';'): continue
    if inasm and t.startswith('global_load_dwordx4'):
        if win is None: win = {'start': i, 'mfma': 0, 'bad': []}
        continue
    if win is None: continue
    if inasm and t.startswith('s_waitcnt vmcnt(0)'):
        rep.setdefault(name, []).append(win); win = None; continue
    if t.startswith('v_mfma'): win['mfma'] += 1
    elif not inasm and t.startswith('s_waitcnt') and 'vmcnt' in t: win['bad'].append((i, t))
    elif t.startswith('s_cbranch') or t.startswith('s_branch') or re.match(r'^\.LBB', t): win['bad'].append((i, t.split(';')[0].strip()))
status = 0
for k, ws in rep.items():
    for w in ws:
        waits = [b for b in w['bad'] if b[1].startswith('s_waitcnt')]
        branchy = any(not b[1].startswith('s_waitcnt') for b in w['bad'])
        flag = ('branches inside (the column-group tail path: one tile per launch)' if branchy else
                'DRAINED by the compiler' if waits else 'ok')
        if waits and not branchy: status = 1
        print('%-16s window at line %d: %3d MFMAs inside, %s %s' % (k, w['start'], w['mfma'], flag, waits[:3] if waits else ''))
sys.exit(status)
