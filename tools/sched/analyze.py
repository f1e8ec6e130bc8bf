#!/usr/bin/env python3
"""Issue-model simulation of a straight-line VALU block of a gfx950 listing, and of a greedy re-ordering of it.
Model (tools/micro/valu_issue.hip on MI355X): one wave issues an independent VALU instruction every ~2.14 cycles; a dependent
one waits 6 cycles after its producer issued (7.75 when the consumer reads three VGPRs)."""
import re
import sys

ISSUE = 2.14
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos", "v_exp", "v_log")


def regs(tok):
    """register units named by an operand token: v12, v[3:4], s5, s[2:3], vcc, exec, scc, |v3|, -v3 ..."""
    tok = tok.strip().strip("|").lstrip("-").strip("|")
    out = []
    m = re.fullmatch(r"([vsa])(\d+)", tok)
    if m:
        return [(m.group(1), int(m.group(2)))]
    m = re.fullmatch(r"([vsa])\[(\d+):(\d+)\]", tok)
    if m:
        return [(m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1)]
    if tok in ("vcc", "exec"):
        return [(tok, 0), (tok, 1)]
    if tok in ("vcc_lo", "exec_lo"):
        return [(tok[:-3], 0)]
    if tok in ("vcc_hi", "exec_hi"):
        return [(tok[:-3], 1)]
    if tok in ("scc", "m0"):
        return [(tok, 0)]
    return out


def parse(line):
    """-> (mnemonic, defs, uses).  VALU only: first operand is the destination (v_cmp*_e32 writes vcc implicitly;
    v_cndmask_e32 reads vcc implicitly; *_e64 compares name their SGPR-pair destination)."""
    line = line.split(";")[0].strip()
    mn, _, rest = line.partition(" ")
    ops = [o.strip() for o in rest.split(",")] if rest.strip() else []
    ops = [o for o in ops if not re.fullmatch(r"(clamp|mul:\d|div:\d|op_sel.*|neg_.*|row_.*|quad_perm.*|bank_mask.*|bound_ctrl.*|src\d_sel.*|dst_.*)", o)]
    defs, uses = [], []
    if mn.startswith("v_cmp") and not mn.startswith("v_cmpx"):
        if mn.endswith("_e32"):
            defs += [("vcc", 0), ("vcc", 1)]
            for o in ops[1:] if ops and ops[0] == "vcc" else ops:
                uses += regs(o)
        else:
            defs += regs(ops[0])
            for o in ops[1:]:
                uses += regs(o)
    else:
        if ops:
            defs += regs(ops[0])
        for o in ops[1:]:
            uses += regs(o)
        if mn.startswith(("v_fmac", "v_mac", "v_pk_fmac")):
            uses += regs(ops[0])
        if mn.startswith("v_cndmask") and mn.endswith("_e32"):
            uses += [("vcc", 0), ("vcc", 1)]
        if mn.startswith(("v_addc", "v_subb", "v_add_co", "v_sub_co", "v_div_fmas")):
            return mn, None, None   # carry users: leave alone
    uses.append(("exec", 0)); uses.append(("exec", 1))
    return mn, defs, uses


def movable(mn):
    return mn.startswith("v_") and not mn.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_cmpx", "v_accvgpr", "v_mfma", "v_div_", "v_mov_b32_dpp",
                                                       "v_permlane", "v_swap", "v_nop", "v_interp", "v_mbcnt")) and "dpp" not in mn and "sdwa" not in mn


def latency(mn_prod, mn_cons, cons_line):
    nv = len(re.findall(r"(?<![a-z_\[])v\d+|v\[\d+:\d+\]", cons_line.split(" ", 1)[1] if " " in cons_line else "")) - 1
    base = 7.75 if nv >= 3 else 6.0
    if mn_prod.startswith(TRANS):
        base += 4
    return base


def simulate(block):
    """in-order issue: t[i] = max(t[i-1] + ISSUE, max_p (t[p] + L))"""
    last_def = {}
    t = []
    now = -ISSUE
    for line in block:
        mn, defs, uses = parse(line)
        ready = now + ISSUE
        if mn.startswith(TRANS):
            ready = now + 2 * ISSUE
        if uses:
            for u in uses:
                if u in last_def and u[0] in ("v", "vcc"):
                    p, pmn = last_def[u]
                    ready = max(ready, t[p] + latency(pmn, mn, line))
        now = ready
        t.append(now)
        if defs:
            for d in defs:
                last_def[d] = (len(t) - 1, mn)
    return now + ISSUE


if __name__ == "__main__":
    lines = [l.strip() for l in open(sys.argv[1]) if l.strip()]
    start, n = int(sys.argv[2]), int(sys.argv[3])
    block = [l for l in lines[start:start + n]]
    valu = [l for l in block if l.startswith("v_")]
    print("block of", len(block), "instructions,", len(valu), "VALU")
    print("in-order simulated cycles: %.0f  (%.2f per instruction)" % (simulate(valu), simulate(valu) / len(valu)))
