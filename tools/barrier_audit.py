#!/usr/bin/env python3
"""barrier_audit.py -- does every s_barrier of a gfx950 kernel wait for the wave's own LDS writes first?

    hipcc -S ... -o k.s ; python tools/barrier_audit.py k.s [more.s ...]       (csrc/Makefile: `make audit`)

__syncthreads() is fence(release, workgroup) + s_barrier + fence(acquire, workgroup); the release fence must become
`s_waitcnt lgkmcnt(0)` whenever one of the wave's LDS writes may still be in flight, or the other waves of the workgroup --
on other SIMDs, with their own path into the LDS -- can read the old value after the barrier.  The ROCm 7.2 compiler drops
that wait on some loop back edges (round 4: the level loop of depthsort.hip; the waves that read a stale level state left the
loop early and the workgroup's barriers no longer paired up).  The depth sort therefore uses gsr_sync() (gsr_depth_key.h: the
wait spelled out, then the barrier), and this script checks the ISA that comes out of every translation unit: a forward data
flow over each kernel's basic blocks, state = "an LDS read or write of this wave may be outstanding" (ds_bpermute and
ds_swizzle touch no memory), cleared by s_waitcnt lgkmcnt(0), reported at s_barrier.

Exit code 1 when a barrier can be reached with an access outstanding.
"""
import re
import sys

LABEL = re.compile(r"^(\.LBB\d+_\d+):")
FUNC = re.compile(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$")
BRANCH = re.compile(r"^\s*(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)")
LDS_ACCESS = re.compile(r"^\s*ds_(read|write|add|sub|rsub|inc|dec|min|max|and|or|xor|mskor|cmpst|wrxchg|wrap|append|consume|gws|ordered|swizzle_write|pk_add|condxchg)")
WAIT0 = re.compile(r"^\s*s_waitcnt\b(.*)")


def waits_lgkm0(rest):
    rest = rest.strip()
    if "lgkmcnt(0)" in rest:
        return True
    if re.fullmatch(r"(0x)?0+", rest):
        return True
    m = re.fullmatch(r"(0x[0-9a-fA-F]+|\d+)", rest)
    if m:
        v = int(rest, 0)
        return ((v >> 8) & 0xF) == 0
    return False


def audit(path):
    funcs = {}
    cur = None
    for ln, line in enumerate(open(path), 1):
        s = line.rstrip("\n")
        if s.startswith("\t.") or not s.strip() or s.lstrip().startswith(";"):
            continue
        m = LABEL.match(s)
        if m:
            if cur is not None:
                blk = {"name": m.group(1), "ins": [], "succ": [], "fall": True}
                cur["blocks"].append(blk)
            continue
        m = FUNC.match(s)
        if m and not s.startswith("\t"):
            cur = {"blocks": [{"name": "<entry>", "ins": [], "succ": [], "fall": True}]}
            funcs[m.group(1)] = cur
            continue
        if cur is None or not s.startswith("\t"):
            continue
        cur["blocks"][-1]["ins"].append((ln, s))
    bad = []
    for fname, f in funcs.items():
        blocks = f["blocks"]
        if not any("s_barrier" in i for b in blocks for _, i in b["ins"]):
            continue
        index = {b["name"]: k for k, b in enumerate(blocks)}
        # split blocks at branches: a conditional branch in the middle of a labelled block ends a basic block
        split = []
        remap = {}
        for b in blocks:
            remap[b["name"]] = len(split)
            part = {"ins": [], "succ": [], "fall": True}
            split.append(part)
            for ln, s in b["ins"]:
                part["ins"].append((ln, s))
                m = BRANCH.match(s)
                if m:
                    part["succ"].append(m.group(2))
                    part["fall"] = m.group(1) != "s_branch"
                    part = {"ins": [], "succ": [], "fall": True}
                    split.append(part)
                elif re.match(r"^\s*(s_endpgm|s_setpc_b64)", s):
                    part["fall"] = False
                    part = {"ins": [], "succ": [], "fall": True}
                    split.append(part)
        n = len(split)
        succ = []
        for k, b in enumerate(split):
            t = [remap[x] for x in b["succ"] if x in remap]
            if b["fall"] and k + 1 < n:
                t.append(k + 1)
            succ.append(t)
        state_in = [False] * n
        work = list(range(n))
        reported = set()
        while work:
            k = work.pop()
            pending = state_in[k]
            for ln, s in split[k]["ins"]:
                if LDS_ACCESS.match(s):
                    pending = True
                else:
                    m = WAIT0.match(s)
                    if m and waits_lgkm0(m.group(1)):
                        pending = False
                    elif re.match(r"^\s*s_barrier\b", s) and pending and ln not in reported:
                        reported.add(ln)
                        bad.append((path, fname, ln))
            for t in succ[k]:
                if pending and not state_in[t]:
                    state_in[t] = True
                    work.append(t)
    return bad, len(funcs)


def main():
    rc = 0
    for path in sys.argv[1:]:
        bad, nf = audit(path)
        print(f"{path}: {nf} functions, {len(bad)} barriers reachable with an LDS access outstanding")
        for p, fn, ln in bad:
            print(f"  {p}:{ln}  in {fn}")
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
