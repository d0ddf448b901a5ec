"""Build-time check of conv_wino.hip's hand-counted waits.  The Makefile runs it on the device assembly of the object it
links (`--asm FILE`: conv_wino.hip is compiled with -save-temps, so the .s is the one of that very compile -- same $(HIPCC),
same $(CXXFLAGS), DEV or not) and fails the build on a violation; without arguments it compiles the file itself (tests).

The kernel requests its transformed weights with `global_load_dwordx4` inside inline asm and waits for them with the
`s_waitcnt vmcnt(4)` that opens the multiply asm block.  The compiler believes the asm outputs are valid as soon as the
request has been issued, so any instruction IT places between the two that reads or writes those registers -- a copy
from live-range splitting, a spill, a reuse -- would see or destroy data that has not arrived.  This script compiles
the file to gfx950 assembly and checks, for every instantiation of the kernel:
  * the kernel has no scratch (no spills);
  * the multiply blocks that follow a weight-request block in program order (one or two, the loop taken round once)
    read the registers it writes as weight operands, every multiply block's 32 weight registers are written by request
    blocks, and no instruction outside the request / multiply asm blocks mentions them in between;
  * the K loop contains no compiler-generated `s_waitcnt vmcnt` (all of them come from the source) and no LDS
    instruction outside inline asm;
  * no vector-memory instruction INSIDE an asm block reads a scalar register (address pair, buffer descriptor) that a VALU
    instruction wrote fewer than five instructions earlier: the compiler's hazard recognizer
    inserts the wait states gfx9 needs there ("VALU writes SGPR -> VMEM reads that SGPR: 5") for its own instructions
    only, not for the contents of inline asm (round 5).
Exit code 0 = all good."""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "n-hans_amd", "csrc")


def vgprs(text):
    """set of VGPR numbers an instruction line mentions"""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


def sgprs(text):
    out = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bs(\d+)\b", text):
        out.add(int(a))
    if re.search(r"\bvcc\b", text):
        out.update((106, 107))
    return out


def valu_sgpr_writes(code):
    """scalar registers a VALU instruction writes: v_readfirstlane / v_readlane / e64 compares (first operand), the carry-out
    of the 64-bit forms (second operand), vcc of the e32 compares and carry forms"""
    t = code.strip()
    if not t.startswith("v_"):
        return set()
    op, _, rest = t.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    w = set()
    if op.startswith(("v_readfirstlane", "v_readlane")) or (op.startswith("v_cmp") and op.endswith("_e64")):
        w |= sgprs(ops[0])
    elif op.startswith("v_cmp"):
        w |= {106, 107}
    if op in ("v_mad_u64_u32", "v_mad_i64_i32") or (("_co_" in op or op.startswith("v_div_scale")) and op.endswith("_e64")):
        if len(ops) > 1:
            w |= sgprs(ops[1])
    elif "_co_" in op:
        w |= {106, 107}
    return w


def check_kernel(name, lines):
    errs = []
    # A developer instantiation (third template argument DBG = 1, DEV builds only) keeps its cycle stamps in an array
    # that lives in scratch: scratch stores outside the K loop are its stamps, and each one is one more vector-memory
    # instruction in front of the epilogue's counted wait (which then waits for MORE than it needs -- safe; the
    # product instantiations must have the exact count).
    dbg = re.search(r"conv_winoILi\d+ELi\d+ELi1E", name) is not None
    # (the descriptor follows the code; scratch shows as scratch_ instructions as well)
    if not dbg and any("scratch_" in l for l in lines):
        errs.append("scratch instructions (spills)")
    # asm blocks
    blocks, cur, inasm = [], None, False
    for i, l in enumerate(lines):
        if "#ASMSTART" in l:
            inasm, cur = True, [i, i, []]
        elif "#ASMEND" in l:
            inasm = False
            cur[1] = i
            blocks.append(cur)
        elif inasm:
            cur[2].append(l)
    # (a request block may be skipped as a whole: s_cmp / s_cbranch_scc0 to a local label in front of its loads)
    def is_load_line(l):
        t = l.strip()
        return not t or "global_load_dwordx4" in t or t.startswith("s_cmp_") or t.startswith("s_cbranch_scc") or re.match(r"^\.?L?\w*\d+:$", t) is not None
    loads = [b for b in blocks if b[2] and any("global_load_dwordx4" in l for l in b[2]) and all(is_load_line(l) for l in b[2])]
    mults = [b for b in blocks if any("v_mfma" in l for l in b[2])]
    if len(loads) < 3 or len(mults) < 4:
        return errs + ["expected >= 3 weight-request blocks and 4 multiply blocks, found %d / %d" % (len(loads), len(mults))]

    def load_regs(b):
        r = set()
        for l in b[2]:
            mm = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", l)
            if mm:
                r.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
        return frozenset(r)

    def mult_loads(b):
        r = set()
        for l in b[2]:
            mm = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", l)
            if mm:
                r.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
        return frozenset(r)

    def mult_regs(b):
        used, vops = set(), set()
        for l in b[2]:
            mm = re.search(r"v_mfma\S+ v\[\d+:\d+\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]", l)
            if mm:
                used.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
                vops.update(range(int(mm.group(3)), int(mm.group(4)) + 1))
        return frozenset(used), vops

    allw = set()
    for b in loads:
        allw |= load_regs(b)
    for b in mults:
        allw |= mult_loads(b)                              # (the next chunk's k-step 0 is requested inside the block)
    for b in mults:
        if not any("s_waitcnt vmcnt(4)" in l for l in b[2][:2]):
            errs.append("a multiply block does not open with s_waitcnt vmcnt(4)")
        w, v = mult_regs(b)
        if w & v:
            errs.append("an MFMA takes a weight register as its V operand")
        if len(w) != 32 or not w <= allw:
            errs.append("a multiply block's weight operands are not registers the request blocks write")
    # program order from the first request to the loop's backward branch, the loop body twice (wrap-around)
    first, last = loads[0][0], mults[-1][1]
    back = None
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    cands = []
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            cands.append((labels[m.group(1)], i))
    # the K loop: the smallest backward-branch span that holds the loop's multiply blocks (all but possibly none before it)
    inloop = [c for c in cands if sum(c[0] <= b[0] and b[1] <= c[1] for b in mults) >= 4]
    if inloop:
        back = min(inloop, key=lambda c: c[1] - c[0])
    if back is None:
        return errs + ["no backward branch found behind the multiply blocks"]
    order = list(range(first, back[1] + 1)) + list(range(back[0], back[1] + 1))
    kind = {}
    for b in loads:
        kind[b[0]] = ("load", load_regs(b), b)
    for b in mults:
        kind[b[0]] = ("mult", mult_regs(b)[0], b)
        if mult_loads(b):
            kind[b[1]] = ("load", mult_loads(b), b)        # its own requests, as an event at the end of the block
    skip = set()
    for b in loads + mults:
        skip.update(range(b[0], b[1] + 1))
    # every request: its registers stay untouched until the multiply blocks that consume them (the next one or two in
    # program order) have run
    for pos, i in enumerate(order[:back[1] + 1 - first]):
        if i not in kind or kind[i][0] != "load":
            continue
        R = kind[i][1]
        nm, touched, lastgood = 0, [], None
        for j in order[pos + 1:]:
            if j in kind and kind[j][0] == "mult":
                if not R <= kind[j][1]:
                    break
                nm += 1
                lastgood = len(touched)
                if nm == 2:
                    break
            elif j in kind and kind[j][0] == "load":
                if kind[j][1] & R:
                    break
            elif j not in skip:
                code = lines[j].split(";")[0]
                if vgprs(code) & R:
                    touched.append((j, lines[j].strip()))
        if nm == 0:
            errs.append("request at line %d: the next multiply block reads other registers" % i)
        for j, t in touched[:lastgood or 0]:
            errs.append("line %d touches a weight register between request and use: %s" % (j, t))
    # the epilogue's constants: asm requests (64-bit address, "off"), then compiler-visible residual requests, then the
    # hand-written s_waitcnt vmcnt(N): exactly N vector-memory instructions lie between the last constants block and
    # the wait, and nothing in between mentions the constants' registers
    def is_const_block(b):
        ld = [l.split(";")[0].rstrip() for l in b[2] if "global_load_dwordx4" in l]
        return len(ld) in (2, 4) and len(ld) == len([l for l in b[2] if l.strip()]) and all(l.endswith(("off", "offset:16")) for l in ld)
    consts = [b for b in blocks if is_const_block(b)]
    waits = [b for b in blocks if len([l for l in b[2] if l.strip()]) == 1 and re.search(r"s_waitcnt vmcnt\(\d+\)\s*$", b[2][0])]
    if len(consts) < 2:
        errs.append("epilogue: constants request blocks not found")
    # Walked along the CONTROL FLOW, not down the text (round 6: with several epilogue instantiations in one kernel the
    # compiler lays their blocks out interleaved -- a linear scan from a request to "the next wait" ran through another
    # instantiation's code): from every constants block, every path is followed through branches and fall-throughs
    # until it reaches a hand-written wait; on the way no instruction may mention a requested register, a later constants
    # block may not load into one, and the wait's count must equal the vector-memory instructions issued since the LAST
    # constants block on that path.
    labels_all = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    block_at = {b[0]: b for b in blocks}
    const_at = {c[0] for c in consts}
    wait_at = {w[0] for w in waits if w[0] > back[1]}
    VM = re.compile(r"\s*(global_|buffer_|scratch_|flat_)")
    seen_err = set()

    def err(msg):
        if msg not in seen_err:
            seen_err.add(msg)
            errs.append(msg)

    checked_waits = set()
    for c in consts:
        if c[0] < back[1]:
            continue
        # state: (line, tracked registers as a frozenset of (register, request line), count since the last constants block)
        stack = [(c[1] + 1, frozenset((r, c[0]) for r in load_regs(c)), 0, 0)]
        seen = set()
        while stack:
            i, tracked, cnt, steps = stack.pop()
            while True:
                if (i, tracked, cnt) in seen or i >= len(lines):
                    break
                seen.add((i, tracked, cnt))
                steps += 1
                if steps > 4000:
                    err("epilogue: no counted wait within 4000 instructions of the constants request at line %d" % c[0])
                    break
                if i in block_at:
                    b = block_at[i]
                    if i in wait_at:
                        n = int(re.search(r"vmcnt\((\d+)\)", b[2][0]).group(1))
                        checked_waits.add(i)
                        if cnt < n or (cnt != n and not dbg):
                            err("epilogue: %d vector-memory instructions between the constants and s_waitcnt vmcnt(%d) (line %d)" % (cnt, n, i))
                        break
                    regs_t = {r for r, _ in tracked}
                    if i in const_at:
                        if load_regs(b) & regs_t:
                            err("epilogue: a register of the constants request at line %d is touched before the wait (line %d): a second request loads into it"
                                % (c[0], i))
                            break
                        tracked = tracked | frozenset((r, i) for r in load_regs(b))
                        cnt = 0
                    else:
                        for l in b[2]:
                            code = l.split(";")[0]
                            if vgprs(code) & regs_t:
                                err("epilogue: a register of the constants request at line %d is touched before the wait (line %d): %s"
                                    % (c[0], i, l.strip()[:80]))
                            if VM.match(code):
                                cnt += 1
                    i = b[1] + 1
                    continue
                code = lines[i].split(";")[0]
                t = code.strip()
                if not t or t.startswith((".", "#")) or t.endswith(":"):
                    i += 1
                    continue
                hit = vgprs(code) & {r for r, _ in tracked}
                if hit:
                    origin = sorted({o for r, o in tracked if r in hit})[0]
                    err("epilogue: a register of the constants request at line %d is touched before the wait (line %d): %s"
                        % (origin, i, t[:80]))
                    break
                if VM.match(code):
                    cnt += 1
                if t.startswith("s_endpgm"):
                    err("epilogue: the constants request at line %d reaches the end of the kernel without a counted wait" % c[0])
                    break
                m = re.match(r"(s_branch|s_cbranch\w*)\s+(\.LBB\d+_\d+)", t)
                if m and m.group(2) in labels_all:
                    if m.group(1) == "s_branch":
                        i = labels_all[m.group(2)]
                        continue
                    stack.append((labels_all[m.group(2)], tracked, cnt, steps))
                i += 1
    if consts and not checked_waits:
        errs.append("epilogue: no counted wait reached from the constants requests")
    inasm = False
    for i in range(first, back[1] + 1):
        l = lines[i]
        if "#ASMSTART" in l:
            inasm = True
        elif "#ASMEND" in l:
            inasm = False
        code = l.split(";")[0]
        if not inasm and re.search(r"s_waitcnt.*vmcnt", code) and "vmcnt(12)" not in code and "vmcnt(4)" not in code:
            errs.append("line %d: a wait the source does not contain: %s" % (i, l.strip()))
        if not inasm and re.match(r"\s*ds_", code):
            errs.append("line %d: compiler-visible LDS access inside the K loop: %s" % (i, l.strip()))
        if re.match(r"\s*scratch_", code):
            errs.append("line %d: scratch access inside the K loop: %s" % (i, l.strip()))
    # asm-boundary hazard: VALU writes SGPR -> vector-memory instruction inside an asm block reads it (5 wait states)
    code_lines = [(i, l.split(";")[0]) for i, l in enumerate(lines)]
    code_lines = [(i, c) for i, c in code_lines if c.strip() and not c.strip().startswith((".", "#")) and not c.strip().endswith(":")]
    inasm_at = set()
    for b in blocks:
        inasm_at.update(range(b[0], b[1] + 1))
    for pos, (i, c) in enumerate(code_lines):
        if i not in inasm_at or not re.match(r"\s*(global_|buffer_|flat_)", c):
            continue
        need = sgprs(c.split(None, 1)[1]) if len(c.split(None, 1)) > 1 else set()
        back = code_lines[max(0, pos - 5):pos]
        for d, (j, p) in enumerate(reversed(back)):
            ws = 0
            m = re.match(r"\s*s_nop (\d+)", p)
            w = valu_sgpr_writes(p)
            if w & need:
                # wait states between the two: the instructions in between, an s_nop N counting N + 1
                ws = sum((int(re.match(r"\s*s_nop (\d+)", q).group(1)) + 1) if re.match(r"\s*s_nop (\d+)", q) else 1
                         for _, q in back[len(back) - d:])
                if ws < 5:
                    errs.append("line %d: %s reads a scalar register a VALU instruction wrote %d wait states earlier (line %d: %s)"
                                % (i, c.strip()[:60], ws, j, p.strip()[:60]))
    return errs


def main():
    args = sys.argv[1:]
    if args and args[0] == "--asm":
        # the device assembly of the ACTUAL build (the Makefile compiles conv_wino.hip with -save-temps and hands the .s
        # of that very invocation over: same compiler, same flags, DEV or not)
        text = open(args[1]).read().split("\n")
    else:
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "w.s")
            cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only",
                   "-fno-slp-vectorize", "-S", os.path.join(CSRC, "conv_wino.hip"), "-o", out] + args
            subprocess.run(cmd, check=True, cwd=CSRC)
            text = open(out).read().split("\n")
    # split into kernels
    kernels, cur = {}, None
    for l in text:
        m = re.match(r"^(_ZN5nhans9conv_wino\w+):", l)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif cur is not None:
            if l.startswith(".Lfunc_end"):
                cur = None
            else:
                kernels[cur].append(l)
    if not kernels:
        print("no conv_wino kernel found")
        return 1
    rc = 0
    for k, lines in kernels.items():
        errs = check_kernel(k, lines)
        print(k, "OK" if not errs else "FAILED")
        for e in errs[:20]:
            print("   ", e)
        rc |= bool(errs)
    return rc


if __name__ == "__main__":
    sys.exit(main())
