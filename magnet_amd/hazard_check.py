"""Build-time check of the shipped code objects for the hazards hipcc does not cover behind inline asm: no v_mfma may read a VGPR that a
VALU instruction wrote fewer than two wait states earlier; no VALU instruction may read an MFMA result before the MFMA's passes + 2;
no VALU instruction may read a transcendental's result in the next issue slot.

Why this exists.  On gfx950 a VALU write of a VGPR needs two wait states before a v_mfma reads that register as
SrcA / SrcB / SrcC.  hipcc's hazard recognizer keeps them for every VALU instruction it emits itself, but it does not look inside inline
asm: a `v_fma_mixhi_f16` written as asm (the fp16 split of this library) directly in front of the MFMA that consumes its result is
left alone, the MFMA reads the register's OLD contents now and then, and the kernel is wrong in one k element of one operand -
intermittently, depending on how busy the matrix pipe is when the pair issues (the round-2 "biases from LDS" failure,
scripts/dev/stress_bias_lds.py).  The sources close every asm block that can feed an MFMA with the missing wait states
(csrc/common.cuh: MAGNET_MIX_TAIL, mfma_operand_fence); this module verifies the RESULT: it disassembles libmagnet_hip.so's gfx950 code
objects and walks every kernel linearly.

    python -m magnet_amd.hazard_check [path/to/lib.so]      -> exit code 1 and a listing when a violation exists
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile

def objdump() -> str:
    """llvm-objdump of the ROCm installation that built the library: $ROCM_PATH, the directory hipcc lives in, /opt/rocm.  The check is
    not optional (a library that was not checked is not shipped): FileNotFoundError when none of them has it."""
    roots = [os.environ.get("ROCM_PATH"), os.environ.get("HIP_PATH")]
    hipcc = shutil.which("hipcc")
    if hipcc:
        roots.append(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))))
    roots.append("/opt/rocm")
    tried = []
    for r in roots:
        if not r:
            continue
        for rel in ("lib/llvm/bin/llvm-objdump", "llvm/bin/llvm-objdump", "bin/llvm-objdump"):
            cand = os.path.join(r, rel)
            tried.append(cand)
            if os.path.exists(cand):
                return cand
    which = shutil.which("llvm-objdump")
    if which:
        return which
    raise FileNotFoundError("llvm-objdump not found (looked in " + ", ".join(dict.fromkeys(tried)) + " and on PATH): the hazard check of the "
                            "built library cannot run; set ROCM_PATH")


MIN_WAIT_STATES = 2
SEAM = 24                # instructions looked at on either side of a backward branch (covers a 16-pass MFMA + 2)
_RANGE = re.compile(r"^([va])\[(\d+):(\d+)\]$")
_ONE = re.compile(r"^([va])(\d+)$")


def _regs(tok: str) -> list:
    """registers of one operand as ('v' | 'a', index) - architectural and accumulation VGPRs are different files"""
    tok = tok.strip()
    m = _RANGE.match(tok)
    if m:
        return [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    m = _ONE.match(tok)
    if m:
        return [(m.group(1), int(m.group(2)))]
    return []


class Kernel(list):
    """the instruction texts of one kernel (a list, as the rules take it) + each instruction's byte address and, for branches, the
    address they jump to (None otherwise) - what loop_seams() needs"""

    def __init__(self, ins=(), addr=(), target=()):
        super().__init__(ins)
        self.addr, self.target = list(addr), list(target)


def disassemble(lib: str) -> list[tuple[str, "Kernel"]]:
    """[(kernel name, Kernel [instruction text ...])] of every gfx950 code object bundled in ``lib``"""
    tool = objdump()
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        try:
            subprocess.run([tool, "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
            objs = [f for f in sorted(os.listdir(tmp)) if "gfx950" in f]
            texts = [subprocess.run([tool, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout for f in objs]
        except subprocess.CalledProcessError as e:
            raise RuntimeError(f"hazard check: {' '.join(e.cmd)} failed with code {e.returncode}:\n{(e.stderr or b'')[-2000:]}") from e
        if not objs:
            raise RuntimeError(f"hazard check: {lib} bundles no gfx950 code object")
        for txt in texts:
            cur = None
            for line in txt.splitlines():
                m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
                if m:
                    cur = (m.group(2), Kernel())
                    cur[1].base = int(m.group(1), 16)
                    out.append(cur)
                elif cur is not None and line.startswith("\t"):
                    parts = line.split("//")
                    ins = parts[0].strip()
                    if not ins:
                        continue
                    am = re.match(r"\s*([0-9A-Fa-f]+):", parts[1]) if len(parts) > 1 else None
                    tm = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", parts[1]) if len(parts) > 1 and ins.startswith(("s_cbranch", "s_branch")) else None
                    cur[1].append(ins)
                    cur[1].addr.append(int(am.group(1), 16) if am else None)
                    cur[1].target.append(cur[1].base + int(tm.group(1), 16) if tm else None)
    return out


def loop_seams(kernels):
    """A linear walk never pairs the last instructions of a loop body with the first ones of the next iteration.  For every BACKWARD
    branch this returns a pseudo kernel: the SEAM instructions up to and including the branch, followed by the SEAM instructions at its
    target - the order they execute in when the branch is taken.  [(name @ branch address, [instruction ...], index of the first
    instruction behind the seam)]"""
    out = []
    for name, ins in kernels:
        addr, target = getattr(ins, "addr", None), getattr(ins, "target", None)
        if not addr:
            continue
        index_of = {a: i for i, a in enumerate(addr) if a is not None}
        for b, tgt in enumerate(target):
            if tgt is None or addr[b] is None or tgt > addr[b] or tgt not in index_of:
                continue
            t = index_of[tgt]
            head = list(ins[max(0, b - SEAM + 1): b + 1])
            out.append((f"{name} @loop 0x{addr[b]:x}->0x{tgt:x}", head + list(ins[t: t + SEAM]), len(head)))
    return out


def violations(kernels) -> list[tuple[str, int, int, str, str]]:
    """(kernel, instruction index of the MFMA, wait states found, the writer, the MFMA) for every VALU-write -> MFMA-read closer than
    MIN_WAIT_STATES.  Linear walk (a branch target in between can only make the real distance longer or the pair unrelated)."""
    bad = []
    for name, ins in kernels:
        last = {}                                        # vgpr -> index of the VALU instruction that wrote it last
        for k, line in enumerate(ins):
            op = line.split()[0]
            args = re.split(r",\s*", line[len(op):].strip())
            if op.startswith(("v_mfma", "v_smfmac")):
                for tok in args[1:4]:
                    for r in _regs(tok):
                        kw = last.get(r)
                        if kw is None or k - kw > MIN_WAIT_STATES:
                            continue
                        ws = 0
                        for q in range(kw + 1, k):
                            o = ins[q].split()
                            ws += int(o[1], 0) + 1 if o[0] == "s_nop" else 1
                        if ws < MIN_WAIT_STATES:
                            bad.append((name, k, ws, ins[kw], line))
                for r in _regs(args[0]):
                    last.pop(r, None)
            elif op.startswith("v_"):
                if not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                    for r in _regs(args[0]):
                        last[r] = k
            elif op.startswith(("ds_read", "ds_load", "global_load", "buffer_load", "scratch_load", "flat_load")):
                for r in _regs(args[0]):
                    last.pop(r, None)
    return bad


def _passes(op: str) -> int:
    """matrix-pipe passes of an MFMA mnemonic (4 cycles each); unknown shapes count as the longest"""
    if "32x32x16" in op or "16x16x4_f32" in op or "16x16x4f32" in op or "32x32x8" in op and "xf32" not in op:
        return 8
    if "16x16x32" in op or "16x16x16" in op:
        return 4
    return 16


def mfma_result_reads(kernels) -> list[tuple[str, int, int, int, str, str]]:
    """the other direction: (kernel, index, wait states found, wait states needed, the MFMA, the reader) for every VALU instruction that
    reads an MFMA result sooner than passes + 2 wait states after the MFMA (the least hipcc keeps for its own instructions on gfx950)"""
    bad = []
    for name, ins in kernels:
        last = {}                                        # vgpr -> (index of the MFMA that wrote it, passes)
        for k, line in enumerate(ins):
            op = line.split()[0]
            args = re.split(r",\s*", line[len(op):].strip())
            if op.startswith(("v_mfma", "v_smfmac")):
                for r in _regs(args[0]):
                    last[r] = (k, _passes(op))
                continue
            if op.startswith("v_"):
                for tok in args[1:]:
                    for r in _regs(tok.split()[0] if tok else tok):
                        hit = last.get(r)
                        if hit is None:
                            continue
                        km, np_ = hit
                        need = np_ + 2
                        if k - km > need:
                            continue
                        ws = 0
                        for q in range(km + 1, k):
                            o = ins[q].split()
                            if o[0] == "s_nop":
                                ws += int(o[1], 0) + 1
                            elif o[0].startswith(("v_mfma", "v_smfmac")):
                                ws += _passes(o[0])              # a following MFMA occupies the pipe for its own passes first
                            else:
                                ws += 1
                        if ws < need:
                            bad.append((name, k, ws, need, ins[km], line))
                for r in _regs(args[0]):
                    last.pop(r, None)
            elif op.startswith(("ds_read", "ds_load", "global_load", "buffer_load", "scratch_load", "flat_load")):
                for r in _regs(args[0]):
                    last.pop(r, None)
    return bad


_TRANS = ("v_exp_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_sin_", "v_cos_")


def trans_result_reads(kernels) -> list[tuple[str, int, int, str, str]]:
    """third rule (gfx940+ "trans forwarding hazard"): a non-transcendental VALU instruction must not read the result of a
    transcendental in the very next issue slot - (kernel, index, 0, the transcendental, the reader)"""
    bad = []
    for name, ins in kernels:
        for k in range(len(ins) - 1):
            op = ins[k].split()[0]
            if not op.startswith(_TRANS):
                continue
            dst = set(_regs(re.split(r",\s*", ins[k][len(op):].strip())[0]))
            nxt = ins[k + 1]
            op2 = nxt.split()[0]
            if not op2.startswith("v_") or op2.startswith(_TRANS):
                continue
            for tok in re.split(r",\s*", nxt[len(op2):].strip())[1:]:
                if dst & set(_regs(tok.split()[0] if tok else tok)):
                    bad.append((name, k + 1, 0, ins[k], nxt))
                    break
    return bad


_SREG = re.compile(r"^(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|m0|exec)$")


def _sregs(tok: str) -> list:
    tok = tok.strip()
    m = re.match(r"^s\[(\d+):(\d+)\]$", tok)
    if m:
        return [("s", i) for i in range(int(m.group(1)), int(m.group(2)) + 1)]
    m = re.match(r"^s(\d+)$", tok)
    if m:
        return [("s", int(m.group(1)))]
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return [("vcc", 0)]
    if tok == "m0":
        return [("m0", 0)]
    return []


def _wait_states(ins, a, b) -> int:
    """wait states between instruction a and instruction b (exclusive): every instruction in between is one, s_nop N is N + 1"""
    ws = 0
    for q in range(a + 1, b):
        o = ins[q].split()
        ws += int(o[1], 0) + 1 if o[0] == "s_nop" else 1
    return ws


_VMEM = ("buffer_", "global_", "flat_", "scratch_", "tbuffer_")


def sgpr_and_lane_hazards(kernels) -> list[tuple[str, int, int, str, str]]:
    """The data hazards of hand-placed (inline-asm) code that the three MFMA rules do not cover - hipcc pads them for its own
    instructions only:
      * a VALU instruction that writes an SGPR / VCC (v_readfirstlane, v_readlane, v_cmp ... with an SGPR destination) and a VMEM
        instruction that reads that SGPR (address, descriptor, offset): 5 wait states;
      * a VALU write of a VGPR and a v_permlane* that reads it: 2 wait states; a v_readlane / v_readfirstlane that reads it: 1;
      * a write of M0 (s_mov_b32 m0 ..., s_add ... m0) and an LDS-DMA load (`... lds`, `*_load_lds_*`) right behind it: 1 wait state
        (the same for s_setreg_b32 in front of an LDS-DMA load).
    -> (kernel, index of the reader, wait states found, the writer, the reader)"""
    bad = []
    for name, ins in kernels:
        s_last, v_last, m0_last = {}, {}, None
        for k, line in enumerate(ins):
            op = line.split()[0]
            args = [a.strip() for a in re.split(r",\s*", line[len(op):].strip())] if len(line) > len(op) else []
            is_vmem = op.startswith(_VMEM)
            if is_vmem:
                for tok in args:
                    for r in _sregs(tok.split()[0] if tok else tok):
                        kw = s_last.get(r)
                        if kw is not None and _wait_states(ins, kw, k) < 5:
                            bad.append((name, k, _wait_states(ins, kw, k), ins[kw], line))
                if (" lds" in line or "_lds_" in op) and m0_last is not None and _wait_states(ins, m0_last, k) < 1:
                    bad.append((name, k, 0, ins[m0_last], line))
            if op.startswith(("v_permlane", "v_readlane", "v_readfirstlane")):
                need = 2 if op.startswith("v_permlane") else 1
                srcs = args if op.startswith("v_permlane") else args[1:2]          # permlane swaps read (and write) both operands
                for tok in srcs:
                    for r in _regs(tok):
                        kw = v_last.get(r)
                        if kw is not None and _wait_states(ins, kw, k) < need:
                            bad.append((name, k, _wait_states(ins, kw, k), ins[kw], line))
            # ---- writers
            if op.startswith("v_") and args:
                dst = args[0]
                if _sregs(dst):
                    for r in _sregs(dst):
                        s_last[r] = k
                elif op.startswith("v_cmp") and not dst.startswith("v"):
                    s_last[("vcc", 0)] = k
                else:
                    for r in _regs(dst):
                        v_last[r] = k
                    if op.startswith("v_permlane") and len(args) > 1:
                        for r in _regs(args[1]):
                            v_last[r] = k
            elif op.startswith("s_") and args:
                for r in _sregs(args[0]):
                    s_last.pop(r, None)                      # a SALU write ends the VALU-written state of that SGPR
                if args[0] == "m0" or op.startswith("s_setreg"):
                    m0_last = k
            if is_vmem or op.startswith("ds_"):
                for r in (_regs(args[0]) if args else []):
                    v_last.pop(r, None)
    return bad


# ------------------------------------------------------------------------------------------------ every control-flow path
# The walks above are linear (+ the seams of backward branches).  A FORWARD branch that skips code makes the executed distance between
# a writer in front of it and a reader behind its target SHORTER than the linear one: the skipped instructions are no wait states (an
# `if (run ends here) { emit }` between an asm-written plane word and the MFMA that reads it).  path_hazards() therefore repeats the
# register rules on the control-flow graph: from every reader it walks BACKWARDS along every path (fall-through and taken branches),
# counting the wait states that path really executes, until the rule's distance is covered or a write of the register is found.
def _cfg(ins):
    """predecessors of every instruction: the previous one unless that is an unconditional branch / s_endpgm, plus every branch that
    targets it"""
    n = len(ins)
    addr, target = getattr(ins, "addr", None) or [None] * n, getattr(ins, "target", None) or [None] * n
    index_of = {a: i for i, a in enumerate(addr) if a is not None}
    preds = [[] for _ in range(n)]
    for i in range(n):
        op = ins[i].split()[0]
        if i + 1 < n and not op.startswith(("s_branch", "s_endpgm", "s_setpc", "s_trap")):
            preds[i + 1].append(i)
        if target[i] is not None and target[i] in index_of:
            preds[index_of[target[i]]].append(i)
    return preds


def _parse(line):
    op = line.split()[0]
    args = [a.strip() for a in re.split(r",\s*", line[len(op):].strip())] if len(line) > len(op) else []
    return op, args


def _v_writes(op, args):
    """VGPR / AGPR registers an instruction writes (loads included: they end a VALU-written state)"""
    if not args:
        return []
    if op.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")):
        return []
    if op.startswith("v_permlane") and "swap" in op:
        return _regs(args[0]) + (_regs(args[1]) if len(args) > 1 else [])
    if op.startswith("v_") or op.startswith(("ds_read", "ds_load", "global_load", "buffer_load", "scratch_load", "flat_load")):
        return _regs(args[0])
    return []


def _weight(op, args, mfma_passes=False):
    if op == "s_nop":
        return int(args[0], 0) + 1
    if mfma_passes and op.startswith(("v_mfma", "v_smfmac")):
        return _passes(op)
    return 1


def path_hazards(kernels) -> list[tuple[str, int, int, str, str]]:
    """(kernel, index of the reader, wait states on the shortest offending path, the writer, the reader) for
      * VALU write -> MFMA read (SrcA / SrcB / SrcC): 2 wait states;
      * MFMA result -> VALU / LDS / VMEM read: passes + 2;
      * transcendental result -> non-transcendental VALU read: 1;
      * VALU write -> v_permlane* read: 2, -> v_readlane / v_readfirstlane read: 1; -> the DPP source of a *_dpp instruction: 2
    evaluated along EVERY path of the control-flow graph."""
    bad = []
    for name, ins in kernels:
        parsed = [_parse(l) for l in ins]
        writes = [set(_v_writes(op, a)) for op, a in parsed]
        preds = _cfg(ins)

        def search(k, regs, need, writer_ok, mfma_passes):
            """shortest-path wait states from a hazardous writer of any of `regs` to instruction k; None when every path is long enough"""
            worst = None
            stack = [(p, 0, frozenset(regs)) for p in preds[k]]
            seen = {}
            while stack:
                i, ws, live = stack.pop()
                if ws >= need or not live:
                    continue
                key = (i, live)
                if seen.get(key, 1 << 30) <= ws:
                    continue
                seen[key] = ws
                op, a = parsed[i]
                hit = writes[i] & live
                if hit:
                    if writer_ok(op):
                        if worst is None or ws < worst[0]:
                            worst = (ws, i)
                    live = live - hit                  # any write ends the search for that register on this path
                    if not live:
                        continue
                ws2 = ws + _weight(op, a, mfma_passes)
                for p in preds[i]:
                    stack.append((p, ws2, live))
            return worst

        is_mfma = lambda o: o.startswith(("v_mfma", "v_smfmac"))          # noqa: E731
        is_valu = lambda o: o.startswith("v_") and not is_mfma(o)        # noqa: E731
        for k, (op, a) in enumerate(parsed):
            if is_mfma(op):
                regs = [r for tok in a[1:4] for r in _regs(tok)]
                w = search(k, regs, MIN_WAIT_STATES, is_valu, False)
                if w:
                    bad.append((name, k, w[0], ins[w[1]], ins[k]))
                continue
            reads = []
            if op.startswith("v_permlane"):
                reads = [r for tok in a[:2] for r in _regs(tok)]
                w = search(k, reads, 2, is_valu, False)
                if w:
                    bad.append((name, k, w[0], ins[w[1]], ins[k]))
            elif op.startswith(("v_readlane", "v_readfirstlane")):
                reads = [r for tok in a[1:2] for r in _regs(tok)]
                w = search(k, reads, 1, is_valu, False)
                if w:
                    bad.append((name, k, w[0], ins[w[1]], ins[k]))
            elif op.endswith("_dpp") and len(a) > 1:
                # a DPP instruction reads its first source from OTHER lanes: a VALU write of that register needs two wait states in front of
                # it (hipcc spaces its own; the hand-written v_fmac_f32_dpp scan of mpnn_edge_bwd_co.hip interleaves four values instead)
                reads = _regs(a[1].split()[0])
                w = search(k, reads, 2, is_valu, False)
                if w:
                    bad.append((name, k, w[0], ins[w[1]], ins[k]))
            if op.startswith("v_") or op.startswith(("ds_write", "ds_store", "global_store", "buffer_store", "flat_store", "scratch_store")):
                srcs = a[1:] if op.startswith("v_") else a
                regs = [r for tok in srcs for r in _regs(tok.split()[0] if tok else tok)]
                if regs:
                    # MFMA results: the longest rule first (the distance depends on the writer's passes: search with the 16-pass
                    # bound and compare with what the writer found needs)
                    w = search(k, regs, 18, is_mfma, True)
                    if w and w[0] < _passes(parsed[w[1]][0]) + 2:
                        bad.append((name, k, w[0], ins[w[1]], ins[k]))
                    if op.startswith("v_") and not op.startswith(_TRANS):
                        w = search(k, regs, 1, lambda o: o.startswith(_TRANS), False)
                        if w:
                            bad.append((name, k, w[0], ins[w[1]], ins[k]))
    return bad


def check(lib: str | None = None) -> list:
    """every violation of every rule in the linked library, loop seams included - [] is what ships"""
    from . import _lib
    return check_kernels(disassemble(lib or _lib.LIB_PATH))


def check_kernels(k) -> list:
    found = violations(k) + [(n, i, ws, m, r) for n, i, ws, _, m, r in mfma_result_reads(k)] + trans_result_reads(k) + sgpr_and_lane_hazards(k)
    seams = loop_seams(k)
    body = [(n, i) for n, i, _ in seams]
    first_behind = {n: f for n, _, f in seams}
    seen = {(w, m) for _, _, _, w, m in found}
    for hit in (violations(body) + [(n, i, ws, m, r) for n, i, ws, _, m, r in mfma_result_reads(body)] + trans_result_reads(body)
                + sgpr_and_lane_hazards(body)):
        n, i, ws, w, m = hit
        if i >= first_behind[n] and (w, m) not in seen:       # the reader sits behind the seam: a pair the linear walk cannot see
            found.append(hit)
    seen = {(w, m) for _, _, _, w, m in found}
    for hit in path_hazards(k):                               # ... and the pairs only a taken forward branch brings close enough
        if (hit[3], hit[4]) not in seen:
            seen.add((hit[3], hit[4]))
            found.append(hit)
    return found


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else None
    found = check(path)
    seen = set()
    for name, k, ws, w, m in found:
        if (name, w, m) not in seen:
            seen.add((name, w, m))
            print(f"{name[:70]}: {ws} wait state(s) between\n      {w}\n      {m}")
    print(f"{len(found)} VALU-write -> MFMA-read pair(s) closer than {MIN_WAIT_STATES} wait states / MFMA results read too early")
    sys.exit(1 if found else 0)
