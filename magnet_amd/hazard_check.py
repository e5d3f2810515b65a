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

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MIN_WAIT_STATES = 2
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


def disassemble(lib: str) -> list[tuple[str, list[str]]]:
    """[(kernel name, [instruction text ...])] of every gfx950 code object bundled in ``lib``"""
    if not os.path.exists(OBJDUMP):
        raise FileNotFoundError(OBJDUMP)
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            txt = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = (m.group(1), [])
                    out.append(cur)
                elif cur is not None and line.startswith("\t"):
                    ins = line.split("//")[0].strip()
                    if ins:
                        cur[1].append(ins)
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


def check(lib: str | None = None) -> list:
    from . import _lib
    k = disassemble(lib or _lib.LIB_PATH)
    return violations(k) + [(n, i, ws, m, r) for n, i, ws, _, m, r in mfma_result_reads(k)] + trans_result_reads(k)


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
