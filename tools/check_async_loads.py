#!/usr/bin/env python3
"""Lint for the inline-asm asynchronous loads of the HIP kernels (rel_head.hip, linear.hip, ...).

The kernels issue `global_load_dwordx4` through inline asm and wait for them with hand-counted `s_waitcnt vmcnt(N)`, so
that many loads stay in flight.  The compiler does not know that the destination registers of such a load are not valid
yet: under register pressure it may copy or overwrite them before the data has landed, and the late data then clobbers
an unrelated value.  This script compiles a source file to gfx950 assembly and checks, for every vector memory load of
every kernel, that no instruction reads or writes its destination registers before an `s_waitcnt` with a vmcnt field has
been executed on every path from the load (data-flow over the basic blocks; counts are not verified, only that SOME
vmcnt wait separates the load from the first use).

    python tools/check_async_loads.py egtr_amd/csrc/rel_head.hip [more.hip ...]      exit status 1 on a finding"""
import re
import subprocess
import sys
import tempfile

REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
LOAD = re.compile(r"^\s*(global_load|flat_load|buffer_load|scratch_load)_\w+\s+(.*)$")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def split_kernels(asm_text):
    """{kernel name: [(asm line number, instruction text)]} -- labels kept as ('label', name) entries."""
    kernels, cur = {}, None
    for ln, line in enumerate(asm_text.splitlines(), 1):
        code = line.split(";")[0].rstrip()
        if not code.strip():
            continue
        m = re.match(r"^([\w.$]+):\s*$", code)
        if m:
            name = m.group(1)
            if not name.startswith(".") and not name.startswith("$"):
                cur = kernels.setdefault(name, [])
            elif cur is not None:
                cur.append((ln, "label", name))
            continue
        if code.lstrip().startswith(".") or cur is None:
            continue
        cur.append((ln, "ins", code.strip()))
    return kernels


def check_kernel(name, items):
    """Forward data-flow over the basic blocks: the set of registers that are the destination of a vector memory load
    with no `s_waitcnt vmcnt` executed since, on ANY path; an instruction touching such a register is a finding."""
    blocks, labels = [], {}
    cur = {"ins": [], "succ": [], "fall": True}
    blocks.append(cur)
    for ln, kind, text in items:
        if kind == "label":
            if cur["ins"] or len(blocks) == 1:
                cur = {"ins": [], "succ": [], "fall": True}
                blocks.append(cur)
            labels[text] = len(blocks) - 1
            continue
        cur["ins"].append((ln, text))
        op = text.split()[0]
        if op in ("s_branch",) or op.startswith("s_cbranch") or op == "s_endpgm":
            tgt = text.split()[1] if op != "s_endpgm" else None
            cur["succ_label"] = tgt
            cur["fall"] = op.startswith("s_cbranch")
            cur = {"ins": [], "succ": [], "fall": True}
            blocks.append(cur)
    for i, b in enumerate(blocks):
        if b.get("succ_label") in labels:
            b["succ"].append(labels[b["succ_label"]])
        if b["fall"] and i + 1 < len(blocks):
            b["succ"].append(i + 1)

    def transfer(b, pending, report):
        pending = dict(pending)
        for ln, ins in b["ins"]:
            if ins.startswith("s_waitcnt") and "vmcnt" in ins:
                pending = {}
                continue
            ops = ins.split(None, 1)[1] if " " in ins else ""
            m = LOAD.match(ins)
            is_dma = m is not None and ("_lds_" in ins.split()[0] or " lds" in ins)  # LDS-DMA: no destination register
            dest = regs(ops.split(",")[0]) if (m and not is_dma) else set()
            srcs = regs(ops.split(",", 1)[1]) if (m and not is_dma and "," in ops) else (regs(ops) if not dest else set())
            # a load that overwrites the destination of an in-flight load is harmless (loads return in order; the
            # compiler does it when the earlier value is dead): only sources count for a load, everything for the rest
            hit = srcs & pending.keys()
            if hit and report is not None:
                r = sorted(hit)[0]
                report.append((name, ln, ins, f"{r[0]}{r[1]} is the destination of the load at asm line {pending[r]}"))
            for r in dest:
                pending[r] = ln
        return pending

    entry = [dict() for _ in blocks]
    work = [0]
    seen = {0}
    while work:
        i = work.pop()
        out = transfer(blocks[i], entry[i], None)
        for j in blocks[i]["succ"]:
            merged = dict(entry[j])
            changed = False
            for r, ln in out.items():
                if r not in merged:
                    merged[r] = ln
                    changed = True
            if changed or j not in seen:
                entry[j] = merged
                seen.add(j)
                work.append(j)
    findings = []
    for i, b in enumerate(blocks):
        if i in seen:
            transfer(b, entry[i], findings)
    return findings


def check_asm(asm_text):
    findings = []
    for name, items in split_kernels(asm_text).items():
        findings.extend(check_kernel(name, items))
    return findings


def compile_to_asm(src):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        src, "-o", f.name], check=True, stderr=subprocess.DEVNULL)
        return open(f.name).read()


def main():
    bad = 0
    for src in sys.argv[1:]:
        findings = check_asm(compile_to_asm(src))
        print(f"{src}: {len(findings)} finding(s)")
        for k, ln, ins, why in findings[:20]:
            print(f"  {k}: asm line {ln}: `{ins}`: {why}")
        bad += len(findings)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
