"""Checks the generated gfx950 code of the box engine's sweep kernels (dune-ddm_amd/csrc/trsv_box.hpp) for a hazard the source cannot
rule out: the kernels request data two steps ahead with inline-asm loads and make it valid with a counted s_waitcnt; to the compiler
the destination registers of such a load hold their value right after the asm statement, so under register pressure it may COPY them
(v_mov, v_accvgpr_write) or re-use them before the data has arrived -- the copy then holds stale bits, and a stale address or product
index faults (seen with an instrumented build of the backward sweep, round 4).

For every inline-asm global load (into registers) of k_box_sweep<false/true> the script follows ALL control-flow paths from the load
to the s_waitcnt that covers it -- the third counted wait on the path (the same turn of the next trip through the three-step loop
body), any s_waitcnt vmcnt(0), or the wait in front of the loop for the first two sets requested there -- and reports every
instruction on the way that reads or writes one of the load's destination registers (a new request into the same registers ends the
path).  Exit code 1 on a finding.

usage: python tools/check_box_isa.py [file.s]      (without a file: compiles ddm_hip.hip to device assembly first, ~1 minute)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# kernel: (counted wait of a turn, head wait in front of the loop, requests per step)
KERNELS = {"_ZN3ddm11k_box_sweepILb0EEEvNS_9BoxParamsE": ("vmcnt(26)", "vmcnt(11)", 11), "_ZN3ddm11k_box_sweepILb1EEEvNS_9BoxParamsE": ("vmcnt(52)", "vmcnt(24)", 14)}


def regs(text):
    out = set()
    for m in re.finditer(r"\b[va]\[(\d+):(\d+)\]|\b[va](\d+)\b", text):
        if m.group(1):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def vregs(text):
    """VGPR numbers only (AGPRs are a different file)"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(1):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def device_asm():
    out = os.path.join(tempfile.mkdtemp(prefix="ddm_isa_"), "ddm_dev.s")
    csrc = os.path.join(ROOT, "dune-ddm_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                           "-S", "--cuda-device-only", "-o", out, "ddm_hip.hip"], cwd=csrc, stderr=subprocess.DEVNULL)
    return out


def check(path):
    s = open(path).read()
    findings = []
    for name, (wimm, himm, nb) in KERNELS.items():
        i = s.index(name + ":")
        lines = [l.strip() for l in s[i:s.index(".Lfunc_end", i)].split("\n")]
        n = len(lines)
        is_asm_load = lambda k: lines[k].startswith("global_load") and k > 0 and "ASMSTART" in lines[k - 1] and " lds" not in lines[k]
        labels = {l.split(":")[0]: k for k, l in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", l)}

        def succ(k):
            l = lines[k]
            m = re.match(r"s_branch\s+(\.LBB\d+_\d+)", l)
            if m:
                return [labels[m.group(1)]]
            if l.startswith("s_endpgm"):
                return []
            out = [k + 1] if k + 1 < n else []
            m = re.match(r"s_cbranch\S*\s+(\.LBB\d+_\d+)", l)
            if m:
                out.append(labels[m.group(1)])
            return out

        asm_loads = [k for k in range(n) if is_asm_load(k)]
        head, last_set = [n], set()                  # (the requests in front of the loop are drained completely: s_waitcnt vmcnt(0))
        nchecked = 0
        for k in asm_loads:
            dst = vregs(lines[k].split()[1].rstrip(","))
            # paths from the load to the wait that covers it: the BOX_DIST-th counted wait, any vmcnt(0), the head wait for the first two sets
            seen = set()
            stack = [(kk, 0) for kk in succ(k)]
            nchecked += 1
            while stack:
                kk, cnt = stack.pop()
                if (kk, cnt) in seen:
                    continue
                seen.add((kk, cnt))
                ll = lines[kk]
                if ll.startswith("s_waitcnt") and "vmcnt" in ll:
                    if "vmcnt(0)" in ll or (himm in ll and k not in last_set and k < head[0]):
                        continue
                    if wimm in ll:
                        cnt += 1
                        if cnt >= 3:
                            continue
                elif ll and ll[0] not in ";." and not ll.startswith("s_"):
                    ops = ll.split(None, 1)
                    if len(ops) == 2 and vregs(ops[1].split(";")[0]) & dst:
                        if is_asm_load(kk) and vregs(ll.split()[1].rstrip(",")) >= dst:
                            continue                  # requested again: a new life of the registers
                        findings.append(f"{name}: registers of `{lines[k][:60]}` (line {k}) touched by `{ll[:70]}` (line {kk}) before their wait")
                        break
                stack.extend((x, cnt) for x in succ(kk))
        print(f"{name}: {nchecked} inline-asm loads followed to their waits")
    return findings


if __name__ == "__main__":
    f = check(sys.argv[1] if len(sys.argv) > 1 else device_asm())
    for line in f:
        print("HAZARD", line)
    print("no hazard found" if not f else f"{len(f)} hazard(s)")
    sys.exit(1 if f else 0)
