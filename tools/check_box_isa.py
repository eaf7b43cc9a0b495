"""Checks the generated gfx950 code of the box engine's sweep kernels (dune-ddm_amd/csrc/trsv_box.hpp) for a hazard the source cannot
rule out: the kernels request data two steps ahead with inline-asm loads and make it valid with a counted s_waitcnt; to the compiler
the destination registers of such a load hold their value right after the asm statement, so under register pressure it may COPY them
(v_mov, v_accvgpr_write) or re-use them before the data has arrived -- the copy then holds stale bits, and a stale address or product
index faults (seen with an instrumented build of the backward sweep, round 4).

For every inline-asm global load of k_box_sweep<false/true> the script follows the instructions up to the s_waitcnt that covers the
load (the second-next counted wait inside the two-step loop body, the next wait in straight-line code) and reports every instruction
that reads or writes one of its destination registers.  The loads of the bounded spin loops are followed by s_waitcnt vmcnt(0) at once
and are skipped.  Exit code 1 on a finding.

usage: python tools/check_box_isa.py [file.s]      (without a file: compiles ddm_hip.hip to device assembly first, ~1 minute)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"_ZN3ddm11k_box_sweepILb0EEEvNS_9BoxParamsE": "vmcnt(15)", "_ZN3ddm11k_box_sweepILb1EEEvNS_9BoxParamsE": "vmcnt(28)"}


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
    for name, wimm in KERNELS.items():
        i = s.index(name + ":")
        lines = [l.strip() for l in s[i:s.index(".Lfunc_end", i)].split("\n")]
        is_asm_load = lambda k: lines[k].startswith("global_load") and k > 0 and "ASMSTART" in lines[k - 1]
        waits = [k for k, l in enumerate(lines) if wimm in l]
        assert len(waits) == 2, f"{name}: expected the two counted waits of the unrolled step loop, found {len(waits)}"
        labels = {l.split(":")[0]: k for k, l in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", l)}
        end = None
        for k in range(waits[1], len(lines)):
            m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", lines[k])
            if m and m.group(1) in labels and labels[m.group(1)] <= waits[0]:
                end = k
                break
        assert end is not None, f"{name}: back edge of the step loop not found"
        start = max(v for v in labels.values() if v <= waits[0])

        def spin_load(k):          # followed by s_waitcnt vmcnt(0) within a few lines: the bounded polls
            return any("vmcnt(0)" in lines[kk] for kk in range(k + 1, min(k + 6, len(lines))))

        def touches(k, path_):
            dst = vregs(lines[k].split()[1].rstrip(","))
            for kk in path_:
                ll = lines[kk]
                if not ll or ll[0] in ";." or kk == k:
                    continue
                ops = ll.split(None, 1)
                if len(ops) == 2 and vregs(ops[1].split(";")[0]) & dst:
                    return kk
            return None

        for k in range(start, end + 1):
            if not is_asm_load(k) or spin_load(k):
                continue
            if waits[0] < k < waits[1]:        # set A: valid at the first wait of the next trip
                t = touches(k, list(range(k + 1, end + 1)) + list(range(start, waits[0])))
            elif k > waits[1]:                 # set B: valid at the second wait of the next trip
                t = touches(k, list(range(k + 1, end + 1)) + list(range(start, waits[1])))
            else:
                continue
            if t is not None:
                findings.append(f"{name}: registers of `{lines[k][:60]}` (line {k}) touched by `{lines[t][:70]}` (line {t}) before their wait")
        for k in range(len(lines)):            # straight-line code in front of / behind the loop
            if (start <= k <= end) or not is_asm_load(k) or spin_load(k):
                continue
            dst = vregs(lines[k].split()[1].rstrip(","))
            for kk in range(k + 1, len(lines)):
                ll = lines[kk]
                if (ll.startswith("s_waitcnt") and "vmcnt" in ll) or ll.startswith(("s_cbranch", "s_branch", ".LBB")):
                    break
                ops = ll.split(None, 1)
                if ll and ll[0] not in ";." and len(ops) == 2 and not ll.startswith("global_load") and vregs(ops[1].split(";")[0]) & dst:
                    findings.append(f"{name}: registers of `{lines[k][:60]}` (line {k}) touched by `{ll[:70]}` (line {kk}) before a wait")
                    break
        nload = sum(is_asm_load(k) for k in range(len(lines)))
        print(f"{name}: {nload} inline-asm loads, step loop lines {start}..{end}, counted waits at {waits}")
    return findings


if __name__ == "__main__":
    f = check(sys.argv[1] if len(sys.argv) > 1 else device_asm())
    for line in f:
        print("HAZARD", line)
    print("no hazard found" if not f else f"{len(f)} hazard(s)")
    sys.exit(1 if f else 0)
