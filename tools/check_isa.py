#!/usr/bin/env python
"""ISA gate: no kernel of libsimx_hip.so may spill VGPRs or use scratch memory unless it is on the allow-list below.

Reads the per-kernel resource reports simxns_amd/csrc/build.sh leaves next to the objects (<unit>.res: the compiler's
-Rpass-analysis=kernel-resource-usage remarks for the gfx950 code it emitted).  Exit 1 and a table on stderr when a hot-path
kernel has `ScratchSize > 0` or `VGPRs Spill > 0`.  Called by __graft_entry__.build(), tools/check_isa.sh and
tests/test_host_cpu.py::test_no_spills_on_the_hot_path.  `--table` prints every kernel's registers / occupancy."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "simxns_amd", "csrc")
CXXFILT = "c++filt"

# kernels that may use scratch: (regex on the demangled name, reason).  Nothing on the timed paths belongs here.
ALLOW = []


def parse(path):
    ks, cur = [], None
    for ln in open(path):
        ln = ln.strip()
        m = re.match(r"Function Name: (\S+)", ln)
        if m:
            cur = {"unit": os.path.basename(path)[:-4], "mangled": m.group(1)}
            ks.append(cur)
            continue
        m = re.match(r"([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)$", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return ks


def demangle(names):
    if not names:
        return []
    try:
        # (binutils' c++filt does not know _Float16's DF16_: spell it as the half type Dh for demangling only)
        out = subprocess.run([CXXFILT], input="\n".join(n.replace("DF16_", "Dh") for n in names), capture_output=True, text=True,
                             timeout=60).stdout.splitlines()
        return out if len(out) == len(names) else names
    except Exception:
        return names


def collect():
    ks = []
    for f in sorted(glob.glob(os.path.join(CSRC, "*.res"))):
        ks += parse(f)
    for k, d in zip(ks, demangle([k["mangled"] for k in ks])):
        k["name"] = re.sub(r"^void ", "", d)
    return ks


def offenders(ks):
    bad = []
    for k in ks:
        scratch, spill = int(k.get("ScratchSize", 0)), int(k.get("VGPRs Spill", 0))
        if (scratch or spill) and not any(re.search(p, k["name"]) for p, _ in ALLOW):
            bad.append((k["unit"], k["name"], scratch, spill, k.get("VGPRs"), k.get("AGPRs"), k.get("Occupancy")))
    return bad


def main():
    ks = collect()
    if not ks:
        sys.stderr.write("check_isa: no .res files under %s (run simxns_amd/csrc/build.sh)\n" % CSRC)
        return 2
    if "--table" in sys.argv:
        for k in ks:
            print("%-14s v%-4s a%-4s occ%-2s scratch%-4s spill%-3s lds%-6s %s" % (k["unit"], k.get("VGPRs"), k.get("AGPRs"), k.get("Occupancy"),
                                                                             k.get("ScratchSize"), k.get("VGPRs Spill"), k.get("LDS Size"), k["name"][:150]))
    bad = offenders(ks)
    for b in bad:
        sys.stderr.write("check_isa: %s: %s\n    scratch %d B/lane, %d VGPRs spilled (VGPRs %s, AGPRs %s, occupancy %s)\n" % b)
    print("check_isa: %d kernels in %d units, %d with scratch/spills outside the allow-list" % (len(ks), len(set(k["unit"] for k in ks)), len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
