#!/usr/bin/env python
"""Per-kernel resource summary from a `hipcc -S --cuda-device-only` assembly file: VGPRs, SGPRs, scratch, code bytes,
occupancy, and a count of the instruction classes that matter for the scan kernels (VALU / LDS / vector memory / scalar
memory / barriers). Usage: python scripts/kernel_stats.py file.s [name-substring]"""
import re
import subprocess
import sys


def demangle(name):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        return name


def main():
    path = sys.argv[1]
    needle = sys.argv[2] if len(sys.argv) > 2 else ""
    cur, body, rows = None, [], []
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is None:
            continue
        body.append(line)
        if line.strip().startswith("; Occupancy:"):
            text = "".join(body)
            info = {k: int(v) for k, v in re.findall(r"; (codeLenInByte|NumVgprs|TotalNumSgprs|ScratchSize|Occupancy)[ :=]+(\d+)", text)}
            ops = [l.split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((";", "."))]
            cls = {"valu": sum(o.startswith("v_") for o in ops), "lds": sum(o.startswith("ds_") for o in ops),
                   "vmem": sum(o.startswith(("global_", "buffer_", "flat_", "scratch_")) for o in ops),
                   "smem": sum(o.startswith("s_load") or o.startswith("s_buffer_load") for o in ops),
                   "barrier": sum(o == "s_barrier" for o in ops), "waitcnt": sum(o == "s_waitcnt" for o in ops)}
            rows.append((cur, info, cls))
            cur = None
    for name, info, cls in rows:
        d = demangle(name)
        if needle and needle not in d:
            continue
        short = re.sub(r"\(.*", "", d)
        print(f"{short[:70]:70s} vgpr {info.get('NumVgprs', -1):3d} sgpr {info.get('TotalNumSgprs', -1):3d} scratch {info.get('ScratchSize', -1):3d} "
              f"code {info.get('codeLenInByte', -1):5d} occ {info.get('Occupancy', -1)} | valu {cls['valu']:4d} lds {cls['lds']:3d} vmem {cls['vmem']:3d} "
              f"smem {cls['smem']:2d} barrier {cls['barrier']} waitcnt {cls['waitcnt']}")


if __name__ == "__main__":
    main()
