"""Instruction mix of the koaf_gemm_kernel main loops, from the device assembly (runs in the build container, no GPU):
compiles csrc/koaf_gemm.hip with -S, finds for each requested instantiation the basic block holding the MFMAs and the
blocks of the same loop before it, and prints MFMA / vector / scalar / LDS / memory instruction counts per k-step and the
most frequent vector opcodes (every block between the loop header and the MFMA block is summed, so code that runs only
when the filter tap changes -- the dgrad gather -- is counted as if it ran every k-step).  Usage: python scripts/isa_mix.py [BM,BN,AM,BMD,TFA,TFB,VEC,F16,NT ...]
(default: 1x1 and 3x3 forward with the fp32 loader, gather / halo / K-major plane-image kernels, fp32-loader weight gradient, dense bf16)."""
import collections
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "oaprogressionmmf_amd" / "csrc" / "koaf_gemm.hip"
# BM,BN,AM,BMD,TFA,TFB,VEC,F16,NT (the template arguments of koaf_gemm_kernel; modes as in the source's enum)
DEFAULT = ["128,128,0,6,1,0,1,1,256", "128,128,1,6,1,0,1,1,256", "128,128,7,6,0,0,1,1,256", "256,128,9,6,0,0,1,1,512",
           "128,128,10,11,0,0,1,1,256", "128,128,3,4,2,1,1,1,256", "128,128,0,0,0,0,1,0,256"]


def mangled(sig):
    bm, bn, am, bmd, tfa, tfb, vec, f16, nt = sig.split(",")
    return f"koaf_gemm_kernelILi{bm}ELi{bn}ELi{am}ELi{bmd}ELi{tfa}ELi{tfb}ELb{vec}ELb{f16}ELi{nt}E"


def classify(ops):
    c = collections.Counter()
    for op in ops:
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "scratch_")): c["vmem"] += 1
    return c


def main():
    sigs = sys.argv[1:] or DEFAULT
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "gemm.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "-S", "--offload-device-only", "-O3", "-std=c++17", "--offload-arch=gfx950",
                        "-Wno-unused-function", f"-I{ROOT / 'include'}", "-mllvm", "-amdgpu-mfma-vgpr-form", str(SRC), "-o", str(out)],
                       check=True, stderr=subprocess.DEVNULL)
        txt = out.read_text().split("\n")
    for sig in sigs:
        name = mangled(sig)
        start = [i for i, l in enumerate(txt) if l.startswith("_ZN") and name in l and l.rstrip().endswith("KoafGemm")]
        if not start:
            print(f"{sig}: not instantiated")
            continue
        s = start[0]
        e = next(i for i in range(s, len(txt)) if txt[i].startswith("\t.amdhsa_kernel") or ".rodata" in txt[i])
        blocks, cur, label = [], [], "entry"
        for ln in txt[s:e]:
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                blocks.append((label, cur)); label, cur = m.group(1), []
            else:
                t = ln.strip()
                if t and not t.startswith((";", ".")):
                    cur.append(t.split()[0] + " " + " ".join(t.split()[1:]))
        blocks.append((label, cur))
        idx = {n: i for i, (n, _) in enumerate(blocks)}
        mi = next(i for i, (_, b) in enumerate(blocks) if sum(1 for x in b if x.startswith("v_mfma")) >= 8)
        # loop header = target of the back edge out of the MFMA block
        head = mi
        for x in blocks[mi][1]:
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", x)
            if m and idx[m.group(1)] <= mi:
                head = min(head, idx[m.group(1)])
        total = collections.Counter()
        ops = []
        for _, b in blocks[head:mi + 1]:
            o = [x.split()[0] for x in b]
            ops += o
            total += classify(o)
        print(f"== <{sig}>  loop blocks {blocks[head][0]}..{blocks[mi][0]}: " + ", ".join(f"{k} {v}" for k, v in total.items())
              + f"  -> {total['valu'] / max(1, total['mfma']):.1f} vector instructions per MFMA")
        vc = collections.Counter(o for o in ops if o.startswith("v_") and not o.startswith("v_mfma"))
        print("   " + ", ".join(f"{k} {v}" for k, v in vc.most_common(12)))


if __name__ == "__main__":
    main()
