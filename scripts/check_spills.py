"""(build container, no GPU) register spills of the persistent / two-source GEMM instantiations: compiles koaf_gemm.hip with
-Rpass-analysis=kernel-resource-usage (3 min) and prints VGPRs / spilled VGPRs / scratch per instantiation that spills, against the
counts recorded when the round-4 regression was fixed (a 64-bit row index in the shared epilogue had cost every kernel ~30 VGPRs and
the persistent 1x1 kernels 40-90 spilled registers: +40 ms per step, invisible in the build output).
    python scripts/check_spills.py"""
import re, subprocess, sys
from pathlib import Path
CS = Path(__file__).resolve().parent.parent / "oaprogressionmmf_amd" / "csrc"
RECORDED = {   # template arguments -> spilled VGPRs at the fixed build (koaf_gemm_kernel<BM, BN, AM, BMD, TFA, TFB, VEC, F16, NT, ACT, EMIT, SD>)
    "128,128,0,6,1,0,1,1,256,0,0,0": 7, "128,128,1,6,1,0,1,1,256,0,0,0": 35, "128,128,1,6,0,0,1,1,256,0,0,0": 14,
    "128,128,2,6,0,0,1,1,256,0,0,0": 15, "128,128,0,6,2,0,1,1,256,0,0,0": 0, "128,128,0,6,3,0,1,1,256,0,0,0": 0,
    "128,64,12,6,0,0,1,1,256,0,0,0": 0, "256,128,9,6,0,0,1,1,512,0,0,0": 0,
    # the streamed kernels (A mode 13): the persistent one-source ones carry two k-tiles of the next tile through the epilogue
    "128,128,13,6,0,0,1,1,256,0,0,2": 16, "128,128,13,6,1,0,1,1,256,0,0,2": 17, "128,128,13,6,0,0,1,1,256,0,1,2": 21,
    "128,128,13,6,1,0,1,1,256,0,1,2": 22, "128,128,13,6,2,0,1,1,256,0,0,2": 0, "128,128,13,6,3,0,1,1,256,0,0,2": 0,
    "128,64,13,6,2,0,1,1,256,0,0,2": 0, "128,64,13,6,3,0,1,1,256,0,0,2": 0,
}
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-I../../include",
       "-mllvm", "-amdgpu-mfma-vgpr-form", "-Rpass-analysis=kernel-resource-usage", "-c", "koaf_gemm.hip", "-o", "/tmp/koaf_gemm_spills.o"]
txt = subprocess.run(cmd, cwd=CS, capture_output=True, text=True).stderr
bad = 0
for b in txt.split("remark: Function Name: ")[1:]:
    name = b.split(" ")[0]
    m = re.search(r"koaf_gemm_kernelI(.*?)EEv", name)
    if not m:
        continue
    args = ",".join(re.findall(r"L[ib](\d+)E", m.group(1)))
    vg, sp, sc = (int(re.search(p, b).group(1)) for p in (r"VGPRs: (\d+)", r"VGPRs Spill: (\d+)", r"ScratchSize \[bytes/lane\]: (\d+)"))
    rec = RECORDED.get(args)
    if sp or rec is not None:
        flag = "" if rec is None or sp <= rec + 8 else "   <-- MORE THAN RECORDED (%d)" % rec
        bad += bool(flag)
        print(f"<{args}>  VGPRs {vg}  spilled {sp}  scratch {sc} B{flag}")
sys.exit(1 if bad else 0)
