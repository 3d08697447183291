"""Build-container check of ff_gemm_x3.hip's code generation: no scratch access and no s_waitcnt vmcnt(0) inside the MFMA runs
of any gemm_x3_kernel / gemm_dma_f32_kernel instantiation.  (The K loop reads its fragments with inline-assembly ds_reads the compiler does not
track: a spill of such a register before the loop's own lgkmcnt wait would store garbage; a vmcnt(0) would drain the LDS-DMA
pipeline every slice.)    python tools/check_x3_asm.py"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "x3.s")
        subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "faceformer_amd", "csrc"), "-S", "--cuda-device-only", "-o", out,
                        os.path.join(ROOT, "faceformer_amd", "csrc", "ff_gemm_x3.hip")], check=True, stderr=subprocess.DEVNULL)
        s = open(out).read()
    bad = 0
    for n in re.findall(r"^(_ZN12_GLOBAL__N_1\d+gemm_(?:x3|dma_f32)_kernelILi\d+ELi\d(?:ELi\d+)?EEEvNS_6X3ArgsE):", s, re.M):
        a = s.index("\n" + n + ":")
        b = s.index("s_endpgm", a)
        body = s[a:b].split("\n")
        idx = [i for i, l in enumerate(body) if "v_mfma" in l]
        runs, start, prev = [], idx[0], idx[0]
        for i in idx[1:]:
            if i - prev > 60:
                runs.append((start, prev))
                start = i
            prev = i
        runs.append((start, prev))
        sc = sum(sum(1 for l in body[x:y] if "scratch_" in l) for x, y in runs)
        vm = sum(sum(1 for l in body[x:y] if "vmcnt(0)" in l) for x, y in runs)
        desc = s[s.index(".amdhsa_kernel " + n):]
        regs = re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1)
        scratch = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", desc).group(1)
        tag = re.search(r"ILi(\d+)ELi(\d)E(?:Li(\d+)E)?", n)
        name = ("gemm_dma_f32_kernel" if "dma_f32" in n else "gemm_x3_kernel") + "<%s, %s%s>" % (
            tag.group(1), tag.group(2), (", %s terms" if "dma_f32" not in n else ", BN %s") % tag.group(3) if tag.group(3) else "")
        print("%s: %s VGPRs, %s B scratch (tile-end paths), MFMA runs %s, scratch ops inside %d, vmcnt(0) inside %d"
              % (name, regs, scratch, [y - x for x, y in runs], sc, vm))
        bad += sc + vm
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
