"""Static instruction histogram per kernel of a gfx950 object file (llvm-objdump disassembly of the embedded device
code): VALU / SALU / LDS / vector-memory / scalar-memory / scratch instruction counts and a few named opcodes.

  python tools/isa_histogram.py <file.o> [name-substring ...] > profiles/rNN_isa_histogram.csv

Static counts over ALL branches of a kernel; the executed per-wave counts are the SQ_INSTS_* counters of
profiles/rNN_*sq_counters*.csv."""
import collections
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"


def device_elf(path, tmp):
    """The gfx950 code object inside a hipcc host object: .hip_fatbin section -> offload bundle -> ELF."""
    fat = os.path.join(tmp, "fat.bin")
    r = subprocess.run([OBJCOPY, "-O", "binary", "--only-section=.hip_fatbin", path, fat], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return path          # already a device code object
    out = os.path.join(tmp, "dev.co")
    for tgt in ("hipv4-amdgcn-amd-amdhsa--gfx950", "hip-amdgcn-amd-amdhsa--gfx950"):
        subprocess.run([BUNDLER, "--unbundle", "--type=o", f"--input={fat}", f"--output={out}", f"--targets={tgt}"],
                       capture_output=True, text=True)
        if os.path.exists(out) and os.path.getsize(out) > 0:
            return out
    return path


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    return "other"


def main():
    path, subs = sys.argv[1], sys.argv[2:]
    with tempfile.TemporaryDirectory() as tmp:
        elf = device_elf(path, tmp)
        txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", elf], capture_output=True, text=True).stdout
    cur, hist = None, collections.OrderedDict()
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            hist[cur] = collections.Counter()
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+([a-z_0-9]+)", line)
        if not m:
            continue
        op = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", m.group(1))
        hist[cur][classify(op)] += 1
        for named in ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_pk_fma_f32", "s_barrier", "s_waitcnt", "v_readfirstlane_b32",
                      "v_readlane_b32", "v_writelane_b32"):
            if op == named:
                hist[cur][named] += 1
    cols = ["valu", "salu", "smem", "lds", "vmem", "scratch", "mfma", "other", "v_fma_f32", "v_fmac_f32", "v_mul_f32",
            "v_pk_fma_f32", "s_barrier", "s_waitcnt", "v_readlane_b32", "v_writelane_b32"]
    print("kernel," + ",".join(cols))
    for k, c in hist.items():
        if subs and not any(s in k for s in subs):
            continue
        if sum(c.values()) < 50:
            continue
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*", "", name)[:80]
        print(name.replace(",", ";") + "," + ",".join(str(c.get(x, 0)) for x in cols))


if __name__ == "__main__":
    main()
