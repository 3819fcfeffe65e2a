#!/usr/bin/env python3
"""SHA-256 of the machine code of one kernel (default: the headline's k_parser_reg<false, false, 4, 768>) inside a built
translation unit (default: fluent-bit_amd/csrc/build/kernels_tile.o): the gfx950 code object is taken out of the fat binary
and the bytes of the kernel's symbol are hashed -- the identity `bench.py` quotes the committed PMC summary under (an edit to a
shared header that leaves this kernel's instructions as they were keeps the summary valid; any change to them drops it).
    python3 tools/kernel_isa_sha.py [object [mangled-name-substring]]"""
import hashlib, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
HEADLINE = "k_parser_regILb0ELb0ELi4ELi768EE"


def kernel_isa_sha(obj=None, name=HEADLINE):
    obj = obj or os.path.join(ROOT, "fluent-bit_amd", "csrc", "build", "kernels_tile.o")
    if not os.path.exists(obj) or not os.path.exists(os.path.join(LLVM, "llvm-objcopy")):
        return None
    with tempfile.TemporaryDirectory() as t:
        fat, co = os.path.join(t, "fat.bin"), os.path.join(t, "dev.co")
        try:
            subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True, capture_output=True)
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + fat, "--output=" + co, "--unbundle"], check=True, capture_output=True)
            syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", co], check=True, capture_output=True, text=True).stdout
            secs = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-SW", co], check=True, capture_output=True, text=True).stdout
        except (subprocess.CalledProcessError, OSError):
            return None
        hit = None
        for ln in syms.splitlines():
            f = ln.split()
            if len(f) >= 8 and f[3] == "FUNC" and name in f[7] and not f[7].endswith(".kd"):
                hit = (int(f[1], 16), int(f[2], 0), f[6])
        if hit is None:
            return None
        addr, size, ndx = hit
        sec = None
        for ln in secs.splitlines():
            ln = ln.strip()
            if ln.startswith("[") and "]" in ln:
                idx = ln[1:ln.index("]")].strip()
                f = ln[ln.index("]") + 1:].split()
                if idx == ndx and len(f) >= 5:
                    sec = (int(f[2], 16), int(f[3], 16))                 # address, file offset
        if sec is None:
            return None
        data = open(co, "rb").read()
        off = sec[1] + (addr - sec[0])
        return hashlib.sha256(data[off:off + size]).hexdigest()[:16] + ":%d" % size


if __name__ == "__main__":
    print(kernel_isa_sha(*(sys.argv[1:3])))
