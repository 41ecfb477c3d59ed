"""The one-wave-per-SIMD MFMA kernels must REPORT the whole register file (512 = 256 arch VGPRs + 256 AGPRs) in the code objects the loader
sees, so that the dispatcher never places another kernel's wave on their SIMD (round 4's co-residency bug;
fvk_common.h: FVK_CLAIM_WHOLE_REGISTER_FILE).  Read from the BUILT library — the device ELFs are carved out of its .hip_fatbin section and their
AMDGPU metadata notes parsed with llvm-readelf — so a compiler that starts ignoring the asm clobber, or a new kernel of the family that forgets
the macro, fails here and not in somebody's two-stream run."""
import os
import re
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "fastvideo_amd", "libfvk_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin"
CLAIMING = ("gemm_w1_kernel", "gemm_w1n_kernel", "vae_conv3w_kernel", "attn_w16_kernel", "attn_w64_kernel", "attn_bs16_kernel", "mfma_sustained_probe_kernel")


def _device_elfs(blob: bytes):
    """Every AMDGPU ELF embedded in the fat binary: carve from the ELF magic to the end of its section header table."""
    out, pos = [], 0
    while True:
        pos = blob.find(b"\x7fELF", pos)
        if pos < 0:
            return out
        hdr = blob[pos:pos + 64]
        if len(hdr) == 64 and hdr[4] == 2 and struct.unpack_from("<H", hdr, 18)[0] == 224:   # ELF64, e_machine EM_AMDGPU
            e_shoff, = struct.unpack_from("<Q", hdr, 40)
            e_shentsize, e_shnum = struct.unpack_from("<HH", hdr, 58)
            out.append(blob[pos:pos + e_shoff + e_shentsize * e_shnum])
        pos += 4


def test_one_wave_per_simd_kernels_report_the_whole_register_file(tmp_path):
    objcopy, readelf = os.path.join(LLVM, "llvm-objcopy"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(LIB) and os.path.exists(objcopy) and os.path.exists(readelf)):
        pytest.skip("library not built or no llvm tools")
    fat = str(tmp_path / "fat.bin")
    subprocess.run([objcopy, "--dump-section", f".hip_fatbin={fat}", LIB, str(tmp_path / "copy.so")], check=True)
    elfs = _device_elfs(open(fat, "rb").read())
    assert len(elfs) >= 10, f"only {len(elfs)} device code objects found in .hip_fatbin"
    seen = {}
    for i, e in enumerate(elfs):
        f = str(tmp_path / f"co{i}.elf")
        open(f, "wb").write(e)
        notes = subprocess.run([readelf, "--notes", f], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            vg = re.search(r"\.vgpr_count:\s+(\d+)", blk)
            if name and vg and any(k in name.group(1) for k in CLAIMING) and "merge" not in name.group(1):
                seen[name.group(1)] = int(vg.group(1))
    kinds = {k: [n for n in seen if k in n] for k in CLAIMING}
    assert all(kinds.values()), f"kernel families missing from the library's metadata: {[k for k, v in kinds.items() if not v]}"
    short = {n: v for n, v in seen.items() if v != 512}
    assert not short, f"kernels that leave room for a foreign wave on their SIMD (vgpr_count != 512): {short}"
