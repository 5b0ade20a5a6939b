"""Register / scratch / LDS usage of every kernel in libmfc.so, from the code objects embedded in the library
(the `.hip_fatbin` offload bundle holds one AMDGPU ELF per translation unit; their notes carry the kernel metadata).

    python tools/kernel_resources.py [libmfc.so]          # table
    from tools.kernel_resources import kernel_resources   # {kernel name: {"vgpr", "agpr", "sgpr", "scratch", "lds", "vgpr_spill", "sgpr_spill"}}
"""
import pathlib
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
EM_AMDGPU = 224


def _amdgpu_elfs(blob: bytes):
    pos = 0
    while True:
        pos = blob.find(b"\x7fELF", pos)
        if pos < 0:
            return
        hdr = blob[pos:pos + 64]
        if len(hdr) == 64 and hdr[4] == 2 and struct.unpack_from("<H", hdr, 18)[0] == EM_AMDGPU:
            shoff, = struct.unpack_from("<Q", hdr, 40)
            shentsize, shnum = struct.unpack_from("<HH", hdr, 58)
            size = shoff + shentsize * shnum
            yield blob[pos:pos + size]
            pos += size
        else:
            pos += 4


def kernel_resources(lib=None) -> dict:
    lib = pathlib.Path(lib) if lib else pathlib.Path(__file__).resolve().parents[1] / "meanflow_audio_codec_amd" / "csrc" / "libmfc.so"
    out = {}
    keys = {"vgpr": "vgpr_count", "agpr": "agpr_count", "sgpr": "sgpr_count", "scratch": "private_segment_fixed_size",
            "lds": "group_segment_fixed_size", "vgpr_spill": "vgpr_spill_count", "sgpr_spill": "sgpr_spill_count"}
    for elf in _amdgpu_elfs(lib.read_bytes()):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
            blk = ".agpr_count:" + blk
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            ent = {}
            for k, mk in keys.items():
                m = re.search(r"\." + mk + r":\s+(\d+)", blk)
                ent[k] = int(m.group(1)) if m else 0
            out[name.group(1)] = ent
    return out


if __name__ == "__main__":
    res = kernel_resources(sys.argv[1] if len(sys.argv) > 1 else None)
    dem = subprocess.run(["c++filt"], input="\n".join(res), capture_output=True, text=True).stdout.split("\n")
    for (k, e), d in sorted(zip(res.items(), dem), key=lambda x: x[1]):
        d = d.replace("(anonymous namespace)::", "")
        print(f"{d[:86]:86s} vgpr {e['vgpr']:4d} agpr {e['agpr']:3d} lds {e['lds']:6d} scratch {e['scratch']:5d} spill v{e['vgpr_spill']} s{e['sgpr_spill']}")
