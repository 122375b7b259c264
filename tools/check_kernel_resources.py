#!/usr/bin/env python3
"""Build-time guard of the hand-scheduled kernels: k_nn_fill_full and k_ft_x / k_ft_plane wait for their LDS-DMA
fetches with COUNTED `s_waitcnt vmcnt(n)` around inline asm the compiler cannot see; an extra VMEM operation from the compiler --
a scratch spill, first of all -- would make a wait return early and stale records would be consumed silently.  This script reads
the code objects inside fiesta_amd/libfiesta_hip.so (its .hip_fatbin section: clang offload bundles; llvm-readelf --notes) and fails if one of
those kernels uses scratch (private_segment_fixed_size != 0) or spills registers.  Called by __graft_entry__.build() and by
tests/test_abi_exports.py::test_hand_scheduled_kernels_do_not_spill (CPU: no GPU needed)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GUARDED = ("k_nn_fill_full", "k_ft_x", "k_ft_plane")


def code_objects(so_path):
    """The gfx950 code objects embedded in the library: its .hip_fatbin section is a sequence of clang offload bundles (one per
    translation unit): magic, entry count, then (offset, size, id length, id) per entry."""
    import struct
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fatbin")
        subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so_path, os.path.join(tmp, "copy.so")], check=True)
        blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, at = [], blob.find(magic)
    while at >= 0:
        n = struct.unpack_from("<Q", blob, at + len(magic))[0]
        p = at + len(magic) + 8
        for _ in range(n):
            off, size, idl = struct.unpack_from("<QQQ", blob, p)
            ident = blob[p + 24:p + 24 + idl].decode()
            p += 24 + idl
            if "gfx950" in ident and size:
                out.append(blob[at + off:at + off + size])
        at = blob.find(magic, at + len(magic))
    return out


def kernel_resources(so_path):
    """{kernel symbol: {private_segment_fixed_size, sgpr_spill_count, vgpr_spill_count, vgpr_count}} of every gfx950 kernel."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for k, co in enumerate(code_objects(so_path)):
            f = os.path.join(tmp, f"co{k}")
            open(f, "wb").write(co)
            notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                if not name:
                    continue
                get = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", blk).group(1)) if re.search(rf"\.{key}:\s+(\d+)", blk) else 0  # noqa: E731
                out[name.group(1)] = {k2: get(k2) for k2 in ("private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count", "vgpr_count")}
    return out


def check(so_path=None):
    so_path = so_path or os.path.join(ROOT, "fiesta_amd", "libfiesta_hip.so")
    res = kernel_resources(so_path)
    guarded = {k: v for k, v in res.items() if any(g in k for g in GUARDED)}
    if not guarded:
        raise SystemExit("check_kernel_resources: none of the guarded kernels found in the library's code objects")
    bad = {k: v for k, v in guarded.items() if v["private_segment_fixed_size"] or v["vgpr_spill_count"] or v["sgpr_spill_count"]}
    if bad:
        raise SystemExit(f"check_kernel_resources: hand-scheduled kernels with scratch or spills (their counted vmcnt waits are no longer safe): {bad}")
    return guarded


if __name__ == "__main__":
    g = check(sys.argv[1] if len(sys.argv) > 1 else None)
    for k, v in sorted(g.items()):
        print(k[:90], v)
