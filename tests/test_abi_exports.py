"""CPU-side checks of the drop-in boundary: libfiesta_hip.so loads without a GPU, exports every symbol
include/fiesta_hip.h declares (and nothing is declared that the Python mirror does not bind), reports zero
devices instead of crashing, and FAILS LOUDLY -- no CPU fallback -- when asked to create a map without one.
"""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build_hip()
    import fiesta_amd
    return fiesta_amd.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from fiesta_amd import _lib
    declared = _lib.declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/fiesta_hip.h but not exported"
    assert sorted(lib._fiesta_signatures) == declared, "python binding and header disagree"
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (fiesta_hip_[a-z0-9_]+)", nm))
    assert exported == set(declared), exported ^ set(declared)


def test_struct_layouts_match_header():
    from fiesta_amd import _lib
    text = open(_lib.HEADER_PATH).read()

    def fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), text, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for nm_ in decl.split(None, 1)[1].split(","):
                out.append(re.sub(r"\[.*?\]", "", nm_).strip())
        return out
    assert fields("fiesta_hip_config") == [f[0] for f in _lib.Config._fields_]
    assert fields("fiesta_hip_stats") == [f[0] for f in _lib.Stats._fields_]
    assert fields("fiesta_hip_raycast_params") == [f[0] for f in _lib.RaycastParams._fields_]


def test_no_gpu_means_loud_failure_not_fallback(lib):
    import torch
    import fiesta_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible; the no-device behaviour is checked on the CPU container")
    assert fiesta_amd.device_count() == 0
    with pytest.raises(fiesta_amd.FiestaHipError):
        fiesta_amd.ESDFMap((0, 0, 0), 0.1, (1.0, 1.0, 1.0))
    assert b"no HIP device" in lib.fiesta_hip_last_error() or lib.fiesta_hip_last_error()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under fiesta_amd/ or include/ may reference it."""
    bad = []
    for base in ("fiesta_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    src = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"(from|import)\s+oracle|oracle_api\.h|pyoracle|libfiesta_port|libfiesta_ref", src):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_hand_scheduled_kernels_do_not_spill():
    """k_nn_fill_full, k_ft_x and k_ft_plane wait for LDS-DMA fetches with COUNTED s_waitcnt vmcnt(n) around inline asm: a scratch
    spill (an extra VMEM operation the count does not know) would make them consume stale records silently.  The
    code objects inside the built library are inspected: no scratch, no spilled registers."""
    import sys
    from fiesta_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import check_kernel_resources
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build_hip()
    guarded = check_kernel_resources.check(_lib.LIB_PATH)
    assert any("k_nn_fill_full" in k for k in guarded) and any("k_ft_x" in k for k in guarded)


def test_path_note_names_follow_the_header():
    """fiesta_hip_stats.path_notes: the names the Python statistics give the bits (`why`) are the header's FIESTA_HIP_NOTE_* in
    bit order, and the struct mirrors end in the same field."""
    import re
    from fiesta_amd._lib import Stats
    text = open(os.path.join(ROOT, "include", "fiesta_hip.h")).read()
    notes = re.findall(r"#define FIESTA_HIP_NOTE_(\w+) (0x[0-9a-fA-F]+)", text)
    assert len(notes) == len(Stats.NOTES) >= 10
    for bit, (name, value) in enumerate(notes):
        assert int(value, 16) == 1 << bit, (name, value)
        assert Stats.NOTES[bit] == name.lower(), (bit, name, Stats.NOTES[bit])
    assert Stats._fields_[-1][0] == "path_notes"
    assert re.search(r"int64_t path_notes;[^}]*\} fiesta_hip_stats;", text, re.S)
