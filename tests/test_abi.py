"""CPU-only: the C-ABI library builds/loads and exports every symbol include/pixelsynth_hip.h declares;
host-only entry points (no GPU work) are exercised against the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import c_oracle
from pixelsynth_amd import _lib, synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(name="pixelsynth_hip.h"):
    txt = open(os.path.join(ROOT, "include", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ps_[a-z0-9_]+)\s*\(", txt)))


DEBUG_ONLY = {"ps_pixelcnn_set_tuning", "ps_pixelcnn_get_tuning", "ps_pixelcnn_time_ar_run_waves", "ps_pixelcnn_time_ar_run_waves_range", "ps_pixelcnn_time_column_step",
              "ps_pixelcnn_debug_cache", "ps_pixelcnn_launch_kinds", "ps_pixelcnn_launch_kind_name", "ps_pixelcnn_launch_counts",
              "ps_pixelcnn_profile_begin", "ps_pixelcnn_profile_end"}


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names, debug = header_symbols(), header_symbols("pixelsynth_hip_debug.h")
    assert len(names) >= 10
    for n in names + debug:
        assert hasattr(L, n), f"{n} declared in include/ but not exported"
    assert set(debug) == DEBUG_ONLY and not set(names) & DEBUG_ONLY, "measurement / tuning entry points belong in pixelsynth_hip_debug.h"
    assert set(_lib.exported_symbols()) == set(names) | set(debug), "python prototypes out of sync with the headers"
    # ... and nothing else of the ps_ namespace leaves the library
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("ps_")}
    assert exported == set(names) | set(debug), exported ^ (set(names) | set(debug))
    assert L.ps_abi_version() == 2
    # the library says what it was built from: the in-tree one is a product build (no tuning / trace / experiment macro)
    info = L.ps_build_info().decode()
    assert "abi 2" in info and "gfx950" in info and ("product build" in info) == (not os.environ.get("PS_HIP_LIB")), info
    assert [L.ps_pixelcnn_launch_kind_name(k).decode() for k in range(L.ps_pixelcnn_launch_kinds())][-3:] == ["k_gemm_ws<0>", "k_gemm_ws<1>", "k_gemm_ws<2>"]


def test_error_channel():
    L = _lib.lib()
    rc = L.ps_custom_order(3, 4, None, None)
    assert rc < 0 and b"null" in L.ps_last_error()
    assert L.ps_splat_workspace_bytes(0, 10, 16, 4.0) == 0
    assert L.ps_splat_workspace_bytes(1, 65536, 256, 4.0) > 65536 * 8


def test_custom_order_matches_oracle():
    L = _lib.lib()
    for name, D in syn.distance_maps():
        d = D.copy()
        order = np.empty((1024, 2), np.int32)
        _lib.check(L.ps_custom_order(32, 32, _lib.ptr(d), _lib.ptr(order)), "ps_custom_order")
        ref, dref = c_oracle.custom_idx(32, 32, D)
        assert np.array_equal(order, ref), name
        assert np.array_equal(d, dref)  # multiplied by 10000 in place like the reference


def test_kernel_masks_match_oracle():
    L = _lib.lib()
    rs = np.random.RandomState(9)
    for n in (4, 8, 32):
        D = rs.randint(-5, 6, size=(n, n)).astype(np.int64)
        order, _ = c_oracle.custom_idx(n, n, D)
        for dil, typ in ((1, "A"), (1, "B"), (2, "B"), (3, "B")):
            m = np.empty((9, n * n), np.float32)
            _lib.check(L.ps_kernel_masks_f32(_lib.ptr(order), n * n, n, n, 3, dil, int(typ == "B"), _lib.ptr(m)),
                       "ps_kernel_masks_f32")
            assert np.array_equal(m[None], c_oracle.unfolded_masks(order, n, n, 3, dil, typ))
    bad = np.zeros((16, 2), np.int32)
    m = np.empty((9, 16), np.float32)
    assert L.ps_kernel_masks_f32(_lib.ptr(bad), 16, 4, 4, 3, 1, 1, _lib.ptr(m)) < 0  # location visited twice


def test_generation_order_matches_oracle():
    L = _lib.lib()
    for name, bg in syn.background_masks(256).items():
        bgu = np.ascontiguousarray(bg, dtype=np.uint8)
        order = np.empty((1024, 2), np.int32)
        blocks = np.empty((32, 32), np.uint8)
        D = np.empty((32, 32), np.int64)
        _lib.check(L.ps_generation_order(_lib.ptr(bgu), 256, 32, _lib.ptr(order), _lib.ptr(blocks), _lib.ptr(D)),
                   "ps_generation_order")
        ref = c_oracle.masks_for_background(bg, 32)
        assert np.array_equal(D, ref["D"]), name
        assert np.array_equal(order, ref["order"]), name
        assert np.array_equal(blocks, ref["bg32"]), name


def test_tuning_names_in_the_integration_notes_are_the_ones_the_library_knows():
    """INTEGRATION.md section E lists the tuning values by name; the table in csrc/lmconv.hip is what ps_pixelcnn_set_tuning accepts.
    Every documented name must exist, and every name of the product build must be documented."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "pixelsynth_amd", "csrc", "lmconv.hip")).read()
    table = src[src.index("const TuningEntry tuning_table[]"):src.index("const TuningEntry *find_tuning")]
    product = table.split("#ifdef PS_TUNING_BUILD")[0]
    known = set(re.findall(r'\{"(\w+)", &Tuning::', product))
    tuning_only = set(re.findall(r'\{"(\w+)", &Tuning::', table)) - known
    assert known and tuning_only == {"column_debug"}
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = doc[doc.index("## E. Tuning values"):doc.index("## F. Two headers")]
    rows = [line.split("|")[1] for line in sec.splitlines() if line.startswith("| `")]
    documented = set(re.findall(r"`(\w+)`", " ".join(rows)))
    assert documented == known, (sorted(documented - known), sorted(known - documented))
    # and struct Tuning has a field for each
    hdr = open(os.path.join(root, "pixelsynth_amd", "csrc", "lmconv_handle.h")).read()
    fields = hdr[hdr.index("struct Tuning {"):hdr.index("};", hdr.index("struct Tuning {"))]
    for name in known | tuning_only:
        assert re.search(r"\b%s\b" % name, fields), name


def test_profiles_readme_lists_what_is_there():
    """profiles/README.md is how bench.py finds the PMC record of roofline.traffic (the LAST *_k_column_pmc.json it names for the run's
    number of views), and how a reader finds the evidence: every file it names exists, every file there is named."""
    import json
    import sys
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    text = open(os.path.join(root, "README.md")).read()
    named = set(re.findall(r"`(r\d\d_[\w.]+\.(?:csv|json|txt))`", text))
    there = {f for f in os.listdir(root) if f != "README.md"}
    assert named <= there, sorted(named - there)
    assert there <= named, sorted(there - named)
    sys.path.insert(0, os.path.dirname(root))
    import bench
    rec = bench.latest_pmc_record(128)
    assert rec is not None and rec["views"] == 128 and rec["traffic_bytes_per_launch"] > 0
    newest = sorted(f for f in there if f.endswith("_k_column_pmc.json"))[-1]
    assert json.load(open(os.path.join(root, newest)))["traffic_bytes_per_launch"] == rec["traffic_bytes_per_launch"], newest
