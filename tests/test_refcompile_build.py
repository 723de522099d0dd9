"""VERDICT r02 item 6, the CPU half: the reference's src/nn/linear/linear.cpp compiles UNMODIFIED against the boundary's
headers (hostcpp/refshim + bm_hip.h / bm_layer.h / bm_functions.h) and every name it references is defined by the
boundary (zhilight_amd.build.build_refcompile fails on a leftover undefined symbol).  Needs the reference tree, i.e. this
container; elsewhere the prebuilt module (if it travelled) is only imported."""
import os
import sys

import pytest


def test_reference_linear_tu_builds_and_links_against_the_boundary():
    from zhilight_amd import _lib, build
    build.build()                   # no-op when up to date (tests/test_abi.py does the same)
    _lib.lib()
    have_reference = all(os.path.exists(os.path.join(build.REFERENCE, t)) for t in build.REF_TUS)
    path = build.build_refcompile() if have_reference else build.refcompile_target()
    if not (path and os.path.exists(path)):
        pytest.skip("no reference tree and no prebuilt module")
    sys.path.insert(0, os.path.dirname(path))
    try:
        import zl_reflinear
    finally:
        sys.path.pop(0)
    assert hasattr(zl_reflinear, "RefLinear") and zl_reflinear.weight_cache_size() == 0


def test_refshim_holds_no_reference_text():
    """the shim directory is this repository's own code: forwarding headers + aliases, no copied reference header"""
    shim = os.path.join(os.path.dirname(__file__), "..", "zhilight_amd", "hostcpp", "refshim")
    for root, _, files in os.walk(shim):
        for f in files:
            text = open(os.path.join(root, f)).read()
            assert "BMENGINE_EXPORT" not in text, f
            assert len(text.splitlines()) < 200, f
