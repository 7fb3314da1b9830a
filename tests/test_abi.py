"""The C-ABI library loads and exports every symbol include/real3daug_hip.h declares (not gpu)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "real3daug_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(r3d_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(pkg):
    names = _declared()
    assert len(names) >= 17
    assert sorted(pkg._lib.EXPORTS) == names


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg._lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.r3d_version() == 0x00020005
    assert lib.r3d_build_info().decode().startswith("sources ")
    assert isinstance(lib.r3d_last_error(), bytes)


def test_batch_descriptor_matches_header(pkg):
    text = open(os.path.join(ROOT, "include", "real3daug_hip.h")).read()
    body = re.search(r"typedef struct r3d_batch \{(.*?)\} r3d_batch_t;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.replace("*", " ").split()
        first = names.index(next(n for n in names if n not in
                                 ("int32_t", "int64_t", "uint32_t", "uint16_t", "uint64_t", "float", "double",
                                  "void", "size_t")))
        fields += [n.strip(",") for n in names[first:]]
    assert [f for f, _ in pkg._lib.BatchDesc._fields_] == fields
    assert ctypes.sizeof(pkg._lib.BatchDesc) == 4 * 4 + 2 * 8 + 17 * 8 + 2 * 8


def test_place_query_matches_header(pkg):
    """r3d_place_query_t: field order of the ctypes mirror = field order of the header, natural
    alignment (the library never sees a query built with another layout than it reads)."""
    text = open(os.path.join(ROOT, "include", "real3daug_hip.h")).read()
    body = re.search(r"typedef struct r3d_place_query_t \{(.*?)\} r3d_place_query_t;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.replace("*", " ").replace("const", " ").split()
        first = names.index(next(n for n in names if n not in
                                 ("int32_t", "int64_t", "uint32_t", "uint8_t", "uint64_t", "float", "double")))
        fields += [re.sub(r"\[.*", "", n.strip(",")) for n in names[first:]]
    assert [f for f, _ in pkg._lib.PlaceQuery._fields_] == fields
    assert ctypes.sizeof(pkg._lib.PlaceQuery) == 7 * 8 + 2 * 8 + 10 * 4 + 32 * 4 + 4 * 8 + (10 + 8 + 2) * 8 + 2 * 8 + 2 * 4 + 8 + 5 * 8 + 8
    assert len(pkg.places.search_radii_sq()) == 49 and pkg.places.search_radii_sq()[0] == 0.1 ** 2


def test_bad_arguments_are_reported_without_a_gpu(pkg):
    lib = pkg._lib.load()
    assert lib.r3d_add_space_for_spherical(None, -1, None, None) == -1
    assert b"add_space" in lib.r3d_last_error()
    assert lib.r3d_fill_spherical(None, 0, None, None, None) == -1
    assert lib.r3d_front_view_workspace_bytes(112, 1440) >= 112 * 1440 * 8
    assert lib.r3d_batch_workspace_bytes(None) == 0
    # the later levels: placement search, cut boxes, rich map
    assert lib.r3d_places_workspace_bytes(0, 0) == 0 and lib.r3d_places_workspace_bytes(4, 3) > 4 * 360 * 8
    assert lib.r3d_find_possible_places(None, 1, 0, 0, 1, 0, None, 1, None, None, None, None, None, 0, None, None, 0, None) == -1
    assert b"places" in lib.r3d_last_error()
    assert lib.r3d_cut_boxes_workspace_bytes(1000, 0) == 0 and lib.r3d_cut_boxes_workspace_bytes(1000, 2) > 0
    assert lib.r3d_cut_boxes(None, 10, 5, 4, None, None, 1, 1, None, None, 10, None, 0, None) == -1
    assert lib.r3d_map_bounds(None, 10, None, None, None) == -1
    assert lib.r3d_map_finish(None, 0, None, None, None) == -1
    assert lib.r3d_places_chunk_ranges(None, 10, 4, None, None) == -1
    assert lib.r3d_places_chunk_ranges_f32(None, 10, None, None) == -1
    assert lib.r3d_batch_insert_many(None, 1, None, None, None, None, 1, None, None, None) == -1
    assert lib.r3d_batch_export_rows(None, None, None, None) == -1


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    with pytest.raises(pkg.R3DError):
        pkg.Real3DAug.insertion.add_space_for_spherical(np.zeros((4, 5)))
    with pytest.raises(pkg.R3DError):
        pkg.SceneBatch(1, 16, 16)
