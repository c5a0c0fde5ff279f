"""The C-ABI library loads on a CPU-only box and exports every symbol include/rarc.h declares.
(No compute calls here: those need a GPU and live in the -m gpu tests.)"""
import ctypes
import os
import re

from rag_arc_amd.hip import binding as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "rarc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rarc_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(B.SIGNATURES)


def test_library_exports_every_declared_symbol():
    lib = B.load_library()
    for name in _declared():
        assert hasattr(lib, name), f"{name} missing from librarc_hip.so"
    assert lib.rarc_version() == 600
    assert lib.rarc_padded_dim(1) == 128 and lib.rarc_padded_dim(768) == 768 and lib.rarc_padded_dim(769) == 896
    assert lib.rarc_search_workspace_bytes(16384) > 256 * 16384 * 8


def test_argument_validation_needs_no_gpu():
    lib = B.load_library()
    rc = lib.rarc_search_f16(None, 10, 768, None, None, 1, 10, 128, 0, -1.0, 1.0, None, None, None, None, 0, 16384, None)
    assert rc == -1 and b"null pointer" in lib.rarc_last_error()
    rc = lib.rarc_rrf_fuse(None, None, 1, 1, 1, 60.0, 1, None, None, None, None)
    assert rc == -1
    try:
        B.check(rc, "rarc_rrf_fuse")
    except B.RarcError as e:
        assert "rarc_rrf_fuse" in str(e)
    else:
        raise AssertionError("check() must raise")


def test_round6_entries_validate_their_arguments_without_a_gpu():
    """rarc_search_batch, rarc_enc32_pack_query_weight, rarc_stream_read (ABI 600): bad arguments come back as error codes with a
    message before anything touches the device."""
    import ctypes

    lib = B.load_library()
    assert lib.rarc_version() == 600
    assert lib.rarc_search_batch(None, None) == -1 and b"null descriptor" in lib.rarc_last_error()
    bad = B.SearchBatch()                      # all zeros: row_format 0, no status buffer
    assert lib.rarc_search_batch(ctypes.byref(bad), None) == -1 and b"null status" in lib.rarc_last_error()
    bad.row_format = 9
    assert lib.rarc_search_batch(ctypes.byref(bad), None) == -1 and b"row_format" in lib.rarc_last_error()
    assert lib.rarc_enc32_pack_query_weight(None, 32, 128, None, None) == -1
    assert lib.rarc_enc32_pack_query_weight(ctypes.c_void_p(16), 33, 128, ctypes.c_void_p(16), None) == -4      # n not a multiple of 32
    assert lib.rarc_enc32_pack_query_weight(ctypes.c_void_p(16), 32, 100, ctypes.c_void_p(16), None) == -4      # k not a multiple of 128
    assert lib.rarc_stream_read(None, 1 << 20, None, None) == -1
    assert lib.rarc_stream_read(ctypes.c_void_p(8), 1 << 20, ctypes.c_void_p(16), None) == -1                    # source not 16-byte aligned


def test_engine_refuses_to_run_without_a_gpu():
    import pytest
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rag_arc_amd.hip.engine import FlatIndexF16

    with pytest.raises(B.RarcError):
        FlatIndexF16(384)


def test_model_structs_match_the_header(tmp_path):
    """Every ctypes mirror of a header struct — encoder (fp16 and fp32-class) and reranker LM, layer and model — has the
    header's size and the header's offset for EVERY field (gcc is the judge)."""
    import ctypes
    import subprocess

    from rag_arc_amd.hip import binding as B

    pairs = [("RarcEncLayer", B.EncLayer), ("RarcEncModel", B.EncModel), ("RarcEnc32Layer", B.Enc32Layer),
             ("RarcEnc32Model", B.Enc32Model), ("RarcLmLayer", B.LmLayer), ("RarcLmModel", B.LmModel),
             ("RarcIoStats", B.IoStats), ("RarcSearchBatch", B.SearchBatch)]
    lines = []
    for cname, cls in pairs:
        lines.append(f'printf("%zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("%zu\\n", offsetof({cname}, {fname}));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rarc.h"\nint main(void){' + "".join(lines) + "return 0;}\n")
    exe = tmp_path / "layout"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = []
    for _, cls in pairs:
        want.append(ctypes.sizeof(cls))
        want += [getattr(cls, fname).offset for fname, _ in cls._fields_]
    assert got == want
